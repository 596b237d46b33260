"""One rank of the data-parallel equivalence check (launched by tests/test_gpu_ddp.py under torch.distributed.run).

Reference semantics (`train.py:293, 379, 391`): DDP averages the per-rank gradients of `loss / accumulation_steps`.  Here every
rank runs the REAL `DaftExprt.forward_backward` with the `GradReducer.section_done` hook live (side-stream weight gradients,
asynchronous all-reduce per section, communication on the last micro-batch only) on its own micro-batches (rank-local
padding, SURVEY App. B: batches are NOT concatenated), and the reduced flat gradient must equal, element-wise, the sum over
all ranks' micro-batches computed by ONE process without any reducer, at the same 1 / (accum * world) scale.
Also: rank-0 parameters reach every rank (different seeds at construction), and after a full `Trainer.step` all ranks hold
bit-identical parameters.  fp32 operand mode, dropout off.  Exit code 0 = pass."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', rank % ndev)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams, no_dropout
    accum = 2
    hp = no_dropout(make_hparams(compute_dtype='fp32', batch_size=4, accumulation_steps=accum))
    torch.manual_seed(1000 + rank)                      # different initial weights per rank: the broadcast must win
    model = DaftExprt(hp).to(dev).train()
    model.set_rank(rank)
    if os.environ.get('DDP_TEST_REGROUP'):
        # the first process group "fails" the hardware-queue probe: GradReducer.pick_group must replace it by a fresh group
        from daft_exprt import streams
        real, calls = streams.collective_runs_beside, []
        streams.collective_runs_beside = lambda busy, group=None, **kw: (calls.append(1), real(busy, group, **kw) if len(calls) > 2 else False)[1]
    trainer = Trainer(model, hp, world)                 # broadcasts rank 0's parameters
    if os.environ.get('DDP_TEST_REGROUP'):
        assert 'group 2' in (trainer.reducer.queue_probe or ''), trainer.reducer.queue_probe
        assert trainer.reducer.group is not None and trainer.reducer.group is not dist.group.WORLD
    ref = None
    if rank == 0:
        torch.manual_seed(1000)
        ref = DaftExprt(hp).to(dev).train()
        assert torch.equal(ref.flat_parameters(), model.flat_parameters())
    chk = model.flat_parameters().double().sum().reshape(1)
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered), 'parameter broadcast failed'

    def micro_batches(r, it):
        out = []
        for k in range(accum):
            cb = synthetic_batch(hp, 4, seed=100 + 1000 * it + 10 * r + k, t_max=150 + 90 * r + 40 * k, force_first_full=True, l_range=(6, 30))
            inputs, targets, _ = model.parse_batch(dev, cb)
            out.append((inputs, targets))
        return out

    weights = trainer.criterion.weights(20000)
    scale = 1. / (accum * world)
    worst = 0.
    for it in range(3):
        model.zero_grad()
        mine = micro_batches(rank, it)
        for k, (inputs, targets) in enumerate(mine):
            hook = trainer.reducer.section_done if k == accum - 1 else None
            model.forward_backward(inputs, targets, weights, grad_scale=scale, section_done=hook)
        trainer.reducer.wait()
        torch.cuda.synchronize()
        g_dp = model.flat_gradients().clone()
        if rank == 0:
            ref.zero_grad()
            for r in range(world):
                for inputs, targets in micro_batches(r, it):
                    ref.forward_backward(inputs, targets, weights, grad_scale=scale)
            torch.cuda.synchronize()
            g_ref = ref.flat_gradients()
            off = 0
            gmax = float(g_ref.abs().max())
            for name, p in ref.named_parameters():
                n = p.numel()
                a, b = g_dp[off: off + n], g_ref[off: off + n]
                off += n
                err = float((a - b).abs().max())
                bound = 1e-5 * float(b.abs().max()) + 1e-6 * gmax
                worst = max(worst, err / bound)
                assert err <= bound, (it, name, err, float(b.abs().max()), gmax)
        # every rank must hold the same reduced gradient, bit for bit
        chk = g_dp.double().sum().reshape(1)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(g, gathered[0]) for g in gathered), 'ranks disagree on the reduced gradient'
        dist.barrier()
    # full optimizer steps through Trainer.step: parameters stay bit-identical across ranks
    solo = None
    if world == 1:
        # a forced one-rank world (DX_FORCE_DIST=1): the same steps by a trainer that issues no collective and runs the whole-buffer
        # optimizer must land on the same parameters, and the operand copies the next forward reads must have followed them
        forced = os.environ.pop('DX_FORCE_DIST', None)
        solo = Trainer(ref, hp, 1)
        if forced is not None:
            os.environ['DX_FORCE_DIST'] = forced
        assert not solo.reducer.active and not solo.sectioned
        assert trainer.reducer.active and (backend != 'nccl' or trainer.sectioned)
    for it in range(2):
        trainer.step(micro_batches(rank, 10 + it), 20000 + it)
        if solo is not None:
            solo.step(micro_batches(rank, 10 + it), 20000 + it)
            if it == 0:
                # not bit-equal by construction: gradients that are mathematically zero (the key third of every in_proj_bias: softmax
                # is invariant to it) or tiny are atomics-order noise, and the first Adam steps turn the SIGN of any gradient into a full
                # lr-sized update.  So: all but a sliver of the elements agree to 1e-6 after one step, and nothing differs by more than
                # two opposite updates.
                torch.cuda.synchronize()
                d = (model.flat_parameters() - ref.flat_parameters()).abs()
                lr = trainer.optimizer.param_groups[0]['lr']
                frac = float((d > 1e-6).float().mean())
                assert frac < 5e-4, f'per-bucket Adam behind the collectives != whole-buffer Adam: {frac:.2e} of the parameters differ by > 1e-6'
                assert float(d.max()) <= 2 * lr * 1.05, (float(d.max()), lr)
                for o in (trainer.optimizer, solo.optimizer):
                    assert o.step_count == 1
                dm = (trainer.optimizer.exp_avg - solo.optimizer.exp_avg).abs().max()
                assert float(dm) <= 1e-5 * float(solo.optimizer.exp_avg.abs().max()), float(dm)
    torch.cuda.synchronize()
    if solo is not None:
        # (the parameters were compared after the FIRST step: from the second on, the sign flips feed back through the forward pass)
        probe_in, probe_tg = micro_batches(rank, 77)[0]
        model.zero_grad(); ref.zero_grad()
        ta = model.forward_backward(probe_in, probe_tg, weights, grad_scale=1.)
        tb = ref.forward_backward(probe_in, probe_tg, weights, grad_scale=1.)
        torch.cuda.synchronize()
        assert torch.allclose(ta, tb, rtol=2e-3, atol=1e-5), (ta, tb)
    chk = model.flat_parameters().double().sum().reshape(1)
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered), 'parameters diverged across ranks after Trainer.step'
    if rank == 0:
        print(f'[ddp_worker] backend={backend} world={world} devices={ndev}: reduced gradients == single-process sum '
              f'(worst err/bound {worst:.3f}); parameters identical across ranks after 2 optimizer steps')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
