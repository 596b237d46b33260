"""Stream ordering of the overlapped step on ONE device, under a collective that behaves like RCCL's: enqueued on its own stream
behind the caller's current stream, finishing LATE, `wait()` only stream-ordered (the gloo-based 2-rank test cannot see a
missing wait: gloo synchronises at call time).  The fake collective flips the sign of the gradient slice -- Adam's update is
nearly invariant to a gradient SCALE, not to its sign -- after a long device-side delay; the run with the delay must reproduce,
bit for bit, the run in which every collective is followed by a device synchronisation.  A slice update issued ahead of its
all-reduce, a zero_grad ahead of the last all-reduce, or an optimizer step that does not wait for the optimizer stream all
change the parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class _Work(object):
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


def _run(monkeypatch, delay_cycles, sync, sectioned, steps=2):
    from daft_exprt import parallel
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams
    comm = torch.cuda.Stream(device=DEV)
    calls = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(comm):
            comm.wait_event(ready)                 # like NCCL: ordered behind the stream the call was made on
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)    # the collective finishes late
            t.mul_(-1.)
            done = torch.cuda.Event()
            done.record(comm)
        calls.append(t.numel())
        if sync:
            torch.cuda.synchronize()
        return _Work(done)

    monkeypatch.setattr(parallel.dist, 'all_reduce', fake_all_reduce)
    monkeypatch.setenv('DX_SECTIONED_ADAM', '1' if sectioned else '0')
    hp = make_hparams(compute_dtype='bf16', batch_size=4, accumulation_steps=2)
    torch.manual_seed(7)
    model = DaftExprt(hp).to(DEV).train()
    trainer = Trainer(model, hp, 1)
    trainer.world = trainer.reducer.world = 2      # two "ranks": the reducer issues its collectives
    for it in range(steps):
        micro = []
        for k in range(2):
            cb = synthetic_batch(hp, 4, seed=50 + 10 * it + k, t_max=200, force_first_full=True, l_range=(6, 30))
            inputs, targets, _ = model.parse_batch(DEV, cb)
            micro.append((inputs, targets))
        terms, gn = trainer.step(micro, 20000 + it)
    torch.cuda.synchronize()
    assert len(calls) == steps * len(trainer.reducer.buckets)     # one collective per bucket per optimizer step (last micro-batch only)
    return model.flat_parameters().clone(), float(gn), torch.cat([trainer.optimizer.exp_avg, trainer.optimizer.exp_avg_sq]).clone()


def _same(p_a, p_b, st_a, st_b, gn_a, gn_b):
    ''' two runs of the same step differ by the order of a few fp32 atomics (LayerNorm / FiLM gradient sums): Adam turns that
        into +-lr on the few per cent of the elements whose gradient is noise, and into ~1e-10 elsewhere; a misordered stream (wrong-sign
        or zeroed gradients) moves EVERY element by ~2 lr = 1.4e-3 '''
    moved = ((p_a - p_b).abs() > 1e-5).float().mean()
    assert float(moved) < 0.2, float(moved)      # measured 3-4 % between two identical runs; a misordered stream gives ~100 %
    m_a, m_b = st_a[: st_a.numel() // 2], st_b[: st_b.numel() // 2]
    # second step: the parameters the noisy elements moved to feed back into the gradients (measured 0.6 % of the largest first moment)
    assert float((m_a - m_b).abs().max()) <= 5e-2 * float(m_a.abs().max())
    assert float((m_a - m_b).abs().mean()) <= 2e-2 * float(m_a.abs().mean())     # a sign flip is 200 %
    assert abs(gn_a - gn_b) <= 1e-2 * abs(gn_b)


@pytest.mark.parametrize('sectioned', [True, False])
def test_late_collectives_do_not_change_the_step(monkeypatch, sectioned):
    p_ref, gn_ref, st_ref = _run(monkeypatch, 0, True, sectioned)
    p_late, gn_late, st_late = _run(monkeypatch, 4_000_000, False, sectioned)     # ~2 ms per collective at 2 GHz
    _same(p_ref, p_late, st_ref, st_late, gn_ref, gn_late)


def test_sectioned_and_whole_buffer_optimizer_agree(monkeypatch):
    ''' per-bucket Adam on the optimizer stream == the whole-buffer step after the backward pass (same arithmetic per element;
        the fused gradient-norm sum differs in summation order only) '''
    p_a, gn_a, st_a = _run(monkeypatch, 0, False, True)
    p_b, gn_b, st_b = _run(monkeypatch, 0, False, False)
    _same(p_a, p_b, st_a, st_b, gn_a, gn_b)
