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
    trainer.world = trainer.reducer.world = 2; trainer.reducer.active = True      # two "ranks": the reducer issues its collectives
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


def _run_snapshots(monkeypatch, producer_delay, sync, sectioned, steps=2):
    ''' world-size-2 semantics over a STREAM-ORDERED fake backend: the collective is work on its own stream that starts when the stream
        it was issued from reaches the call (NCCL / RCCL's contract; gloo synchronises on the host and hides a missing dependency).
        Every collective takes a SNAPSHOT of its bucket at the moment it executes.  `producer_delay` (cycles) makes the bucket's
        PRODUCERS late: a device-side sleep in front of every batch of weight-gradient launches on the side stream and in front of the
        data-gradient chain on the launch stream -- a collective (or a slice update) that is not ordered behind BOTH streams then
        snapshots a bucket that is still being written. '''
    from daft_exprt import parallel
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams
    comm = torch.cuda.Stream(device=DEV)
    snaps = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(comm):
            comm.wait_event(ready)
            snaps.append(t.clone())                # what a real all-reduce would have sent
            t.mul_(-1.)                            # (sign flip: Adam is blind to a gradient scale, not to its sign)
            done = torch.cuda.Event()
            done.record(comm)
        if sync:
            torch.cuda.synchronize()
        return _Work(done)

    monkeypatch.setattr(parallel.dist, 'all_reduce', fake_all_reduce)
    monkeypatch.setenv('DX_SECTIONED_ADAM', '1' if sectioned else '0')
    if producer_delay:
        orig_issue, orig_flush = DaftExprt._issue_side, DaftExprt._flush_wgrads

        def late_issue(self, pend):
            with torch.cuda.stream(self._side_stream):
                torch.cuda._sleep(producer_delay)              # the weight gradients of this block start late
            return orig_issue(self, pend)

        def late_flush(self):
            if self._wgrad_pending:
                torch.cuda._sleep(producer_delay // 4)         # ... and the data-gradient chain is held up as well
            return orig_flush(self)
        monkeypatch.setattr(DaftExprt, '_issue_side', late_issue)
        monkeypatch.setattr(DaftExprt, '_flush_wgrads', late_flush)
    hp = make_hparams(compute_dtype='bf16', batch_size=4, accumulation_steps=2)
    torch.manual_seed(7)
    model = DaftExprt(hp).to(DEV).train()
    trainer = Trainer(model, hp, 1)
    trainer.world = trainer.reducer.world = 2
    trainer.reducer.active = True
    for it in range(steps):
        micro = []
        for k in range(2):
            cb = synthetic_batch(hp, 4, seed=50 + 10 * it + k, t_max=200, force_first_full=True, l_range=(6, 30))
            inputs, targets, _ = model.parse_batch(DEV, cb)
            micro.append((inputs, targets))
        if sync:
            torch.cuda.synchronize()
        trainer.step(micro, 20000 + it)
    torch.cuda.synchronize()
    return snaps, model.flat_parameters().clone(), [n for n, _, _ in trainer.reducer.buckets]


@pytest.mark.parametrize('sectioned', [True, False])
def test_collective_reads_a_bucket_only_after_every_producer(monkeypatch, sectioned):
    ''' canary for the PRODUCER side of the overlap: with the weight-gradient stream and the launch stream both running late, what
        each collective reads must still be the finished bucket -- compared with the same step run under full synchronisation '''
    ref, p_ref, names = _run_snapshots(monkeypatch, 0, True, sectioned)
    late, p_late, _ = _run_snapshots(monkeypatch, 3_000_000, False, sectioned)        # ~1.5 ms per delay at 2 GHz, ~25 delays per step
    assert len(ref) == len(late) == 2 * len(names)
    for i, (a, b) in enumerate(zip(ref[:len(names)], late[:len(names)])):              # step 1: identical parameters in both runs
        assert a.shape == b.shape
        err = float((a - b).norm()) / (float(a.norm()) + 1e-30)
        assert err <= 1e-3, f'bucket "{names[i]}" was read {err:.3f} away from its finished value: a producer was still writing it'
    moved = ((p_ref - p_late).abs() > 1e-5).float().mean()
    assert float(moved) < 0.2, float(moved)


def test_stream_probe_tells_shared_from_separate_hardware_queues():
    ''' `streams.runs_beside`: a stream never runs beside itself; `streams.pick` returns a stream that does run beside the launch
        stream (and beside a second picked one), which is what the weight-gradient / optimizer streams are built from '''
    from daft_exprt import streams
    main = torch.cuda.current_stream()
    assert not streams.runs_beside(main, main)
    a = streams.pick([main], what='test stream')
    assert streams.runs_beside(a, main) and streams.runs_beside(main, a)
    b = streams.pick([main, a], what='second test stream')
    assert streams.runs_beside(b, main) and streams.runs_beside(b, a)
