"""The captured step (`train.CapturedStep`: one hipGraph replay per optimizer step, per-step scalars in the device-side
`DxStepScalars` block) against the same step launched kernel by kernel.

What a replay must reproduce is everything the reference's trainer recomputes on the host every iteration and hands to its
kernels by value (train.py:139-151, 368-401, 491-494; loss.py:22-28): the learning rate, Adam's bias corrections, the adversarial
loss weight, and a fresh dropout stream.  A replay draws them from device memory; these tests pin that
  * the dropout masks of a replay are the eager step's masks for the same step id, bit for bit (learning rate 0, so both runs see
    identical parameters and the forward pass -- which has no atomics -- must agree exactly),
  * a training run on replays stays on the eager run's trajectory (parameters / Adam moments / loss terms / gradient norm; the
    two differ by the order of a few fp32 atomics only, like two eager runs do),
  * the schedule scalars really change between replays (warm-up learning rate and adversarial ramp at small iteration numbers),
  * the cache policy: second sight captures, new shapes run eager, a full cache does not thrash."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(monkeypatch, graph, accum=1, lr0=None, seed=11, n_batches=2, batch=4, **hp_extra):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams
    monkeypatch.setenv('DX_STEP_GRAPH', '1' if graph else '0')
    extra = dict(compute_dtype='bf16', batch_size=batch, accumulation_steps=accum)
    if lr0 is not None:
        extra.update(initial_learning_rate=lr0, max_learning_rate=lr0)
    extra.update(hp_extra)
    hp = make_hparams(**extra)
    torch.manual_seed(seed)
    model = DaftExprt(hp).to(DEV).train()
    trainer = Trainer(model, hp, 1)
    assert (trainer.captured is not None) == graph
    groups = []
    for i in range(n_batches):
        micro = []
        for k in range(accum):
            cb = synthetic_batch(hp, batch, seed=900 + 10 * i + k, t_max=160 + 40 * i, force_first_full=True, l_range=(8, 24))
            inputs, targets, _ = model.parse_batch(DEV, cb)
            micro.append((inputs, targets))
        groups.append(micro)
    return hp, model, trainer, groups


def _run(monkeypatch, graph, steps, it0, **kw):
    hp, model, trainer, groups = _setup(monkeypatch, graph, **kw)
    mels, terms, gns = [], [], []
    for s in range(steps):
        t, gn = trainer.step(groups[s % len(groups)], it0 + s)
        mels.append(model.last_outputs[3].clone())
        terms.append(t.clone())
        gns.append(gn.clone())
    torch.cuda.synchronize()
    return model, trainer, mels, torch.stack(terms).cpu(), torch.stack(gns).cpu().flatten()


@pytest.mark.parametrize('accum', [1, 2])
def test_replayed_dropout_masks_are_the_eager_masks(monkeypatch, accum):
    ''' learning rate 0: the parameters never move, so the forward output of step s depends on the step's dropout seeds alone '''
    steps = 7
    m_e, tr_e, mel_e, terms_e, _ = _run(monkeypatch, False, steps, 20000, accum=accum, lr0=0.)
    m_g, tr_g, mel_g, terms_g, _ = _run(monkeypatch, True, steps, 20000, accum=accum, lr0=0.)
    assert tr_g.captured.captures == 2 and tr_g.captured.replays == steps - 2, (tr_g.captured.captures, tr_g.captured.replays, tr_g.captured.broken)
    assert torch.equal(m_e.flat_parameters(), m_g.flat_parameters())
    for s in range(steps):
        assert torch.equal(mel_e[s], mel_g[s]), f'step {s}: the replay drew other dropout masks than the eager step'
    # consecutive replays of one graph differ from each other (the salt moves): not one frozen mask
    assert not torch.equal(mel_g[2], mel_g[4])
    assert torch.allclose(terms_e, terms_g, rtol=1e-5, atol=1e-6)      # loss sums use fp32 atomics


def test_replays_follow_the_eager_trajectory_and_schedule(monkeypatch):
    ''' iterations 1.. : warm-up learning rate and adversarial weight change every step (train.py:139-151, loss.py:22-28) '''
    steps = 8
    m_e, tr_e, _, terms_e, gn_e = _run(monkeypatch, False, steps, 1)
    m_g, tr_g, _, terms_g, gn_g = _run(monkeypatch, True, steps, 1)
    assert tr_g.captured.replays == steps - 2 and tr_g.captured.broken is None
    p_e, p_g = m_e.flat_parameters(), m_g.flat_parameters()
    moved = ((p_e - p_g).abs() > 2e-4).float().mean()     # 8 steps of <= 1.0007e-4 each: a wrong lr / bias correction / sign moves everything
    assert float(moved) < 0.05, float(moved)
    for a, b in ((tr_e.optimizer.exp_avg, tr_g.optimizer.exp_avg), (tr_e.optimizer.exp_avg_sq, tr_g.optimizer.exp_avg_sq)):
        assert float((a - b).abs().mean()) <= 8e-2 * float(a.abs().mean())   # (two eager runs differ by ~3 % after 8 steps: fp32-atomic order, fed back through Adam)
    assert tr_e.optimizer.step_count == tr_g.optimizer.step_count == steps
    # (bit-identity of replays is the lr = 0 test above; two EAGER runs of these 8 steps already differ by up to ~3 % per term)
    assert torch.allclose(terms_e, terms_g, rtol=6e-2, atol=1e-5), (terms_e - terms_g).abs().max()
    assert torch.allclose(gn_e, gn_g, rtol=1e-1)          # squared norms, 8 steps apart on a 4-utterance batch: the runs decorrelate by a few per cent
    # the speaker term carries the ramped weight: iteration * 1e-6 (it grows step by step also inside the replays)
    assert terms_g[-1, 0] > terms_g[2, 0] > 0.


def test_captured_learning_rate_is_read_from_the_step_block(monkeypatch):
    ''' two trainers replay the same graphs at different iterations: the parameter update scales with the schedule's lr '''
    hp, model, trainer, groups = _setup(monkeypatch, True, n_batches=1)
    for s in range(3):                                   # eager, capture + replay, replay
        trainer.step(groups[0], 1 + s)
    p0 = model.flat_parameters().clone()
    trainer.step(groups[0], 10)                          # lr(10) = 1e-4 + 9e-8 * 10
    d_small = (model.flat_parameters() - p0).abs().max()
    p1 = model.flat_parameters().clone()
    trainer.step(groups[0], 10000)                       # lr = 1e-3: ten times the step
    d_big = (model.flat_parameters() - p1).abs().max()
    assert trainer.captured.replays == 4
    assert 5. < float(d_big / d_small) < 20., (float(d_small), float(d_big))


def test_cache_policy(monkeypatch):
    hp, model, trainer, groups = _setup(monkeypatch, True, n_batches=4)
    cap = trainer.captured
    cap.max_graphs = 2
    for s in range(16):                                  # round-robin over 4 keys, 2 slots: two keys replay, two stay eager, no re-capture
        trainer.step(groups[s % 4], 100 + s)
    assert cap.captures == 2 and cap.broken is None, (cap.captures, cap.broken)
    assert cap.replays == 2 * 3 - 0 - 0 and cap.eager_steps == 16 - cap.replays, (cap.replays, cap.eager_steps)
    # prepare(): the set-up call bench.py uses
    cap.max_graphs = 3
    assert cap.prepare(groups[2], 100)
    n = cap.replays
    trainer.step(groups[2], 200)
    assert cap.replays == n + 1
    torch.cuda.synchronize()
    assert torch.isfinite(model.flat_parameters()).all()
