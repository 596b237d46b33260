"""Adam fused with the refresh of the MFMA operand copies (`dx_adam_pack_step`) against the two-stage form it replaces: the flat Adam
launch (`dx_adam_step`, the arithmetic of torch.optim.Adam as configured at train.py:299-301, pinned against the reference in
test_gpu_model) followed by the pack launches of the next forward pass.  Same arithmetic per element, so parameters, both moments,
the logged gradient norm (up to the order of the per-workgroup partial sums) and EVERY operand copy -- forward packing, flipped
data-gradient packing and the fragment-order copies of both -- must agree bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _state(mode, fuse, steps=2):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    from tests.util import make_hparams
    hp = make_hparams(compute_dtype=mode, batch_size=4, accumulation_steps=1)
    torch.manual_seed(3)
    model = DaftExprt(hp).to(DEV).train()
    trainer = Trainer(model, hp, 1)
    trainer.captured = None
    trainer.optimizer.fuse_pack = bool(fuse)
    cb = synthetic_batch(hp, 4, seed=21, t_max=150, force_first_full=True, l_range=(6, 20))
    inputs, targets, _ = model.parse_batch(DEV, cb)
    g_fixed = None
    for it in range(steps):
        # identical gradients in both runs (the backward pass has fp32 atomics): run the step's forward / backward, then overwrite the
        # gradient buffer with a fixed pseudo-random one before the optimizer sees it
        model.forward_backward(inputs, targets, trainer.criterion.weights(20000 + it))
        gen = torch.Generator(device=DEV).manual_seed(100 + it)
        model.flat_gradients().copy_(torch.randn(model.n_params, generator=gen, device=DEV) * 1e-2)
        trainer.optimizer.param_groups[0]['lr'] = 1e-3
        if fuse == 'buckets':      # the data-parallel form: one fused launch per gradient bucket, in backward order
            trainer.optimizer.begin_step()
            for _, off, n in trainer.reducer.buckets:
                trainer.optimizer.step_slice(off, n)
            gn = trainer.optimizer.end_step().clone()
        else:
            gn = trainer.optimizer.step().clone()
        model.zero_grad()
    W = model._weights(need_dgrad=True)       # flat path: re-packs here; fused path: everything is already current
    torch.cuda.synchronize()
    return model.flat_parameters().clone(), trainer.optimizer.exp_avg.clone(), trainer.optimizer.exp_avg_sq.clone(), gn, \
        {k: v.clone() for k, v in W.items()}


@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
def test_fused_adam_pack_equals_adam_then_pack(mode):
    p0, m0, v0, gn0, W0 = _state(mode, False)
    p1, m1, v1, gn1, W1 = _state(mode, True)
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert abs(float(gn0) - float(gn1)) <= 1e-5 * float(gn0)
    assert set(W0) == set(W1) and len(W0) >= 54
    kinds = set()
    for k in W0:
        assert torch.equal(W0[k], W1[k]), k
        kinds.add(k.split(':')[0] if ':' in k else 'fwd')
    assert kinds == ({'fwd', 'T', 'F', 'FT'} if mode == 'bf16' else {'fwd', 'T'}), kinds


@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
def test_fused_adam_pack_per_bucket_equals_adam_then_pack(mode):
    ''' the per-bucket form the multi-rank trainer runs behind each all-reduce (`FusedAdam.step_slice`) '''
    p0, m0, v0, gn0, W0 = _state(mode, False)
    p1, m1, v1, gn1, W1 = _state(mode, 'buckets')
    assert torch.equal(p0, p1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert abs(float(gn0) - float(gn1)) <= 1e-5 * float(gn0)
    for k in W0:
        assert torch.equal(W0[k], W1[k]), k


def test_fused_path_skips_the_pack_launches():
    ''' after a fused step `_weights` must find every copy current (no pack launch), after a flat step it must re-pack '''
    from daft_exprt import ops
    calls = []
    orig = ops.pack_weights_batched
    ops.pack_weights_batched = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        _state('bf16', True)
        fused_calls = len(calls)
        del calls[:]
        _state('bf16', False)
        flat_calls = len(calls)
    finally:
        ops.pack_weights_batched = orig
    # step 1 packs in both runs (the copies do not exist before the first forward pass); afterwards only the flat run packs
    assert fused_calls < flat_calls and fused_calls <= 3, (fused_calls, flat_calls)
