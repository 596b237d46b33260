"""`DaftExprt.overlap_predictor`: in a teacher-forced step the local prosody predictor's forward feeds nothing but the loss and its
backward needs nothing but the loss gradients, so both run on the weight-gradient stream beside upsampling / decoder.  Same kernels,
same arguments, same dropout sites in the same order -- only the stream differs: predictions and loss terms must be bit-equal to
the in-line schedule, gradients equal up to the run-to-run noise of the fp32 atomics (per-channel LayerNorm / FiLM / bias sums), over several steps in a
row (buffers of one step are re-used by the next: a missing cross-stream dependency shows up as garbage sooner or later)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('mode', ['bf16', 'fp32'])
def test_predictor_beside_the_decoder_equals_the_inline_schedule(mode):
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    from tests.util import make_hparams
    hp = make_hparams(compute_dtype=mode, batch_size=6)
    res = []
    for overlap in (True, False):
        torch.manual_seed(3)
        model = DaftExprt(hp).to(DEV).train()
        model.overlap_predictor = overlap
        weights = DaftExprtLoss(0, hp).weights(20000)
        out = []
        for k in range(4):                                   # different batch geometries back to back
            cb = synthetic_batch(hp, 6, seed=40 + k, t_max=300 - 60 * k, force_first_full=True, l_range=(15, 50 + 10 * k))
            inputs, targets, _ = model.parse_batch(DEV, cb)
            model.zero_grad()
            model._step_id = 10 + k
            terms = model.forward_backward(inputs, targets, weights)
            logits, films, (dur, energy, pitch), mel, _ = model.last_outputs
            torch.cuda.synchronize()
            out.append(([t.clone() for t in (mel, dur, energy, pitch, logits)], terms.clone(), model.flat_gradients().clone()))
        res.append(out)
        assert model._predictor_beside() == overlap
    for k, ((p0, t0, g0), (p1, t1, g1)) in enumerate(zip(*res)):
        for a, b in zip(p0, p1):
            assert torch.equal(a, b), (mode, k)
        assert torch.allclose(t0, t1, rtol=1e-6, atol=0.), (mode, k, t0.tolist(), t1.tolist())
        # (the FiLM / LayerNorm / bias gradient sums are fp32 atomics: 1e-7 of run-to-run wobble, which in bf16 mode flips an operand
        #  rounding here and there on its way through the prosody encoder's backward -- two in-line runs differ by as much)
        tol = 2e-3 if mode == 'bf16' else 2e-5
        assert float((g0 - g1).norm()) <= tol * float(g1.norm()), (mode, k, float((g0 - g1).norm()) / float(g1.norm()))
