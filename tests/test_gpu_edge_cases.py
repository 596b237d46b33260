"""Edge cases of the padded-batch semantics, HIP (exact-fp32 operand mode) vs the CPU oracle on random weights:
single utterance, one-symbol / few-frame utterances, lengths straddling the 64/128-row tile and 32/256-key stage
boundaries, a batch without any padding, pad extents 0/1/2, and gradients on the same cases."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O
from tests.util import make_hparams, no_dropout, gradient_report

DEV = 'cuda:0'


def _batch(hp, Ls, Ts, seed):
    ''' collated batch with explicit (L_b, T_b); items are passed already sorted by L (descending) '''
    from daft_exprt.data_loader import DaftExprtDataCollate
    rng = np.random.RandomState(seed)
    items = []
    for L, T in zip(Ls, Ts):
        d = np.zeros(L, dtype=np.int64)
        for _ in range(T):
            d[rng.randint(0, L)] += 1
        items.append([torch.from_numpy(rng.randint(1, 76, size=L)), torch.from_numpy((d * 256. / 22050.).astype(np.float32)),
                      torch.from_numpy(d), torch.from_numpy(rng.randn(L).astype(np.float32) * (d > 0)),
                      torch.from_numpy(rng.randn(L).astype(np.float32) * (d > 0)), torch.from_numpy(rng.uniform(0, 3, T).astype(np.float32)),
                      torch.from_numpy((rng.randn(T) * (rng.rand(T) > .3)).astype(np.float32)),
                      torch.from_numpy(np.clip(rng.randn(80, T) * 1.2 - 1., np.log(1e-5), 2.).astype(np.float32)),
                      int(rng.randint(0, 11)), 'd', f'f{len(items)}'])
    return DaftExprtDataCollate(hp)(items)


CASES = {
    'single_utterance': ([7], [23]),
    'one_symbol_few_frames': ([3, 1], [5, 2]),
    'no_padding_at_all': ([6, 6, 6], [40, 40, 40]),
    'pad_extent_1_and_2': ([9, 8, 7], [33, 32, 31]),
    'tile_boundaries': ([70, 65, 64, 63, 2], [257, 129, 128, 127, 65]),
    'long_and_tiny': ([40, 1], [300, 1]),
}


@pytest.mark.parametrize('case', list(CASES))
def test_forward_and_gradients_match_oracle(case):
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    Ls, Ts = CASES[case]
    hp = no_dropout(make_hparams(compute_dtype='fp32'))
    torch.manual_seed(11)
    model = DaftExprt(hp)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    batch = _batch(hp, Ls, Ts, seed=len(Ls) * 100 + Ts[0])
    inputs, targets, _ = model.parse_batch(DEV, batch)
    crit = DaftExprtLoss(0, hp)
    model.zero_grad()
    out = model(inputs)
    loss, _ = crit(out, targets, 20000)
    loss.backward()
    torch.cuda.synchronize()
    cin = tuple(t.cpu() for t in inputs)
    ref = O.forward(P, hp, cin, training=True)
    ref_loss, _ = O.loss(hp, ref, (cin[1], cin[3], cin[4], cin[8], cin[10]), 20000)
    rel = lambda a, b: float((a.detach().cpu().float() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))
    assert rel(out[3][0], ref[3][0]) < 5e-4, ('mel', rel(out[3][0], ref[3][0]))
    assert rel(out[2][0], ref[2][0]) < 5e-4 and rel(out[2][1], ref[2][1]) < 5e-4 and rel(out[2][2], ref[2][2]) < 5e-4
    assert rel(out[4], ref[4]) < 5e-4 and rel(out[0], ref[0]) < 5e-4
    assert abs(float(loss) - float(ref_loss)) < 2e-4 * abs(float(ref_loss))
    grads = torch.autograd.grad(ref_loss, list(P.values()))
    # every one of the 193 gradients ELEMENT-WISE (fp32 operand mode: 3e-3 of a tensor's max + 1e-5 of the model's largest element)
    got = {n: p.grad for n, p in model.named_parameters()}
    worst = gradient_report(got, dict(zip(P.keys(), grads)), 3e-3, 1e-5)
    print(case, 'worst gradient tensors', worst[:4])
    assert worst[0][0] <= 1., worst[:4]


def test_inference_rejects_too_short_utterance_like_the_reference():
    ''' an utterance shorter than one analysis window makes the reference raise IndexError (extract_features.py:104) '''
    from daft_exprt.model import DaftExprt
    hp = make_hparams(compute_dtype='fp32')
    m = DaftExprt(hp).to(DEV).eval()
    d = torch.full((1, 3), 0.005, device=DEV)     # 3 symbols of 5 ms: below the 23 ms threshold -> all zero -> no frames
    with pytest.raises(IndexError):
        m.get_int_durations(d, hp)
