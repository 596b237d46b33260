"""K2 (dx_ff_fused_fwd): the feed-forward half of an FFT block in one launch against
  (a) the two-launch path it replaces (dx_conv1d + dx_conv1d_ln, same dropout counters -> same mask): every output the backward
      pass or the next block reads -- y, the bf16 copy, the saved LayerNorm input / statistics, the hidden tensor on the rows the
      reference computes (0 .. length inclusive: the first padding row is not masked between the convolutions);
  (b) the CPU oracle's conv_ff with bf16 operand rounding (no dropout).
Ragged batches whose lengths straddle the 126-row tile height, a one-row utterance, FiLM on / off, dropout on / off."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O

DEV = 'cuda:0'


def _setup(lens, N, C, film, seed):
    from daft_exprt import ops
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    a = torch.randn(B, N, 128, generator=g)
    lengths = torch.tensor(lens)
    a = a * (torch.arange(N)[None, :, None] < lengths[:, None, None])      # the attention sub-layer's output is masked
    w1 = torch.randn(C, 128, 3, generator=g) / (128 * 3) ** 0.5
    b1 = torch.randn(C, generator=g) * 0.1
    w2 = torch.randn(128, C, 3, generator=g) / (C * 3) ** 0.5
    b2 = torch.randn(128, generator=g) * 0.1
    gamma, beta = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    fl = torch.randn(B, 256, generator=g) * 0.3 + torch.cat((torch.ones(128), torch.zeros(128))) if film else None
    d = lambda t: None if t is None else t.to(DEV)
    dev = dict(a=d(a), a_lp=d(a).to(torch.bfloat16), lengths=d(lengths), b1=d(b1), b2=d(b2), gamma=d(gamma), beta=d(beta), film=d(fl),
               w1p=ops.pack_conv_weight(d(w1), torch.bfloat16), w2p=ops.pack_conv_weight(d(w2), torch.bfloat16))
    cpu = dict(a=a, lengths=lengths, w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta, film=fl)
    return dev, cpu


CASES = [([300, 127, 126, 125, 1], 300, 1024, True), ([64, 64, 3], 64, 1024, False), ([253, 252, 130, 2], 260, 256, True),
         ([1000, 777, 511, 129], 1000, 1024, True)]


@pytest.mark.parametrize('case', range(len(CASES)))
@pytest.mark.parametrize('p_drop', [0., 0.1])
def test_fused_matches_two_launch_path(case, p_drop):
    from daft_exprt import ops
    lens, N, C, film = CASES[case]
    D, _ = _setup(lens, N, C, film, seed=case)
    seed = 4242
    h_ref = ops.conv1d(D['a_lp'], D['w1p'], D['b1'], out_dtype=torch.bfloat16, relu=True, skip_lengths=D['lengths'])
    ref = ops.conv1d_ln(h_ref, D['w2p'], D['b2'], D['a'], D['gamma'], D['beta'], D['lengths'], film=D['film'], save=True,
                        p_pre=p_drop, seed_pre=seed, lp_copy=True)
    got = ops.ff_fused_fwd(D['a_lp'], D['w1p'], D['b1'], D['w2p'], D['b2'], D['a'], D['gamma'], D['beta'], D['lengths'],
                           ops.ff_plan(D['lengths'], N), film=D['film'], save=True, p_pre=p_drop, seed_pre=seed, lp_copy=True)
    torch.cuda.synchronize()
    y, y_lp, s_out, mean, rstd, h = got
    ry, ry_lp, rs, rmean, rrstd = ref
    # hidden tensor: identical rounding points (bf16 operands, fp32 accumulation over 384 products), other summation order
    for b, l in enumerate(lens):
        top = min(l + 1, N)
        a_, b_ = h[b, :top].float(), h_ref[b, :top].float()
        assert float((a_ - b_).abs().max()) <= 2e-2 * float(b_.abs().max()) + 1e-3, ('h', b)     # a few 1-ulp bf16 flips at most
        assert float(((a_ - b_).abs() > 1e-3 * float(b_.abs().max())).float().mean()) < 2e-3, ('h flips', b)
        tail = h[b, top: min(N, l + 1 + 130)]
        assert not tail.any(), ('h tail not zero', b)
    scale = float(ry.abs().max())
    assert float((y - ry).abs().max()) <= 2e-3 * scale, float((y - ry).abs().max()) / scale
    assert float((y_lp.float() - ry_lp.float()).abs().max()) <= 1e-2 * scale
    B = len(lens)
    valid = (torch.arange(N, device=DEV)[None, :] < D['lengths'][:, None])
    # saved LayerNorm input / statistics: compared on the valid rows (the unplanned two-launch path leaves whatever it computed
    # in the padding rows of its live tiles; the LayerNorm backward never reads them)
    assert float(((s_out - rs) * valid.unsqueeze(2)).abs().max()) <= 2e-3 * float((rs * valid.unsqueeze(2)).abs().max())
    assert float(((mean - rmean).view(B, N) * valid).abs().max()) <= 2e-3 * float((rs * valid.unsqueeze(2)).abs().max())
    rel = ((rstd - rrstd).view(B, N) * valid).abs() / (rrstd.view(B, N).abs() + 1e-12)
    assert float(rel.max()) <= 5e-3
    for b, l in enumerate(lens):     # padded rows exactly zero everywhere
        assert not y[b, l:].any() and not y_lp[b, l:].any() and not s_out[b, l:].any()
        assert not mean.view(B, N)[b, l:].any() and not rstd.view(B, N)[b, l:].any()
    if p_drop > 0.:                  # same counters -> same mask: a differing mask would move y by O(1)
        o2 = ops.ff_fused_fwd(D['a_lp'], D['w1p'], D['b1'], D['w2p'], D['b2'], D['a'], D['gamma'], D['beta'], D['lengths'],
                              ops.ff_plan(D['lengths'], N), film=D['film'], save=False, p_pre=p_drop, seed_pre=seed + 1)[0]
        assert float((o2 - y).abs().max()) > 1e-2 * scale


@pytest.mark.parametrize('case', [0, 2])
def test_fused_matches_oracle_with_bf16_operands(case):
    from daft_exprt import ops
    lens, N, C, film = CASES[case]
    D, Cc = _setup(lens, N, C, film, seed=10 + case)
    y = ops.ff_fused_fwd(D['a_lp'], D['w1p'], D['b1'], D['w2p'], D['b2'], D['a'], D['gamma'], D['beta'], D['lengths'],
                         ops.ff_plan(D['lengths'], N), film=D['film'])[0]
    torch.cuda.synchronize()
    P = {'f.convs.0.conv.weight': Cc['w1'], 'f.convs.0.conv.bias': Cc['b1'], 'f.convs.2.conv.weight': Cc['w2'], 'f.convs.2.conv.bias': Cc['b2'],
         'f.layer_norm.weight': Cc['gamma'], 'f.layer_norm.bias': Cc['beta']}
    O.OPERAND_DTYPE = torch.bfloat16
    try:
        ref = O.conv_ff(P, 'f.', Cc['a'], Cc['film'], 0., False)
    finally:
        O.OPERAND_DTYPE = None
    ref = ref * (torch.arange(N)[None, :, None] < Cc['lengths'][:, None, None])
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 5e-3, err
