"""Kernel-level checks that the golden/oracle parity tests cannot give: dropout is ON here.
  * forward and backward kernels regenerate the SAME dropout mask (adjoint / directional-derivative identities);
  * the keep rate is 1 - p; masked-out entries are exactly zero and kept entries are scaled by 1/(1-p)."""
import pytest
import torch


@pytest.fixture(autouse=True)
def _rows_past_the_fill_end_are_dropped(monkeypatch):
    ''' dead rows are zero-filled only below dx_fill_end (csrc/dx_common.h); the rows past it are unwritten: see the shim '''
    from tests.util import install_unwritten_shim
    install_unwritten_shim(monkeypatch)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _attn_inputs(B, N, H, E, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B, N, 3 * E, generator=g) * 0.7).to(DEV).to(dtype)
    lens = torch.randint(N // 2, N + 1, (B,), generator=g)
    lens[0] = N
    return qkv, lens.to(DEV)


@pytest.mark.parametrize('H', [8, 2])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_dropout_mask_is_shared_by_forward_and_backward(H, dtype):
    from daft_exprt import ops
    B, N, E, p, seed = 3, 150, 128, 0.3, 12345
    torch.manual_seed(4321)                               # (d_o below: the identities are sums with cancellation, keep the draw fixed)
    qkv, lens = _attn_inputs(B, N, H, E, 1, dtype)
    o, lse = ops.attention_fwd(qkv, lens, H, p, seed)
    valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None]).unsqueeze(2)
    d_o = (torch.randn(B, N, E, device=DEV) * valid).to(dtype)
    dqkv = ops.attention_bwd(qkv, o, d_o, lse, lens, H, p, seed).float()
    tol = 2e-3 if dtype == torch.float32 else 1e-1
    # (1) O is linear in V for a fixed mask:  <dO, O(V)> == <dV, V>
    lhs = float((d_o.float() * o.float() * valid).sum())
    rhs = float((dqkv[:, :, 2 * E:] * qkv[:, :, 2 * E:].float()).sum())
    assert abs(lhs - rhs) <= tol * max(abs(lhs), 1.), (lhs, rhs)
    # (2) directional derivative wrt q and k (central differences, same seed -> same mask)
    if dtype == torch.float32:
        g = torch.Generator().manual_seed(2)
        d = torch.zeros_like(qkv)
        d[:, :, : 2 * E] = torch.randn(B, N, 2 * E, generator=g).to(DEV)
        eps = 2e-3
        f = lambda t: float((ops.attention_fwd(t, lens, H, p, seed)[0].double() * d_o.double() * valid).sum())
        num = (f(qkv + eps * d) - f(qkv - eps * d)) / (2 * eps)
        ana = float((dqkv.double() * d.double()).sum())
        assert abs(num - ana) <= 2e-2 * max(abs(ana), 1.), (num, ana)
    # (3) a different seed gives a different output; p = 0 gives the deterministic one
    o2, _ = ops.attention_fwd(qkv, lens, H, p, seed + 1)
    assert float((o2.float() - o.float()).abs().max()) > 1e-3


@pytest.mark.parametrize('H', [2, 8])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_launch_order_does_not_change_results(H, dtype):
    ''' dx_length_order: a permutation by decreasing length (ties: lower index first); attention launched in that order
        returns the bits of the identity order, forward and backward '''
    from daft_exprt import ops
    B, N, E, p, seed = 7, 300, 128, 0.1, 99
    qkv, _ = _attn_inputs(B, N, H, E, 5, dtype)
    lens = torch.tensor([120, 300, 1, 64, 300, 257, 129], device=DEV)
    order = ops.length_order(lens)
    assert order.dtype == torch.int32 and order.tolist() == [1, 4, 5, 6, 0, 3, 2]
    o0, lse0 = ops.attention_fwd(qkv, lens, H, p, seed)
    o1, lse1 = ops.attention_fwd(qkv, lens, H, p, seed, order=order)
    valid = (torch.arange(N, device=DEV)[None, :] < lens[:, None])
    assert torch.equal(o0[valid], o1[valid]) and torch.equal(lse0.transpose(1, 2)[valid], lse1.transpose(1, 2)[valid])
    d_o = (torch.randn(B, N, E, device=DEV) * valid.unsqueeze(2)).to(dtype)
    g0 = ops.attention_bwd(qkv, o0, d_o, lse0, lens, H, p, seed)
    g1 = ops.attention_bwd(qkv, o0, d_o, lse0, lens, H, p, seed, order=order)
    assert torch.equal(g0[valid], g1[valid])


def test_attention_dropout_keep_rate():
    ''' with v = 1 every output equals sum_j P_drop[i, j] = (kept probability mass) / (1 - p): its mean over queries is 1 '''
    from daft_exprt import ops
    B, N, H, E, p = 4, 512, 8, 128, 0.1
    qkv, lens = _attn_inputs(B, N, H, E, 3, torch.float32)
    qkv[:, :, : 2 * E] = 0.          # uniform attention: P = 1/len for every valid key
    qkv[:, :, 2 * E:] = 1.
    lens[:] = N
    o, _ = ops.attention_fwd(qkv, lens, H, p, 777)
    m = float(o.mean())
    assert abs(m - 1.) < 5e-3, m          # E[keep]/(1-p) = 1; std of a 512-key average ~ 0.015 / sqrt(#queries)
    per_q = o[:, :, 0]
    assert float(per_q.std()) > 1e-3      # dropout really happened


def _attn_ref_grads(qkv, d_o, lens, H):
    ''' fp32 torch autograd of the attention core (no dropout), pad keys masked, pad queries excluded '''
    B, N, E3 = qkv.shape
    E, dh = E3 // 3, E3 // 3 // H
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = [t.reshape(B, N, H, dh).transpose(1, 2) for t in x.split(E, dim=2)]
    s = (q @ k.transpose(2, 3)) / dh ** 0.5
    pad = torch.arange(N, device=qkv.device)[None, :] >= lens[:, None]
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    s = s.masked_fill((lens == 0)[:, None, None, None], 0.)          # a zero-length utterance: keep its (unused) softmax finite
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, N, E)
    (o * d_o.float() * (~pad).unsqueeze(2)).sum().backward()
    return x.grad * (~pad).unsqueeze(2)


@pytest.mark.parametrize('N,lens', [(150, [150, 97, 33]), (1000, [1000, 624, 31, 257]), (1024, [1024, 1, 512]), (70, [0, 70, 64])])
def test_attention_fused_backward_matches_two_pass_and_fp32_reference(N, lens):
    ''' the fused d_head = 16 backward kernel (one recomputation of S / dP for dQ, dK, dV; keys split over the waves, queries
        streamed, dQ partials summed in wave order) against (a) the two-pass kernels on the same dropout mask: same bf16 rounding
        points, so only fp32 summation order differs; (b) torch fp32 autograd without dropout '''
    from daft_exprt import ops
    B, H, E = len(lens), 8, 128
    g = torch.Generator().manual_seed(N)
    qkv = (torch.randn(B, N, 3 * E, generator=g) * 0.7).to(DEV).to(torch.bfloat16)
    lens_t = torch.tensor(lens, device=DEV)
    valid = (torch.arange(N, device=DEV)[None, :] < lens_t[:, None]).unsqueeze(2)
    d_o = (torch.randn(B, N, E, generator=g).to(DEV) * valid).to(torch.bfloat16)
    for p, seed in ((0.1, 4242), (0., 0)):
        o, lse = ops.attention_fwd(qkv, lens_t, H, p, seed)
        two = ops.attention_bwd(qkv, o, d_o, lse, lens_t, H, p, seed, algo=ops.ATTN_TWO_PASS).float()
        fus = ops.attention_bwd(qkv, o, d_o, lse, lens_t, H, p, seed, algo=ops.ATTN_FUSED).float()
        assert torch.isfinite(fus).all()
        assert float((fus * ~valid).abs().max()) == 0.          # pad rows are exact zeros
        scale = float(two.abs().max())
        err = float((fus - two).abs().max())
        assert err <= 1.5e-2 * scale, (p, err, scale)            # a handful of 1-ulp bf16 flips of the stored gradients
        assert float((fus - two).abs().mean()) <= 2e-4 * scale
        again = ops.attention_bwd(qkv, o, d_o, lse, lens_t, H, p, seed, algo=ops.ATTN_FUSED).float()
        assert torch.equal(again, fus)                           # no atomics: bit-reproducible
        order = ops.length_order(lens_t)
        assert torch.equal(ops.attention_bwd(qkv, o, d_o, lse, lens_t, H, p, seed, order=order, algo=ops.ATTN_FUSED).float(), fus)
    ref = _attn_ref_grads(qkv, d_o, lens_t, H)
    scale = float(ref.abs().max())
    assert float((fus - ref).abs().max()) <= 3e-2 * scale and float((fus - ref).abs().mean()) <= 2e-3 * scale


def test_attention_fused_backward_is_refused_where_it_does_not_apply():
    from daft_exprt import ops
    qkv, lens = _attn_inputs(2, 64, 2, 128, 1, torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, lens, 2, 0., 0)
    with pytest.raises(RuntimeError, match='fused kernel'):
        ops.attention_bwd(qkv, o, torch.zeros_like(o), lse, lens, 2, algo=ops.ATTN_FUSED)


@pytest.mark.parametrize('C', [128, 256, 1024])
def test_layernorm_dropout_forward_backward_consistency(C):
    from daft_exprt import ops
    B, N = 3, 70
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, N, C, generator=g).to(DEV)
    res = torch.randn(B, N, C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    film = torch.randn(B, 2 * C, generator=g).to(DEV)
    lens = torch.tensor([70, 33, 51], device=DEV)
    kw = dict(p_pre=0.2, seed_pre=11, p_post=0.3, seed_post=22)
    run = lambda xx: ops.layernorm_fwd(xx, gamma, beta, residual=res, film=film, lengths=lens, save=True, save_s=True, **kw)
    y, s, mean, rstd = run(x)
    dy = torch.randn(B, N, C, generator=g).to(DEV)
    dgam, dbet, dfilm = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(B, 2 * C, device=DEV)
    ds, dx = ops.layernorm_bwd(dy, s, mean, rstd, gamma, beta, dgam, dbet, film=film, dfilm=dfilm, lengths=lens, **kw)
    d = torch.randn(B, N, C, generator=g).to(DEV)
    eps = 1e-2
    f = lambda t: float((run(t)[0].double() * dy.double()).sum())
    num = (f(x + eps * d) - f(x - eps * d)) / (2 * eps)
    ana = float((dx.double() * d.double()).sum())
    assert abs(num - ana) <= 2e-2 * max(abs(ana), 1.), (num, ana)
    # post-dropout keep rate on the valid rows (FiLM beta makes dropped entries equal film_beta, so test without FiLM)
    y2, _, _, _ = ops.layernorm_fwd(x, gamma, beta, p_post=0.3, seed_post=5)
    zeros = float((y2 == 0).float().mean())
    assert abs(zeros - 0.3) < 0.02, zeros


def test_wgrad_matches_autograd_all_dtype_pairs():
    from daft_exprt import ops
    from oracle import daft_exprt_cpu as O
    g = torch.Generator().manual_seed(9)
    B, N, Cin, Cout = 5, 300, 128, 256
    lens = torch.tensor([300, 120, 255, 7, 64])
    x = torch.randn(B, N, Cin, generator=g)
    dy = torch.randn(B, N, Cout, generator=g) * (torch.arange(N)[None, :, None] < (lens[:, None, None] + 2))
    for taps in (1, 3):
        w = (torch.randn(Cout, Cin, taps, generator=g) / 10).requires_grad_(True)
        b = torch.zeros(Cout, requires_grad=True)
        (O.conv1d_cl(x, w, b) * dy).sum().backward()
        for cd, dyt, xt, tol in ((torch.float32, torch.float32, torch.float32, 2e-5), (torch.bfloat16, torch.bfloat16, torch.float32, 2e-2),
                                 (torch.bfloat16, torch.float32, torch.bfloat16, 2e-2), (torch.bfloat16, torch.bfloat16, torch.bfloat16, 2e-2)):
            for use_ws in (True, False):                      # partial tiles + fixed-order reduce | fp32 atomics
                ops.WGRAD_WORKSPACE = use_ws
                dw = torch.zeros(Cout, Cin, taps, device=DEV) if taps == 3 else torch.zeros(Cout, Cin, device=DEV)
                db = torch.zeros(Cout, device=DEV)
                ops.conv1d_wgrad(dy.to(DEV).to(dyt), x.to(DEV).to(xt), dw, db, cd, lens.to(DEV))
                ref = w.grad if taps == 3 else w.grad[:, :, 0]
                assert float((dw.cpu() - ref).abs().max()) <= tol * float(ref.abs().max()), (taps, cd, dyt, xt, use_ws)
                assert float((db.cpu() - b.grad).abs().max()) <= tol * float(b.grad.abs().max()) + 1e-4, (taps, cd, use_ws)
            ops.WGRAD_WORKSPACE = True
        w.grad = None


def test_wgrad_ragged_shapes_accumulate_and_fixed_order():
    ''' channel counts that do not fill a tile, N not a multiple of the 128-position item, no lengths, dw += semantics,
        and bit-identical results run to run on the workspace path '''
    from daft_exprt import ops
    from oracle import daft_exprt_cpu as O
    g = torch.Generator().manual_seed(10)
    for (B, N, Cin, Cout, taps, lens) in ((3, 333, 80, 200, 3, None), (2, 129, 136, 72, 1, None), (7, 1, 128, 128, 3, None),
                                          (4, 515, 128, 384, 1, torch.tensor([515, 0, 130, 126]))):
        x = torch.randn(B, N, Cin, generator=g)
        dy = torch.randn(B, N, Cout, generator=g)
        if lens is not None:
            dy = dy * (torch.arange(N)[None, :, None] < (lens[:, None, None] + 2))
        w = (torch.randn(Cout, Cin, taps, generator=g) / 10).requires_grad_(True)
        b = torch.zeros(Cout, requires_grad=True)
        (O.conv1d_cl(x, w, b) * dy).sum().backward()
        ref = w.grad if taps == 3 else w.grad[:, :, 0]
        init = torch.randn(ref.shape, generator=g)
        outs = []
        for _ in range(2):
            dw, db = init.clone().to(DEV), torch.zeros(Cout, device=DEV)
            ops.conv1d_wgrad(dy.to(DEV), x.to(DEV), dw, db, torch.float32, lens.to(DEV) if lens is not None else None)
            outs.append(dw.cpu())
        scale = float(ref.abs().max())
        assert float((outs[0] - init - ref).abs().max()) <= 3e-5 * scale + 1e-5, (B, N, Cin, Cout, taps)
        assert torch.equal(outs[0], outs[1])
        assert float((db.cpu() - b.grad).abs().max()) <= 3e-5 * float(b.grad.abs().max()) + 1e-4


def test_wgrad_ring_kernel_ragged_bf16():
    ''' the LDS-DMA ring weight-gradient kernel (bf16 operands, k = 3): ragged lengths incl. an empty utterance, channel
        counts that do not fill a 128 x 128 tile, N not a multiple of the 64-row item, sliced (strided) operands, bias
        gradient from the loader waves, run-to-run reproducibility of dW; against autograd on the bf16-rounded operands '''
    from daft_exprt import ops
    from oracle import daft_exprt_cpu as O
    g = torch.Generator().manual_seed(21)
    for (B, N, Cin, Cout, lens) in ((5, 333, 136, 200, [333, 0, 64, 65, 200]), (3, 700, 1024, 128, [700, 31, 450]), (2, 70, 128, 384, None)):
        xw = torch.randn(B, N, Cin + 8, generator=g).to(torch.bfloat16)
        x = xw[:, :, 8:]                                             # row stride != channel count
        dy = torch.randn(B, N, Cout, generator=g)
        lt = torch.tensor(lens) if lens is not None else None
        if lt is not None:
            dy = dy * (torch.arange(N)[None, :, None] < lt[:, None, None])
        dy = dy.to(torch.bfloat16)
        w = (torch.randn(Cout, Cin, 3, generator=g) / 10).requires_grad_(True)
        b = torch.zeros(Cout, requires_grad=True)
        (O.conv1d_cl(x.float(), w, b) * dy.float()).sum().backward()
        outs = []
        for _ in range(2):
            dw, db = torch.zeros(Cout, Cin, 3, device=DEV), torch.zeros(Cout, device=DEV)
            ops.conv1d_wgrad(dy.to(DEV), xw.to(DEV)[:, :, 8:], dw, db, torch.bfloat16, lt.to(DEV) if lt is not None else None)
            outs.append(dw.cpu())
        scale = float(w.grad.abs().max())
        assert float((outs[0] - w.grad).abs().max()) <= 2e-3 * scale, (B, N, Cin, Cout, float((outs[0] - w.grad).abs().max()), scale)
        assert torch.equal(outs[0], outs[1])
        assert float((db.cpu() - b.grad).abs().max()) <= 2e-3 * float(b.grad.abs().max()) + 1e-3


@pytest.mark.parametrize('taps,cin,film,p', [(3, 1024, False, 0.1), (1, 384, True, 0.2), (3, 128, True, 0.), (1, 128, False, 0.)])
def test_conv_lnbwd_fusion_matches_two_launches(taps, cin, film, p):
    ''' dx_conv1d_lnbwd == dx_conv1d(ACCUMULATE) followed by dx_layernorm_bwd (same dropout counter stream), including
        ragged lengths, dead tiles past length + 2, FiLM gradients and the per-channel reductions '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(taps * 100 + cin)
    B, N = 4, 333
    lens = torch.tensor([333, 120, 63, 1]).to(DEV)
    dyin = torch.randn(B, N, cin, generator=g).to(DEV)
    w = (torch.randn(cin, 128, taps, generator=g) / (cin * taps) ** 0.5).to(DEV)     # forward conv 128 -> cin
    n_idx = torch.arange(N, device=DEV)[None, :, None]
    res = torch.randn(B, N, 128, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)   # zero past length + halo, as upstream
    s_in = torch.randn(B, N, 128, generator=g).to(DEV)
    gamma, beta = torch.randn(128, generator=g).to(DEV), torch.randn(128, generator=g).to(DEV)
    fl = torch.randn(B, 256, generator=g).to(DEV) if film else None
    mean = s_in.mean(-1).reshape(-1).contiguous()
    rstd = (1. / torch.sqrt(s_in.var(-1, unbiased=False) + 1e-5)).reshape(-1).contiguous()
    for cd, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        wp = ops.pack_conv_weight(w, cd, transpose_flip=True)
        x = dyin.to(cd) * (n_idx < lens[:, None, None] + 2)
        # two launches
        y0 = res.clone()
        ops.conv1d(x, wp, None, out=y0, accumulate=True, skip_lengths=lens)
        dg0, db0 = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
        df0 = torch.zeros(B, 256, device=DEV) if film else None
        ds0, dx0 = ops.layernorm_bwd(y0, s_in, mean, rstd, gamma, beta, dg0, db0, film=fl, dfilm=df0, lengths=lens, p_pre=p, seed_pre=77,
                                     skip_lengths=lens, lp_only=True)
        # one launch
        y1 = res.clone()
        dg1, db1 = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
        df1 = torch.zeros(B, 256, device=DEV) if film else None
        dx1 = ops.conv1d_lnbwd(x, wp, y1, s_in, mean, rstd, gamma, beta, lens, dg1, db1, film=fl, dfilm=df1, p_pre=p, seed_pre=77)
        sc = float(ds0.abs().max())
        assert float((y1 - ds0).abs().max()) <= tol * sc, (cd, float((y1 - ds0).abs().max()), sc)
        assert float((dx1.float() - dx0.float()).abs().max()) <= max(tol, 1e-2) * sc   # bf16 output rounding
        for a, b_ in ((dg1, dg0), (db1, db0)) + (((df1, df0),) if film else ()):
            assert float((a - b_).abs().max()) <= max(tol, 1e-4) * float(b_.abs().max()) + 1e-4


def _plan_rows(table):
    ''' {b: sorted [(n0, rows)]} of a tile plan '''
    out = {}
    for b, n0, rows, _ in table.cpu().tolist():
        if rows > 0:
            out.setdefault(b, []).append((n0, rows))
    return {b: sorted(v) for b, v in out.items()}


@pytest.mark.parametrize('lens_list,N', [([1000, 517, 300, 129, 64, 1, 999, 730], 1000), ([257, 256, 255, 3], 257), ([40, 33], 40)])
def test_conv_tile_plan_covers_every_row_once(lens_list, N):
    ''' dx_conv_tile_plan: a multiple of 256 tiles, every valid row of every utterance in exactly one tile, tiles <= 256 rows and
        equal (+-1 row per piece) inside an utterance '''
    from daft_exprt import ops
    lens = torch.tensor(lens_list, device=DEV)
    table, B, n = ops.conv_tile_plan(lens, N)
    torch.cuda.synchronize()
    assert table.shape[0] % 256 == 0 and table.shape[0] >= len(lens_list) * ((N + 255) // 256)
    rows = _plan_rows(table)
    for b, ln in enumerate(lens_list):
        pos = 0
        for n0, r in rows[b]:
            assert n0 == pos and 0 < r <= 256
            pos += r
        assert pos == ln
        hs = [r for _, r in rows[b]]
        assert max(hs) - min(hs) <= len(hs)       # equal pieces (the last one takes the remainder)
    tall = max(r for v in rows.values() for _, r in v)
    # no shorter maximum height fits the tile budget
    assert tall == 1 or sum(-(-ln // (tall - 1)) for ln in lens_list) > table.shape[0]


@pytest.mark.parametrize('film', [False, True])
def test_planned_conv_ln_and_lnbwd_match_unplanned(film):
    ''' the balanced-tile launches (256-row tiles from dx_conv_tile_plan, LDS-DMA ring, padding-fill workgroups) give the
        results of the fixed-tile launches: outputs bit-identical on valid rows (same summation order), zeros on padding
        rows, per-channel reductions equal up to the order of the atomics '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(11)
    B, N, cin = 6, 700, 1024
    lens = torch.tensor([700, 433, 257, 256, 130, 5]).to(DEV)
    n_idx = torch.arange(N, device=DEV)[None, :, None]
    valid = n_idx < lens[:, None, None]
    x = (torch.randn(B, N, cin, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)).to(torch.bfloat16)
    w = (torch.randn(128, cin, 3, generator=g) / (cin * 3) ** 0.5).to(DEV)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    bias, gamma, beta = (torch.randn(128, generator=g).to(DEV) for _ in range(3))
    res = torch.randn(B, N, 128, generator=g).to(DEV) * valid
    fl = torch.randn(B, 256, generator=g).to(DEV) if film else None
    plan = ops.conv_tile_plan(lens, N)
    outs = []
    for pl in (None, plan):
        outs.append(ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=pl))
    for a, b_, name in zip(outs[0], outs[1], ('y', 'y_lp', 's', 'mean', 'rstd')):
        m = valid if a.dim() == 3 else valid.reshape(-1)
        a, b_ = a.float(), b_.float()
        assert torch.equal(a * m, b_ * m), (name, float(((a - b_) * m).abs().max()))
        assert float((b_ * ~m).abs().max()) == 0., name           # padding rows of the planned launch are zeros
    # backward
    y_ln, _, s_in, mean, rstd = outs[0]
    wpt = ops.pack_conv_weight((torch.randn(cin, 128, 3, generator=g) / (cin * 3) ** 0.5).to(DEV), torch.bfloat16, transpose_flip=True)
    gin = torch.randn(B, N, 128, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)
    r = []
    for pl in (None, plan):
        y = gin.clone()
        dg, db = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
        df = torch.zeros(B, 256, device=DEV) if film else None
        dx = ops.conv1d_lnbwd(x, wpt, y, s_in, mean, rstd, gamma, beta, lens, dg, db, film=fl, dfilm=df, p_pre=0.1, seed_pre=4, plan=pl)
        r.append((y, dx.float(), dg, db, df))
    assert torch.equal(r[0][0] * valid, r[1][0] * valid) and torch.equal(r[0][1] * valid, r[1][1] * valid)
    assert float((r[1][0] * ~valid).abs().max()) == 0. and float((r[1][1] * ~valid).abs().max()) == 0.
    for k in (2, 3) + ((4,) if film else ()):
        assert float((r[0][k] - r[1][k]).abs().max()) <= 1e-4 * float(r[0][k].abs().max()) + 1e-5, k
    # k = 1 backward variant (QKV data gradient, K = 384) on the same plan
    xq = (torch.randn(B, N, 384, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)).to(torch.bfloat16)
    wq = ops.pack_conv_weight((torch.randn(384, 128, generator=g) / 384 ** 0.5).to(DEV), torch.bfloat16, transpose_flip=True)
    r = []
    for pl in (None, plan):
        y = gin.clone()
        dg, db = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
        df = torch.zeros(B, 256, device=DEV) if film else None
        dx = ops.conv1d_lnbwd(xq, wq, y, s_in, mean, rstd, gamma, beta, lens, dg, db, film=fl, dfilm=df, p_pre=0.1, seed_pre=5, plan=pl)
        r.append((y, dx.float(), dg, db, df))
    assert torch.equal(r[0][0] * valid, r[1][0] * valid) and torch.equal(r[0][1] * valid, r[1][1] * valid)
    assert float((r[1][0] * ~valid).abs().max()) == 0. and float((r[1][1] * ~valid).abs().max()) == 0.
    for k in (2, 3) + ((4,) if film else ()):
        assert float((r[0][k] - r[1][k]).abs().max()) <= 1e-4 * float(r[0][k].abs().max()) + 1e-5, k


@pytest.mark.parametrize('film', [False, True])
@pytest.mark.parametrize('lens_list,N', [([700, 433, 257, 256, 130, 5], 700), ([1000, 999, 31, 1, 0, 640, 512, 300], 1000), ([40, 17], 40)])
def test_splitk_conv_ln_and_lnbwd_match_ring_kernel_and_fp32_reference(film, lens_list, N):
    ''' the split-K kernel (fragment-order weights from L2 into registers, the two K halves of a workgroup added through LDS)
        on the balanced tiles of the ring kernel: same inputs, same dropout counters -> results equal up to the fp32 summation order
        of the contraction (two halves instead of one chain); padding rows exact zeros; bit-reproducible; and the conv itself
        against a torch fp32 reference on the same bf16-rounded operands '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(N + len(lens_list))
    B, cin = len(lens_list), 1024
    lens = torch.tensor(lens_list).to(DEV)
    n_idx = torch.arange(N, device=DEV)[None, :, None]
    valid = n_idx < lens[:, None, None]
    x = (torch.randn(B, N, cin, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)).to(torch.bfloat16)
    w = (torch.randn(128, cin, 3, generator=g) / (cin * 3) ** 0.5).to(DEV)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    wf = ops.pack_frag_major(wp)
    bias, gamma, beta = (torch.randn(128, generator=g).to(DEV) for _ in range(3))
    res = torch.randn(B, N, 128, generator=g).to(DEV) * valid
    fl = torch.randn(B, 256, generator=g).to(DEV) if film else None
    plan = ops.conv_tile_plan(lens, N)
    ring = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=plan)
    sk = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=plan, w_frag=wf)
    sk2 = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=plan, w_frag=wf)
    for a, b_, c_, name in zip(ring, sk, sk2, ('y', 'y_lp', 's', 'mean', 'rstd')):
        m = valid if a.dim() == 3 else valid.reshape(-1)
        a, b_, c_ = a.float(), b_.float(), c_.float()
        assert torch.isfinite(b_).all(), name
        assert torch.equal(b_, c_), name                                    # fixed summation order: reproducible
        assert float((b_ * ~m).abs().max()) == 0., name                     # padding rows are zeros
        tol = 2e-2 if name == 'y_lp' else 2e-4
        assert float(((a - b_) * m).abs().max()) <= tol * max(1., float((a * m).abs().max())), (name, float(((a - b_) * m).abs().max()))
    # second GEMM in the forward epilogue (the next block's QKV projection): y2 = y_lp . w2^T + b2 from the same launch == dx_conv1d
    # on the launch's own y_lp; the five regular outputs unchanged bit for bit; padding rows of y2 zero
    for n2 in (384, 128):
        w2 = ops.pack_conv_weight((torch.randn(n2, 128, generator=g) / 128 ** 0.5).to(DEV), torch.bfloat16)
        b2 = torch.randn(n2, generator=g).to(DEV)
        out6 = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=plan, w_frag=wf,
                             w2_packed=w2, b2=b2)
        assert len(out6) == 6 and out6[5] is not None and out6[5].shape == (B, N, n2)
        for a, b_ in zip(out6[:5], sk):
            assert torch.equal(a, b_)
        ref2 = ops.conv1d(out6[1], w2, b2, out_dtype=torch.bfloat16, skip_lengths=lens).float()
        y2 = out6[5].float()
        assert torch.isfinite(y2).all()
        live = n_idx < lens[:, None, None] + 2           # (rows in [len, len + 2) of a live tile: bias, like dx_conv1d; dead rows: zeros)
        assert float(((y2 - ref2) * valid).abs().max()) <= 1e-2 * float(ref2.abs().max())
        assert float((y2 * ~live).abs().max()) <= float(b2.abs().max()) + 1e-3
    assert ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, p_pre=0.1, seed_pre=9, lp_copy=True, plan=plan, w2_packed=w2,
                         b2=b2)[5] is None               # no fragment-order weights -> ring kernel: the caller launches the projection
    # s = dropout(conv) + residual with p = 0: the plain conv against torch on the bf16-rounded operands
    s0 = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, film=fl, save=True, lp_copy=True, plan=plan, w_frag=wf)[2]
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.to(torch.bfloat16).float(), bias, padding=1).transpose(1, 2) + res
    assert float(((s0 - ref) * valid).abs().max()) <= 2e-3 * float(ref.abs().max())
    # backward variant
    _, _, s_in, mean, rstd = ring
    wpt = ops.pack_conv_weight((torch.randn(cin, 128, 3, generator=g) / (cin * 3) ** 0.5).to(DEV), torch.bfloat16, transpose_flip=True)
    wft = ops.pack_frag_major(wpt)
    gin = torch.randn(B, N, 128, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)
    r = []
    for frag in (None, wft, wft):
        y = gin.clone()
        dg, db = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
        df = torch.zeros(B, 256, device=DEV) if film else None
        dx = ops.conv1d_lnbwd(x, wpt, y, s_in, mean, rstd, gamma, beta, lens, dg, db, film=fl, dfilm=df, p_pre=0.1, seed_pre=4, plan=plan, w_frag=frag)
        r.append((y, dx.float(), dg, db, df))
    assert torch.equal(r[1][0], r[2][0]) and torch.equal(r[1][1], r[2][1])
    assert float((r[1][0] * ~valid).abs().max()) == 0. and float((r[1][1] * ~valid).abs().max()) == 0.
    for k, tol in ((0, 2e-4), (1, 2e-2)):
        assert float(((r[0][k] - r[1][k]) * valid).abs().max()) <= tol * max(1., float((r[0][k] * valid).abs().max())), k
    for k in (2, 3) + ((4,) if film else ()):
        assert float((r[0][k] - r[1][k]).abs().max()) <= 1e-3 * float(r[0][k].abs().max()) + 1e-4, k
    # second GEMM in the epilogue (y2 = dx . w2^T, the output-projection data gradient): the launch's own dx through dx_conv1d,
    # bit for bit (same bf16 operands, one 128-term fp32 MFMA chain per output either way), zeros on the padding rows; nothing
    # else of the launch changes
    w2 = ops.pack_conv_weight((torch.randn(128, 128, generator=g) / 128 ** 0.5).to(DEV), torch.bfloat16, transpose_flip=True)
    y = gin.clone()
    dg, db = torch.zeros(128, device=DEV), torch.zeros(128, device=DEV)
    df = torch.zeros(B, 256, device=DEV) if film else None
    y2_buf = torch.full((B, N, 128), float('nan'), dtype=torch.bfloat16, device=DEV)   # (the caching allocator hands the kernel whatever is there)
    del y2_buf
    dx2, y2 = ops.conv1d_lnbwd(x, wpt, y, s_in, mean, rstd, gamma, beta, lens, dg, db, film=fl, dfilm=df, p_pre=0.1, seed_pre=4, plan=plan,
                               w_frag=wft, w2_packed=w2)
    assert torch.equal(y, r[1][0]) and torch.equal(dx2.float(), r[1][1])
    ref2 = ops.conv1d(dx2, w2, None, out_dtype=torch.bfloat16, skip_lengths=lens)
    assert torch.isfinite(y2.float()).all() and float((y2.float() * ~valid).abs().max()) == 0.
    assert float(((y2.float() - ref2.float()) * valid).abs().max()) <= 1e-2 * float(ref2.float().abs().max())
    dxr, y2r = ops.conv1d_lnbwd(x, wpt, gin.clone(), s_in, mean, rstd, gamma, beta, lens, torch.zeros(128, device=DEV), torch.zeros(128, device=DEV),
                                film=fl, dfilm=torch.zeros(B, 256, device=DEV) if film else None, p_pre=0.1, seed_pre=4, plan=plan, w2_packed=w2)
    assert float(((y2r.float() - ref2.float()) * valid).abs().max()) <= 5e-2 * float(ref2.float().abs().max())   # no fragment copy: separate launch


@pytest.mark.parametrize('B,N,cin', [(1, 5, 256), (300, 40, 128), (70, 130, 1024)])
def test_planned_conv_ln_edge_batches(B, N, cin):
    ''' balanced-tile launches on batches far from the training shape: a single 5-row utterance, more utterances than
        tiles per pass (B > 256, several 64-utterance rounds of the padding-fill scan), zero-length utterances '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(B * 7 + N)
    lens = torch.randint(0, N + 1, (B,), generator=g)
    lens[0] = N
    lens = lens.to(DEV)
    n_idx = torch.arange(N, device=DEV)[None, :, None]
    valid = n_idx < lens[:, None, None]
    x = (torch.randn(B, N, cin, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)).to(torch.bfloat16)
    wp = ops.pack_conv_weight((torch.randn(128, cin, 3, generator=g) / (cin * 3) ** 0.5).to(DEV), torch.bfloat16)
    bias, gamma, beta = (torch.randn(128, generator=g).to(DEV) for _ in range(3))
    res = torch.randn(B, N, 128, generator=g).to(DEV) * valid
    plan = ops.conv_tile_plan(lens, N)
    a = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, save=True, p_pre=0.2, seed_pre=3, lp_copy=True)
    b_ = ops.conv1d_ln(x, wp, bias, res, gamma, beta, lens, save=True, p_pre=0.2, seed_pre=3, lp_copy=True, plan=plan)
    for u, v, name in zip(a, b_, ('y', 'y_lp', 's', 'mean', 'rstd')):
        m = valid if u.dim() == 3 else valid.reshape(-1)
        assert torch.equal(u.float() * m, v.float() * m), name
        assert float((v.float() * ~m).abs().max()) == 0., name


@pytest.mark.parametrize('shape', [(48, 128, 1280, False), (48, 128, 128, True), (48, 128, 11, False), (5, 128, 11, True), (256, 128, 1280, False),
                                   (600, 128, 128, True)])
def test_small_linear_heads_match_fp32_reference(shape):
    ''' dx_linear_small_fwd / _bwd (classifier, FiLM projections; M <= 512 rows take the LDS-tiled kernels, more the generic
        ones) against an fp64 restatement: y = relu?(x W^T + b), dx = scale * (dy * relu') W, dW += (dy * relu')^T x, db += sum '''
    from daft_exprt import ops
    M, K, O, relu = shape
    g = torch.Generator().manual_seed(M + O)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(O, K, generator=g) / K ** 0.5, torch.randn(O, generator=g)
    dy = torch.randn(M, O, generator=g)
    y_ref = x.double() @ w.double().t() + b.double()
    if relu:
        y_ref = y_ref.clamp_min(0.)
    gmask = dy.double() * ((y_ref > 0) if relu else 1.)
    dx_ref, dw_ref, db_ref = -0.5 * gmask @ w.double(), gmask.t() @ x.double(), gmask.sum(0)
    d = lambda t: t.to(DEV)
    y = ops.linear_small_fwd(d(x), d(w), d(b), relu=relu)
    dw, db = torch.ones(O, K, device=DEV), torch.ones(O, device=DEV)          # the kernels ACCUMULATE into dw / db
    dx = ops.linear_small_bwd(d(dy), y if relu else None, d(x), d(w), dw, db, relu=relu, dx_scale=-0.5)
    torch.cuda.synchronize()
    close = lambda a, r, what: float((a.double().cpu() - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-6 or pytest.fail(what)
    close(y, y_ref, 'y'); close(dx, dx_ref, 'dx'); close(dw - 1., dw_ref, 'dw'); close(db - 1., db_ref, 'db')


def test_wgrad_multi_equals_separate_calls():
    ''' dx_conv1d_wgrad_multi (the four weight gradients of an FFT block's backward pass in one call: four GEMM launches, ONE launch that
        adds all partial tiles) == four dx_conv1d_wgrad calls, bit for bit (same partial tiles, same summation order per element);
        dW / db are accumulated into '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(5)
    lens_list = [700, 433, 257, 130, 64, 5, 0, 699]
    B, N = len(lens_list), 700
    lens = torch.tensor(lens_list).to(DEV)
    live = (torch.arange(N, device=DEV)[None, :, None] < lens[:, None, None] + 2)
    mk = lambda c: (torch.randn(B, N, c, generator=g).to(DEV) * live).to(torch.bfloat16)
    shapes = [(128, 1024, 3), (1024, 128, 3), (128, 384, 1), (128, 128, 1)]          # (Cin, Cout, taps)
    items = []
    for cin, cout, taps in shapes:
        dw = torch.randn((cout, cin, taps) if taps == 3 else (cout, cin), generator=g).to(DEV)
        items.append((mk(cout), mk(cin), dw, torch.randn(cout, generator=g).to(DEV)))
    ref = [(dw.clone(), db.clone()) for _, _, dw, db in items]
    for (dy, x, _, _), (dw, db) in zip(items, ref):
        ops.conv1d_wgrad(dy, x, dw, db, torch.bfloat16, lens)
    ops.conv1d_wgrad_multi(items, torch.bfloat16, lens)
    torch.cuda.synchronize()
    for (_, _, dw, db), (rw, rb), sh in zip(items, ref, shapes):
        assert torch.equal(dw, rw), sh
        assert float((db - rb).abs().max()) <= 1e-4 * float(rb.abs().max()), sh       # bias sums: fp32 atomics


def test_loss_partials_form_matches_atomics_form_and_is_reproducible():
    ''' dx_loss_fwd_bwd with a workspace (the training step's form: per-workgroup terms added in a fixed order by the last of three
        launches) == the atomics form (four launches) to fp32 summation order; the workspace form is bit-reproducible; gradients
        identical (the transposed mel gradient is the plain one transposed) '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(3)
    B, L, T, C, S = 7, 33, 301, 80, 10
    in_len = torch.tensor([33, 20, 1, 33, 7, 15, 30]).to(DEV)
    out_len = torch.tensor([301, 150, 3, 64, 65, 200, 300]).to(DEV)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    dur, en, pi, dur_t, en_t, pi_t = (r(B, L) for _ in range(6))
    mel, mel_t = r(B, C, T), r(B, C, T)
    logits, ids, post = r(B, S), torch.randint(0, S, (B,), generator=g).to(DEV), r(2, 9)
    w = (1e-2, 1e-3, 1., 1., 1., 1.)

    def run(transposed):
        gr = {'d_dur': torch.empty_like(dur), 'd_energy': torch.empty_like(en), 'd_pitch': torch.empty_like(pi),
              'd_mel': torch.empty((B, T, C) if transposed else (B, C, T), device=DEV), 'd_spk': torch.empty_like(logits)}
        dpost = torch.zeros_like(post)
        t = ops.loss_fwd_bwd(dur, en, pi, dur_t, en_t, pi_t, in_len, mel, mel_t, out_len, logits, ids, post, w, grads=gr, d_post_mult=dpost,
                             grad_scale=0.5, d_mel_transposed=transposed)
        torch.cuda.synchronize()
        return t.clone(), gr, dpost
    t_ws, g_ws, p_ws = run(True)
    t_ws2, _, _ = run(True)
    t_at, g_at, p_at = run(False)
    assert torch.equal(t_ws, t_ws2)
    assert torch.allclose(t_ws, t_at, rtol=2e-6, atol=1e-7), (t_ws.tolist(), t_at.tolist())
    assert abs(float(t_ws[7]) - float(t_ws[:7].sum())) <= 1e-5 * abs(float(t_ws[7]))
    assert torch.equal(g_ws['d_mel'].transpose(1, 2), g_at['d_mel'])
    for k in ('d_dur', 'd_energy', 'd_pitch', 'd_spk'):
        assert torch.equal(g_ws[k], g_at[k]), k
    assert torch.equal(p_ws, p_at)
