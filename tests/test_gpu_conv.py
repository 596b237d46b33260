"""K1/K3/K12 parity: dx_conv1d (MFMA implicit-GEMM conv / linear) vs the CPU oracle's conv1d_cl.
Tolerances: fp32 operands (v_mfma_f32_32x32x2_f32, exact fp32 products) 1e-5 abs + 1e-5 rel of the
row scale; bf16 operands 2e-2 of the output scale (inputs rounded to 8 mantissa bits, fp32 accumulate)."""
import pytest
import torch


@pytest.fixture(autouse=True)
def _rows_past_the_fill_end_are_dropped(monkeypatch):
    ''' dead rows are zero-filled only below dx_fill_end (csrc/dx_common.h); the rows past it are unwritten: see the shim '''
    from tests.util import install_unwritten_shim
    install_unwritten_shim(monkeypatch)

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O


def _run(B, N, Cin, Cout, taps, cdtype, xdtype, ydtype, relu=False, mask=False, trans=False, gate=False, seed=0):
    from daft_exprt import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, Cin, generator=g)
    w = torch.randn(Cout, Cin, taps, generator=g) / (Cin * taps) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    lens = torch.randint(1, N + 1, (B,), generator=g)
    lens[0] = N
    ref = O.conv1d_cl(x, w, bias)
    if relu:
        ref = torch.relu(ref)
    gate_t = torch.randn(B, N, Cout, generator=g) if gate else None
    if gate:
        ref = ref * (gate_t > 0)
    if mask:
        ref = ref * (torch.arange(N)[None, :, None] < lens[:, None, None])
    dev = torch.device('cuda:0')
    xd = x.to(dev).to(xdtype)
    wp = ops.pack_conv_weight(w.to(dev), cdtype)
    gd = None
    if gate:
        gd = gate_t.to(dev).to(ydtype)
        if trans:
            gd = gd.transpose(1, 2).contiguous()
    y = ops.conv1d(xd, wp, bias.to(dev), out_dtype=ydtype, relu=relu, relu_gate=gd,
                   mask_lengths=lens.to(dev) if mask else None, transposed_out=trans)
    torch.cuda.synchronize()
    y = y.float().cpu()
    if trans:
        y = y.transpose(1, 2)
    err = (y - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = 2e-5 * max(scale, 1.) if (cdtype == torch.float32) else 2.5e-2 * scale
    assert err <= tol, f'max err {err:.3e} vs tol {tol:.3e} (scale {scale:.3f})'


@pytest.mark.parametrize('taps', [1, 3])
@pytest.mark.parametrize('shape', [(2, 37, 80, 1024), (3, 130, 128, 384), (2, 257, 1024, 128), (1, 5, 128, 80), (2, 129, 256, 256)])
def test_conv_fp32_exact(shape, taps):
    B, N, Cin, Cout = shape
    _run(B, N, Cin, Cout, taps, torch.float32, torch.float32, torch.float32)


@pytest.mark.parametrize('taps', [1, 3])
@pytest.mark.parametrize('shape', [(2, 37, 80, 1024), (3, 130, 128, 384), (2, 257, 1024, 128), (1, 5, 128, 80)])
def test_conv_bf16(shape, taps):
    B, N, Cin, Cout = shape
    _run(B, N, Cin, Cout, taps, torch.bfloat16, torch.float32, torch.bfloat16)
    _run(B, N, Cin, Cout, taps, torch.bfloat16, torch.bfloat16, torch.float32)


def test_conv_tall_tiles():
    ''' 256-row tiles (MI = 4) of the wide k = 3 GEMMs: forced on a small problem -- ragged N, one utterance shorter than a tile '''
    import os
    import subprocess
    import sys
    code = ("import torch, sys; sys.path.insert(0, 'tests'); import test_gpu_conv as t; "
            "t._run(3, 300, 1024, 256, 3, torch.bfloat16, torch.bfloat16, torch.bfloat16, relu=True, mask=True); "
            "t._run(2, 77, 512, 384, 3, torch.bfloat16, torch.bfloat16, torch.float32, gate=True); print('ok')")
    env = dict(os.environ, DX_CONV_WIDE_MI='4')
    env['PYTHONPATH'] = os.pathsep.join([os.getcwd(), os.path.join(os.getcwd(), 'ubisoft-laforge-daft-exprt_amd'), env.get('PYTHONPATH', '')])
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, cwd=os.getcwd())
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout + r.stderr


def test_conv_epilogues():
    _run(2, 70, 128, 1024, 3, torch.float32, torch.float32, torch.float32, relu=True)
    _run(2, 70, 128, 80, 1, torch.float32, torch.float32, torch.float32, mask=True, trans=True)
    _run(3, 131, 1024, 128, 3, torch.float32, torch.float32, torch.float32, gate=True, mask=True)
    _run(3, 131, 128, 1024, 3, torch.bfloat16, torch.float32, torch.bfloat16, relu=True, gate=True)


def test_pack_transpose_flip_is_conv_data_gradient():
    ''' dX = conv(dY, flipped/transposed W) -- checked against autograd on the oracle conv. '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(5)
    B, N, Cin, Cout = 2, 45, 128, 256
    x = torch.randn(B, N, Cin, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, generator=g) / 20
    dy = torch.randn(B, N, Cout, generator=g)
    (O.conv1d_cl(x, w, None) * dy).sum().backward()
    dev = torch.device('cuda:0')
    wd = ops.pack_conv_weight(w.to(dev), torch.float32, transpose_flip=True)
    dx = ops.conv1d(dy.to(dev), wd, None)
    torch.cuda.synchronize()
    assert (dx.cpu() - x.grad).abs().max().item() < 2e-5 * x.grad.abs().max().item() + 1e-5


DEV = torch.device('cuda:0')


def test_batched_weight_pack_matches_single_pack():
    ''' every brick edge case of dx_pack_conv_weights_batched: channel counts below / not a multiple of the 32-wide brick,
        both tap counts, both layouts, fp32 and bf16 operands '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(80, 128, 1), (200, 80, 3), (8, 8, 3), (1024, 128, 3), (33, 40, 1), (128, 1024, 3)]
    for dtype in (torch.bfloat16, torch.float32):
        ws = [torch.randn((co, ci, t) if t > 1 else (co, ci), generator=g).to(DEV) for co, ci, t in shapes]
        entries, refs = [], []
        for w, (co, ci, t) in zip(ws, shapes):
            for tf in (False, True):
                out = torch.full((t, ci, co) if tf else (t, co, ci), float('nan'), dtype=dtype, device=DEV)
                entries.append((w, out, tf))
                refs.append(ops.pack_conv_weight(w, dtype, transpose_flip=tf))
        table = ops.pack_table(entries, DEV)
        ops.pack_weights_batched(*table, dtype)
        for (w, out, tf), ref in zip(entries, refs):
            assert torch.equal(out, ref), (tuple(w.shape), tf, dtype)


@pytest.mark.parametrize('taps', [1, 3])
@pytest.mark.parametrize('shape', [(3, 300, 512), (1, 77, 256), (5, 128, 1024), (48, 129, 256), (2, 1000, 1024)])
def test_conv_weights_in_registers_kernel(shape, taps):
    ''' the Cin = 128 / Cout % 256 == 0 bf16 path (conv_wreg_kernel): ragged N, single utterance, more workgroups than
        position tiles, ReLU, gate, mask_lengths, and skip_lengths with dead tiles (written as zeros, live rows = full conv) '''
    from daft_exprt import ops
    B, N, Cout = shape
    g = torch.Generator().manual_seed(B * 1000 + N + taps)
    x = torch.randn(B, N, 128, generator=g)
    w = torch.randn(Cout, 128, taps, generator=g) / (128 * taps) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    lens = torch.randint(1, N + 1, (B,), generator=g)
    lens[0] = N if B > 1 else max(1, N // 3)
    gate = torch.randn(B, N, Cout, generator=g)
    xb = x.to(DEV).to(torch.bfloat16)
    full = O.conv1d_cl(xb.float().cpu(), w.to(torch.bfloat16).float(), bias)
    wp = ops.pack_conv_weight(w.to(DEV), torch.bfloat16)
    n_idx = torch.arange(N)[None, :, None]
    cases = [dict(relu=True), dict(gate=True), dict(mask=True, relu=True), dict(skip=True), dict(skip=True, gate=True, out=torch.float32)]
    for c in cases:
        od = c.get('out', torch.bfloat16)
        ref = torch.relu(full) if c.get('relu') else full
        if c.get('gate'):
            ref = ref * (gate > 0)
        if c.get('mask'):
            ref = ref * (n_idx < lens[:, None, None])
        y = ops.conv1d(xb, wp, bias.to(DEV), out_dtype=od, relu=bool(c.get('relu')),
                       relu_gate=gate.to(DEV).to(od) if c.get('gate') else None,
                       mask_lengths=lens.to(DEV) if c.get('mask') else None,
                       skip_lengths=lens.to(DEV) if c.get('skip') else None).float().cpu()
        scale = float(full.abs().max())
        if c.get('skip'):   # tiles that start at or past len + 2 are zeros; everything before them is the full convolution
            dead = (n_idx // 128) * 128 >= (lens[:, None, None] + 2)
            assert float((y * dead).abs().max()) == 0., c
            ref = ref * (~dead)
        tol = (1e-2 if od == torch.bfloat16 else 1e-4) * scale    # operands are pre-rounded: only the output rounding remains
        assert float((y - ref).abs().max()) <= tol, (shape, taps, c, float((y - ref).abs().max()), tol)


@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('lens_list,N', [([700, 433, 257, 256, 130, 5], 700), ([1000, 31, 0, 640], 1000), ([40, 17], 40)])
def test_wide_gemm_kernel_matches_tiled_kernel_and_fp32_reference(relu, lens_list, N):
    ''' dx_conv1d_wide (256 x 256 tiles on the full register file, fragment-order weights from L2 into registers) against
        dx_conv1d's tiled kernel on the same bf16 operands -- fp32 sums in another order: equal within bf16 output rounding -- and
        against torch fp32 on the bf16-rounded operands; rows past length + 2 are zeros; bit-reproducible '''
    from daft_exprt import ops
    g = torch.Generator().manual_seed(N + 7 * len(lens_list))
    B, cin, cout = len(lens_list), 1024, 1024
    lens = torch.tensor(lens_list).to(DEV)
    n_idx = torch.arange(N, device=DEV)[None, :, None]
    x = (torch.randn(B, N, cin, generator=g).to(DEV) * (n_idx < lens[:, None, None] + 2)).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, generator=g) / (cin * 3) ** 0.5).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV) if relu else None
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    wf = ops.pack_frag_major(wp)
    tiled = ops.conv1d(x, wp, bias, out_dtype=torch.bfloat16, relu=relu, skip_lengths=lens).float()
    plan = ops.conv_tile_plan(lens, N, halo=2, round_to=64)
    wide = ops.conv1d(x, wp, bias, out_dtype=torch.bfloat16, relu=relu, skip_lengths=lens, w_frag=wf, wide_plan=plan)
    again = ops.conv1d(x, wp, bias, out_dtype=torch.bfloat16, relu=relu, skip_lengths=lens, w_frag=wf, wide_plan=plan)
    assert wide.dtype == torch.bfloat16 and torch.equal(wide, again)
    wide = wide.float()
    live = n_idx < lens[:, None, None] + 2
    assert float((wide * ~live).abs().max()) == 0.
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w.to(torch.bfloat16).float(), bias, padding=1).transpose(1, 2)
    if relu:
        ref = torch.relu(ref)
    scale = float(ref.abs().max())
    assert float(((wide - ref) * live).abs().max()) <= 1e-2 * scale           # bf16 output rounding of values up to `scale`
    assert float(((wide - ref) * live).abs().mean()) <= 2e-3 * float((ref * live).abs().mean() + 1e-9)
    valid = n_idx < lens[:, None, None]
    assert float(((wide - tiled) * valid).abs().max()) <= 1e-2 * scale


@pytest.mark.gpu
def test_batch_prep_equals_the_three_separate_launches():
    ''' dx_batch_prep: both tile plans and the launch order of one lengths tensor from one launch == dx_conv_tile_plan (halo 0),
        dx_conv_tile_plan (halo 2, tile count a multiple of 64), dx_length_order; switched-off outputs come back as None '''
    from daft_exprt import ops
    for lens, N in (([1000, 3, 0, 517, 256, 255, 999], 1000), ([160, 40, 41, 159, 0, 1], 160), ([70], 70)):
        lt = torch.tensor(lens, device=DEV)
        p0, p2, od = ops.batch_prep(lt, N)
        r0 = ops.conv_tile_plan(lt, N)
        r2 = ops.conv_tile_plan(lt, N, halo=2, round_to=64)
        ro = ops.length_order(lt)
        assert p0[1:] == r0[1:] and p2[1:] == r2[1:]
        assert torch.equal(p0[0], r0[0]) and torch.equal(p2[0], r2[0]) and torch.equal(od, ro)
    a, b, c = ops.batch_prep(lt, N, plan=False, wide=True, order=False)
    assert a is None and c is None and torch.equal(b[0], r2[0])


@pytest.mark.parametrize('B,N,Cout,frag', [(3, 301, 1024, True), (2, 128, 256, False), (5, 1000, 1024, True), (1, 7, 512, False)])
def test_relu_bits_gate_equals_gating_on_the_activation(B, N, Cout, frag):
    ''' dx_conv1d_relu_bits: the ReLU forward that also leaves one bit per element returns the SAME activation as the plain call, and
        the data gradient gated by those bits is bit-equal to the one gated by the activation -- incl. mask_lengths (grouped steps: rows
        at or past the hard sequence end are zero and carry no bit), skip_lengths (dead tiles) and values that round to bf16 zero '''
    from daft_exprt import ops
    from tests.util import drop_unwritten
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(B * 1000 + N)
    x = (torch.randn(B, N, 128, generator=g) * (torch.rand(B, N, 1, generator=g) < 0.9)).to(dev).to(torch.bfloat16)   # some all-zero rows: h = relu(bias)
    w = torch.randn(Cout, 128, 3, generator=g) / 20.
    bias = torch.randn(Cout, generator=g) * 0.05
    bias[::7] = 0.                                                   # with a zero row: exact zeros in front of the ReLU
    lens = torch.randint(1, N + 1, (B,), generator=g)
    lens[0] = N
    lens, nmax = lens.to(dev), torch.clamp(lens + 1, max=N).to(dev)
    wp = ops.pack_conv_weight(w.to(dev), torch.bfloat16)
    wf = ops.pack_frag_major(wp) if frag else None
    args = dict(out_dtype=torch.bfloat16, relu=True, skip_lengths=lens, mask_lengths=nmax, w_frag=wf)
    assert ops.relu_bits_ok(x, wp, torch.bfloat16)
    h0 = ops.conv1d(x, wp, bias.to(dev), **args)
    h1, bits = ops.conv1d(x, wp, bias.to(dev), relu_bits=True, **args)
    torch.cuda.synchronize()
    assert bits.dtype == torch.int32 and tuple(bits.shape) == (B, Cout // 32, N)
    a, b = drop_unwritten(h0.float().cpu(), lens.cpu()), drop_unwritten(h1.float().cpu(), lens.cpu())
    assert torch.equal(a, b)
    live = torch.arange(N)[None, :] < (lens.cpu()[:, None] + 2)
    share = float((a[live] > 0).float().mean())
    assert 0.2 < share < 0.8, share                                  # the gate is neither all open nor all closed
    # data gradient of the 1024 -> 128 partner: dz (B, N, 128) x W2^T -> (B, N, Cout), gated
    dz = torch.randn(B, N, 128, generator=g).to(dev).to(torch.bfloat16)
    w2 = torch.randn(128, Cout, 3, generator=g) / 50.
    w2t = ops.pack_conv_weight(w2.to(dev), torch.bfloat16, transpose_flip=True)
    w2f = ops.pack_frag_major(w2t) if frag else None
    d0 = ops.conv1d(dz, w2t, None, out_dtype=torch.bfloat16, relu_gate=h0, skip_lengths=lens, w_frag=w2f)
    d1 = ops.conv1d(dz, w2t, None, out_dtype=torch.bfloat16, relu_gate=bits, skip_lengths=lens, w_frag=w2f)
    torch.cuda.synchronize()
    a, b = drop_unwritten(d0.float().cpu(), lens.cpu()), drop_unwritten(d1.float().cpu(), lens.cpu())
    assert torch.equal(a[live], b[live]), float((a - b).abs().max())
    assert float(a[live].abs().max()) > 0.


@pytest.mark.parametrize('B,N,film', [(4, 515, True), (48, 1000, False), (3, 130, True)])
def test_ln_fused_gemm_with_rederived_residual_equals_stored_residual(B, N, film):
    ''' dx_conv1d_ln_vres: the FF LayerNorm-fused GEMM that re-derives its residual a = mask(LN1(s1)) from the saved LayerNorm input and
        row statistics of the launch that produced the stream gives what the launch reading the stored fp32 `a` gives (same expression in
        both epilogues: bit-equal), and the producer with store_y = False leaves the same bf16 copy, s1 and statistics '''
    from daft_exprt import ops
    from tests.util import drop_unwritten
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(N)
    bf = torch.bfloat16
    lens = torch.randint(1, N + 1, (B,), generator=g)
    lens[0] = N
    lens = lens.to(dev)
    o = torch.randn(B, N, 128, generator=g).to(dev).to(bf)
    x = torch.randn(B, N, 128, generator=g).to(dev)
    wo = ops.pack_conv_weight((torch.randn(128, 128, generator=g) / 11.).to(dev), bf)
    g1, b1 = (1. + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
    g2, b2 = (1. + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
    bo = (0.1 * torch.randn(128, generator=g)).to(dev)
    kw = dict(save=True, p_pre=0.1, seed_pre=1234, lp_copy=True)
    a, a_lp, s1, m1, r1 = ops.conv1d_ln(o, wo, bo, x, g1, b1, lens, **kw)
    n_, a_lp2, s1b, m1b, r1b = ops.conv1d_ln(o, wo, bo, x, g1, b1, lens, store_y=False, **kw)
    torch.cuda.synchronize()
    assert n_ is None
    for p, q in ((a_lp, a_lp2), (s1, s1b), (m1.view(B, N), m1b.view(B, N)), (r1.view(B, N), r1b.view(B, N))):
        assert torch.equal(drop_unwritten(p.float().cpu(), lens.cpu()), drop_unwritten(q.float().cpu(), lens.cpu()))
    h = torch.relu(torch.randn(B, N, 1024, generator=g)).to(dev).to(bf)
    w2 = ops.pack_conv_weight((torch.randn(128, 1024, 3, generator=g) / 55.).to(dev), bf)
    w2f = ops.pack_frag_major(w2)
    wq = ops.pack_conv_weight((torch.randn(384, 128, generator=g) / 11.).to(dev), bf)
    bq = (0.1 * torch.randn(384, generator=g)).to(dev)
    fl = torch.randn(B, 256, generator=g).to(dev) if film else None
    plan = ops.conv_tile_plan(lens, N)
    assert ops.splitk_ln_ok(B, N, bf, w2, plan, w2f)
    kw2 = dict(film=fl, save=True, p_pre=0.1, seed_pre=77, lp_copy=True, plan=plan, w_frag=w2f, w2_packed=wq, b2=bq)
    ref = ops.conv1d_ln(h, w2, bo, a, g2, b2, lens, **kw2)
    got = ops.conv1d_ln(h, w2, bo, s1, g2, b2, lens, residual_ln=(m1, r1, g1, b1), **kw2)
    torch.cuda.synchronize()
    assert ref[5] is not None and got[5] is not None
    for k, (p, q) in enumerate(zip(ref, got)):
        p, q = (t.view(B, N, -1) if t.dim() == 1 else t for t in (p, q))
        p, q = drop_unwritten(p.float().cpu(), lens.cpu()), drop_unwritten(q.float().cpu(), lens.cpu())
        assert torch.equal(p, q), (k, float((p - q).abs().max()))
