#!/usr/bin/env python
"""Micro-benchmarks of individual HIP kernels on the GPU box (development aid).
usage: python tools/bench_ops.py conv"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd'))
from daft_exprt import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_conv():
    dev = torch.device('cuda:0')
    B, N = int(os.environ.get('BENCH_B', '48')), 1000
    for (cin, cout, taps) in [(80, 1024, 3), (1024, 1024, 3), (1024, 128, 3), (128, 1024, 3), (128, 384, 1), (128, 128, 1), (128, 80, 1)]:
        for cd, xd, yd in [(torch.bfloat16, torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32, torch.float32)]:
            x = torch.randn(B, N, cin, device=dev).to(xd)
            w = torch.randn(cout, cin, taps, device=dev) / (cin * taps) ** 0.5
            wp = ops.pack_conv_weight(w, cd)
            bias = torch.zeros(cout, device=dev)
            out = torch.empty(B, N, cout, device=dev, dtype=yd)
            ms = timeit(lambda: ops.conv1d(x, wp, bias, out_dtype=yd, relu=True, out=out))
            fl = 2. * B * N * cin * cout * taps
            print(f'conv {cin:5d}->{cout:5d} k{taps} {str(cd)[6:]:9s}: {ms:8.3f} ms  {fl / ms / 1e9:9.1f} TFLOP/s')


if __name__ == '__main__':
    if sys.argv[1] == 'conv':
        bench_conv()


def bench_wgrad():
    dev = torch.device('cuda:0')
    B, N = 48, 1000
    lens = torch.randint(250, 1001, (B,), device=dev)
    lens[0] = N
    bf = torch.bfloat16
    for (cin, cout, taps, dyd, xd) in [(128, 1024, 3, bf, bf), (1024, 128, 3, bf, bf), (1024, 1024, 3, bf, bf), (256, 256, 3, bf, bf), (128, 384, 1, bf, bf),
                                       (128, 128, 1, bf, bf), (128, 1024, 3, bf, torch.float32)]:
        x = torch.randn(B, N, cin, device=dev).to(xd)
        dy = torch.randn(B, N, cout, device=dev).to(dyd)
        dw = torch.zeros(cout, cin, taps, device=dev) if taps > 1 else torch.zeros(cout, cin, device=dev)
        db = torch.zeros(cout, device=dev)
        ms = timeit(lambda: ops.conv1d_wgrad(dy, x, dw, db, torch.bfloat16, lens))
        fl = 2. * float(lens.sum()) * cin * cout * taps
        print(f'wgrad {cin:5d}->{cout:5d} k{taps}: {ms:8.3f} ms  {fl / ms / 1e9:9.1f} TFLOP/s (valid rows)')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'wgrad':
    bench_wgrad()


def bench_ln():
    dev = torch.device('cuda:0')
    B, N = 48, 1000
    lens = torch.randint(250, 1001, (B,), device=dev)
    lens[0] = N
    for C, dt in [(128, torch.float32), (1024, torch.bfloat16)]:
        x = torch.randn(B, N, C, device=dev).to(dt)
        res = torch.randn(B, N, C, device=dev) if C == 128 else None
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        film = torch.randn(B, 2 * C, device=dev) if C == 128 else None
        y, s, mean, rstd = ops.layernorm_fwd(x, g, b, residual=res, film=film, lengths=lens if C == 128 else None, save=True, save_s=(C == 128),
                                             out_dtype=dt, skip_lengths=lens)
        ms_f = timeit(lambda: ops.layernorm_fwd(x, g, b, residual=res, film=film, lengths=lens if C == 128 else None, save=True,
                                                save_s=(C == 128), out_dtype=dt, skip_lengths=lens))
        dy = torch.randn(B, N, C, device=dev).to(dt)
        dg, db, dfilm = torch.zeros(C, device=dev), torch.zeros(C, device=dev), (torch.zeros(B, 2 * C, device=dev) if C == 128 else None)
        sin = s if C == 128 else x
        ms_b = timeit(lambda: ops.layernorm_bwd(dy, sin, mean, rstd, g, b, dg, db, film=film, dfilm=dfilm, lengths=lens if C == 128 else None,
                                                d_dtype=dt, skip_lengths=lens, relu_input=(C != 128)))
        print(f'ln C={C}: fwd {ms_f * 1e3:7.1f} us  bwd {ms_b * 1e3:7.1f} us')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'ln':
    bench_ln()


def bench_convln():
    dev = torch.device('cuda:0')
    B, N = int(os.environ.get('BENCH_B', '48')), int(os.environ.get('BENCH_N', '1000'))
    lens = torch.randint(250, 1001, (B,), device=dev) if N == 1000 else torch.full((B,), N, device=dev)
    lens[0] = N
    for cin, taps in [(1024, 3), (128, 1)]:
        x = torch.randn(B, N, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(128, cin, taps, device=dev) / (cin * taps) ** 0.5
        wp = ops.pack_conv_weight(w, torch.bfloat16)
        bias, g, bt = torch.zeros(128, device=dev), torch.ones(128, device=dev), torch.zeros(128, device=dev)
        res = torch.randn(B, N, 128, device=dev)
        film = torch.randn(B, 256, device=dev)
        plan = ops.conv_tile_plan(lens, N) if os.environ.get('BENCH_PLAN') and taps == 3 else None
        t_fused = timeit(lambda: ops.conv1d_ln(x, wp, bias, res, g, bt, lens, film=film, save=True, p_pre=0.1, seed_pre=5, lp_copy=True, plan=plan))
        t_conv = timeit(lambda: ops.conv1d(x, wp, bias, out_dtype=torch.float32, skip_lengths=lens))
        z = ops.conv1d(x, wp, bias, out_dtype=torch.float32, skip_lengths=lens)
        t_ln = timeit(lambda: ops.layernorm_fwd(z, g, bt, residual=res, film=film, lengths=lens, save=True, save_s=True, p_pre=0.1, seed_pre=5,
                                                skip_lengths=lens, lp_copy=True))
        print(f'conv {cin}->128 k{taps}: fused conv+LN {t_fused * 1e3:6.1f} us | conv {t_conv * 1e3:6.1f} us + LN {t_ln * 1e3:6.1f} us')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'convln':
    bench_convln()


def bench_attn():
    dev = torch.device('cuda:0')
    B, N, E = int(os.environ.get('ATTN_B', '48')), 1000, 128
    g = torch.Generator().manual_seed(0)
    lo = int(os.environ.get('ATTN_LMIN', '250'))
    base = torch.randint(lo, 1001, (B,), generator=g)
    base[0] = N
    orders = {'sorted': base.sort(descending=True).values, 'random': base, 'ascending': base.sort().values}
    for tag in os.environ.get('ATTN_ORDER', 'sorted').split(','):
        lens = orders[tag].to(dev)
        for H in (8, 2):
            qkv = torch.randn(B, N, 3 * E, device=dev).to(torch.bfloat16)
            o, lse = ops.attention_fwd(qkv, lens, H, 0.1, 7)
            d_o = torch.randn(B, N, E, device=dev).to(torch.bfloat16)
            t_f = timeit(lambda: ops.attention_fwd(qkv, lens, H, 0.1, 7))
            t_f0 = timeit(lambda: ops.attention_fwd(qkv, lens, H, 0., 7))
            order = ops.length_order(lens)
            t_b = timeit(lambda: ops.attention_bwd(qkv, o, d_o, lse, lens, H, 0.1, 7, order=order))
            t_b2 = timeit(lambda: ops.attention_bwd(qkv, o, d_o, lse, lens, H, 0.1, 7, order=order, algo=ops.ATTN_TWO_PASS))
            el = float((lens.double() ** 2).sum()) * H
            print(f'attn[{tag}] d_h={E // H}: fwd {t_f * 1e3:6.1f} us (no dropout {t_f0 * 1e3:6.1f}) bwd {t_b * 1e3:6.1f} us (two-pass {t_b2 * 1e3:6.1f}) | '
                  f'{el / 1e6:.0f} M (q,k) pairs -> fwd {t_f * 1e-3 / el * 1e12:.2f} ps/pair bwd {t_b * 1e-3 / el * 1e12:.2f} ps/pair')

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'attn':
    bench_attn()


def bench_mel():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from daft_exprt import extract_features as fe
    from daft_exprt.hparams import HyperParams
    hp = HyperParams(verbose=False, training_files='none', validation_files='none', output_directory='/nonexistent', language='english',
                     speakers=['a', 'b'])
    dev = torch.device('cuda:0')
    for B, secs in ((256, 10.), (48, 11.6), (1, 10.)):
        S = int(secs * 22050)
        wavs = (torch.rand(B, S, device=dev) * 2 - 1) * 0.3
        n = torch.full((B,), S, dtype=torch.int64, device=dev)
        mel, en, nfr = fe.mel_spectrogram_batch(wavs, n, hp)
        ms = timeit(lambda: fe.mel_spectrogram_batch(wavs, n, hp), iters=10)
        frames = float(nfr.sum())
        by = frames * (256 * 4 + 80 * 4 + 4)          # algorithmic: hop samples in, mel column + energy out
        print(f'mel front-end B={B} {secs:.1f}s: {ms:8.3f} ms  {frames / ms / 1e3:8.2f} M frames/s  {by / ms / 1e6:7.1f} GB/s algorithmic '
              f'({by / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)  RTF {B * secs / (ms * 1e-3):.0f}x')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'mel':
    bench_mel()


def bench_lnbwd():
    dev = torch.device('cuda:0')
    B, N = int(os.environ.get('BENCH_B', '48')), int(os.environ.get('BENCH_N', '1000'))
    lens = torch.randint(250, 1001, (B,), device=dev) if N == 1000 else torch.full((B,), N, device=dev)
    lens[0] = N
    cin, taps = 1024, 3
    x = torch.randn(B, N, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(128, cin, taps, device=dev) / (cin * taps) ** 0.5
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    g, bt = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    y = torch.randn(B, N, 128, device=dev)
    s_in = torch.randn(B, N, 128, device=dev)
    mean, rstd = torch.zeros(B, N, device=dev), torch.ones(B, N, device=dev)
    dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
    film, dfilm = torch.randn(B, 256, device=dev), torch.zeros(B, 256, device=dev)
    plan = ops.conv_tile_plan(lens, N) if os.environ.get('BENCH_PLAN') else None
    t0 = timeit(lambda: ops.conv1d_lnbwd(x, wp, y, s_in, mean, rstd, g, bt, lens, dg, db, p_pre=0.1, seed_pre=3, plan=plan))
    t1 = timeit(lambda: ops.conv1d_lnbwd(x, wp, y, s_in, mean, rstd, g, bt, lens, dg, db, film=film, dfilm=dfilm, p_pre=0.1, seed_pre=3, plan=plan))
    print(f'lnbwd-fused data gradient 1024->128 k3, {int(lens.sum())} valid rows: {t0 * 1e3:6.1f} us | with FiLM gradients {t1 * 1e3:6.1f} us')
    # the k = 1 form: QKV data gradient (384 -> 128) + attention-LayerNorm backward
    x1 = torch.randn(B, N, 384, device=dev).to(torch.bfloat16)
    w1 = ops.pack_conv_weight(torch.randn(128, 384, 1, device=dev) / 384 ** 0.5, torch.bfloat16)
    t2 = timeit(lambda: ops.conv1d_lnbwd(x1, w1, y, s_in, mean, rstd, g, bt, lens, dg, db, p_pre=0.1, seed_pre=3))
    t3 = timeit(lambda: ops.conv1d(x1, w1, None, out_dtype=torch.float32, skip_lengths=lens))
    t4 = timeit(lambda: ops.conv1d_lnbwd(x1, w1, y, s_in, mean, rstd, g, bt, lens, dg, db, film=film, dfilm=dfilm, p_pre=0.1, seed_pre=3))
    print(f'lnbwd-fused data gradient 384->128 k1: {t2 * 1e3:6.1f} us | with FiLM gradients {t4 * 1e3:6.1f} us | plain conv 384->128 (fp32 out) {t3 * 1e3:6.1f} us')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'lnbwd':
    bench_lnbwd()

