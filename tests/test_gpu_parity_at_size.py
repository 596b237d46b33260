"""Parity with the CPU oracle AT THE SHAPES THE BENCH RUNS (BASELINE configs C2, C5, C4) and on the C1 variant.

The oracle needs ~100 s and tens of GB for a B = 48, T <= 1000 training step, so the comparison uses the fact that utterances
never interact inside the model: the HIP path runs the FULL batch (so every kernel takes the tile shapes, plans,
stage counts and workgroup splits of the bench), the oracle runs a 4-utterance slice that keeps row 0 (L_max) and the
T = T_max row -- hence the same pad extents for every kept utterance (results depend on the pad extent).
  * predictions of the kept rows: element-wise;
  * loss terms: the HIP loss kernel on the kept rows vs the oracle's loss on the slice;
  * gradients: the loss gradient of the dropped rows is zeroed before the hand-written backward runs over the FULL batch, which
    makes the result (n_keep / B) x [gradient of the slice's six batch-mean terms] + [post-multiplier term] -- the oracle
    computes exactly that by autograd; all 193 tensors are compared ELEMENT-WISE.
Tolerances (each relative to the max magnitude of the tensor compared).
  fp32 operand mode vs the oracle: 2e-4 on predictions, 1e-4 on loss terms, 2e-3 per gradient element (plus 2e-5 of the largest
  gradient element of the model, for tensors whose true gradient is ~0 such as the key biases).
  bf16 operand mode (the bench's).  With random-init weights the network amplifies a perturbation ~3.6x per FFT block
  (measured: two runs that differ by a handful of 1-ulp bf16 roundings in block 0 are 2.6e-3 apart after the 4-block phoneme
  encoder and 8e-3 apart at the duration head), so end to end NOTHING tracks the bf16 path closely, not even an oracle with the
  same rounding points.  It is therefore checked three ways:
    (i)   STAGE BY STAGE at the full size: every FFT block (12), every conv+LayerNorm stage (5) and the mel projection of the
          HIP forward is re-computed by the oracle with `OPERAND_DTYPE = bfloat16` (same functions, bf16 rounding inserted
          where the HIP path rounds: GEMM operands, stored wide tensors, flash-style probabilities) FROM THE HIP PATH'S OWN
          fp32 INPUT of that stage: mean error <= 3e-4 of the tensor's mean magnitude, at most 1 % of the elements further
          than 1e-3 of the max, none further than 2e-2 (a few bf16 ulps) -- a wrong index / mask / tile / reduction shows;
    (i-b) the BACKWARD of the same stages, likewise at full size and from the HIP path's own stage input and upstream gradient
          (`_stagewise_bf16_backward`): the data gradient of every FFT block (on the boundaries the fused kernels materialise:
          dL/d(s2) in, dL/d(s2 of the block below) out -- the LayerNorm backward fused into the QKV data-gradient GEMM, the
          fused attention backward, the register-weights and split-K data-gradient GEMMs and the ring weight gradients all sit
          inside one stage) and of every conv+LayerNorm stage, plus EVERY parameter gradient of the stage, against the
          bf16-emulating oracle's autograd: mean error <= 1e-3 of the mean magnitude, <= 1 % of the elements further than
          1e-2 of the max -- the check that a 5 % systematic error in a fused backward epilogue cannot pass;
    (ii)  end to end against the exact fp32 oracle, i.e. the reference's arithmetic: what a user of the bf16 mode sees at
          T = 1000 with random-init weights -- tolerances in TOL['bf16'] (the alignments and through them the mel move the
          most, SURVEY App. B item 9: the reference itself moves ~10 % there under bf16 autocast);
    (iii) gradients end to end against the bf16-emulating oracle AND the fp32 oracle at the TOL['bf16'] tolerance (10 % of a
          tensor's max per element; 35 % on the four sigma-path parameters of the upsampler, whose gradients exist only through
          the alignments -- the tensor that bf16 moves by 15-26 %).
Dropout is off (the reference's Philox streams are not reproduced)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import daft_exprt_cpu as O
from tests.util import drop_unwritten, make_hparams, no_dropout, gradient_report

DEV = 'cuda:0'
TOL = {'fp32': dict(pred=2e-4, loss=1e-4, grad=2e-3, floor=2e-5),
       'bf16': dict(pred=3e-2, loss=2e-2, grad=0.10, floor=1e-3, pred_mel=1.5e-1, pred_weights=3e-1)}
TOL['bf16_emulated'] = TOL['bf16']
def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _keep_rows(inputs, n_keep):
    ''' row 0 (L_max), the first row with T = T_max, then the next rows in order '''
    out_len = inputs[9].cpu()
    rows = [0, int(out_len.argmax())]
    for r in range(out_len.shape[0]):
        if len(set(rows)) >= n_keep:
            break
        rows.append(r)
    return sorted(set(rows))


def _hip_full_batch(model, inputs, targets, weights, rows):
    ''' forward over the full batch, loss gradient restricted to `rows`, backward over the full batch '''
    from daft_exprt import ops
    model.zero_grad()
    (logits, films, (dur, energy, pitch), mel, w), S = model._forward(inputs, True, True)
    B, n_mel, T = mel.shape
    g = {'d_dur': torch.empty_like(dur), 'd_energy': torch.empty_like(energy), 'd_pitch': torch.empty_like(pitch),
         'd_mel': torch.empty((B, T, n_mel), dtype=torch.float32, device=mel.device), 'd_spk': torch.empty_like(logits)}
    post = model._P['prosody_encoder.post_multipliers']
    ops.loss_fwd_bwd(dur, energy, pitch, targets[0], targets[1], targets[2], inputs[5], mel, targets[3], inputs[9], logits, targets[4],
                     post, weights, grads=g, d_post_mult=model._G['prosody_encoder.post_multipliers'], grad_scale=1., d_mel_transposed=True)
    drop = torch.ones(B, dtype=torch.bool, device=mel.device)
    drop[rows] = False
    for t in g.values():
        t[drop] = 0.
    model._backward(S, g['d_spk'], g['d_dur'], g['d_energy'], g['d_pitch'], g['d_mel'], d_mel_is_bt=True)
    torch.cuda.synchronize()
    idx = torch.tensor(rows, device=mel.device)
    sl = lambda t: t.index_select(0, idx).contiguous()
    terms = ops.loss_fwd_bwd(sl(dur), sl(energy), sl(pitch), sl(targets[0]), sl(targets[1]), sl(targets[2]), sl(inputs[5]), sl(mel),
                             sl(targets[3]), sl(inputs[9]), sl(logits), sl(targets[4]), post, weights)
    preds = {'speaker': sl(logits), 'duration': sl(dur), 'energy': sl(energy), 'pitch': sl(pitch), 'mel': sl(mel), 'weights': sl(w),
             'enc_film': sl(films[0]), 'pp_film': sl(films[1]), 'dec_film': sl(films[2])}
    return preds, terms.cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}


def _oracle_slice(hp, state, inputs, rows, B, iteration, operand_dtype=None):
    O.OPERAND_DTYPE = operand_dtype
    try:
        return _oracle_slice_impl(hp, state, inputs, rows, B, iteration)
    finally:
        O.OPERAND_DTYPE = None


def _oracle_slice_impl(hp, state, inputs, rows, B, iteration):
    P = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()}
    cin = tuple(t.cpu()[rows] for t in inputs)
    out = O.forward(P, hp, cin, training=True)
    total, terms = O.loss(hp, out, (cin[1], cin[3], cin[4], cin[8], cin[10]), iteration)
    objective = (len(rows) / float(B)) * (total - terms['post_mult_loss']) + terms['post_mult_loss']
    grads = torch.autograd.grad(objective, list(P.values()), allow_unused=True)
    preds = {'speaker': out[0], 'duration': out[2][0], 'energy': out[2][1], 'pitch': out[2][2], 'mel': out[3][0], 'weights': out[4],
             'enc_film': out[1][1], 'pp_film': out[1][2], 'dec_film': out[1][3]}
    keys = ('speaker_loss', 'post_mult_loss', 'duration_loss', 'energy_loss', 'pitch_loss', 'mel_spec_l1_loss', 'mel_spec_l2_loss')
    t = torch.stack([terms[k].detach().float().reshape(()) for k in keys] + [total.detach().float().reshape(())])
    return preds, t, {k: (g if g is not None else torch.zeros_like(P[k])) for k, g in zip(P, grads)}


def _compare(mode, hip, ora, what, sigma_factor=None):
    tol = TOL[mode]
    (hp_preds, hp_terms, hp_grads), (or_preds, or_terms, or_grads) = hip, ora
    errs = {k: _rel(hp_preds[k], or_preds[k]) for k in or_preds}
    print(what, mode, 'prediction errors', {k: f'{v:.2e}' for k, v in errs.items()})
    assert all(v <= tol.get(f'pred_{k}', tol['pred']) for k, v in errs.items()), (what, mode, errs)
    lerr = float((hp_terms - or_terms).abs().max() / or_terms.abs().max())
    print(what, mode, 'loss terms', hp_terms.tolist(), or_terms.tolist())
    assert lerr <= tol['loss'], (what, mode, lerr)
    worst = gradient_report(hp_grads, or_grads, tol['grad'], tol['floor'], sigma_factor=sigma_factor or (2. if mode == 'fp32' else 3.5))
    print(what, mode, 'worst gradient tensors (err / bound, name, max abs err, max abs ref):')
    for w in worst[:6]:
        print('   ', f'{w[0]:.3f}', w[1], f'{w[2]:.3e}', f'{w[3]:.3e}')
    assert worst[0][0] <= 1., (what, mode, worst[:4])


def _stagewise_bf16(model, hp, state, inputs, rows, what):
    ''' (i) of the module docstring: each stage of the HIP forward at full size vs the bf16-emulating oracle on that stage's
        own HIP input (kept rows) '''
    model._trace = []
    try:
        with torch.no_grad():
            model._forward(inputs, True, False)
        torch.cuda.synchronize()
        trace, model._trace = model._trace, None
    finally:
        model._trace = None
    _stagewise_check(trace, hp, state, rows, what)


def _stagewise_check(trace, hp, state, rows, what, train=False):
    ''' the recorded stages of a HIP forward pass (training `_forward` or `inference`) against the bf16-emulating oracle.
        train = True: dropout on -- the caller has replaced `O.dropout` by a feed of the HIP pass's own masks
        (tests/test_gpu_dropout_parity.py), consumed in stage order '''
    P = state
    cfgs = {'prosody_encoder': hp.prosody_encoder, 'phoneme_encoder': hp.phoneme_encoder, 'frame_decoder': hp.frame_decoder}
    cpu = lambda t: None if t is None else t.detach().float().cpu()[rows]
    report, n_checked = [], 0
    O.OPERAND_DTYPE = torch.bfloat16
    try:
        with torch.no_grad():
            for kind, names, x, film, lengths, out in trace:
                # rows past the fill end of a (B, N, C) activation are unwritten (csrc/dx_common.h dx_fill_end): dropped before the
                # oracle sees the tensor / before the comparison
                lfill = (lengths if lengths is not None else names[2]).cpu()[rows]
                drop = lambda t: t if (t is None or kind == 'mel_projection' and t.shape[1] != x.shape[1]) else drop_unwritten(t, lfill)
                xs, fs = drop(cpu(x)), cpu(film)
                out_dtype = None if isinstance(out, tuple) else out.dtype
                out = tuple(drop(cpu(o)) for o in out) if isinstance(out, tuple) else drop(cpu(out))
                ls = None if lengths is None else lengths.cpu()[rows]
                N = xs.shape[1]
                if kind == 'fft_block':
                    pad = ~O.valid_mask(ls, N)
                    cfg = cfgs[names.split('.')[0]]
                    a = O.multi_head_attention(P, names + '.attention.', xs, pad, cfg['attn_nb_heads'], cfg['attn_dropout'] if train else 0.,
                                               train).masked_fill(pad.unsqueeze(2), 0.)
                    u = O.conv_ff(P, names + '.feed_forward.', out[0], fs, cfg['conv_dropout'] if train else 0., train).masked_fill(pad.unsqueeze(2), 0.)
                    pairs = [(names + ' attention+LN', out[0], a), (names + ' FF+LN (HIP attention output in)', out[1], u)]
                elif kind == 'conv_ln':
                    conv_name, ln_name, skip = names
                    skip = skip.cpu()[rows]
                    y = torch.relu(O.conv1d_cl(xs, P[conv_name + '.conv.weight'], P[conv_name + '.conv.bias']))
                    if y.shape[2] != 128:
                        y = O._stored_lp(y)
                    y = O.layer_norm(y, P[ln_name + '.weight'], P[ln_name + '.bias'])
                    if train:
                        y = O.dropout(y, (hp.prosody_encoder if conv_name.startswith('prosody_encoder') else hp.local_prosody_predictor)['conv_dropout'], True)
                    if fs is not None:
                        C = fs.shape[1] // 2
                        y = fs[:, None, :C] * y + fs[:, None, C:]
                    if ls is not None:
                        y = y.masked_fill(~O.valid_mask(ls, N).unsqueeze(2), 0.)
                    if out_dtype == torch.bfloat16:
                        y = O._op(y)
                    pairs = [(conv_name + ' conv+ReLU+LN', out, y)]
                else:
                    pad = ~O.valid_mask(ls, N)
                    mel = O.linear_mfma(xs, P[names + '.weight'], P[names + '.bias']).masked_fill(pad.unsqueeze(2), 0.).transpose(1, 2)
                    pairs = [(names, out, mel)]
                for name, got, ref in pairs:
                    # rows past len + 2 of un-masked stages are never consumed (padding early-out writes zeros there)
                    if kind == 'mel_projection':
                        got, ref = got.transpose(1, 2), ref.transpose(1, 2)
                    live = O.valid_mask(skip + 2, N) if kind == 'conv_ln' else torch.ones(got.shape[:2], dtype=torch.bool)
                    d = (got - ref).abs()[live]
                    mx, mean = float(d.max() / ref.abs().max()), float(d.mean() / ref[live].abs().mean())
                    frac = float((d > 1e-3 * ref.abs().max()).float().mean())
                    report.append((max(mx / 2e-2, mean / 3e-4, frac / 1e-2), name, mx, mean, frac))
                    n_checked += 1
    finally:
        O.OPERAND_DTYPE = None
    report.sort(reverse=True)
    print(what, f'stage-by-stage bf16 check, {n_checked} stage outputs; worst (score, stage, max, mean, share > 1e-3):')
    for r in report[:8]:
        print('   ', f'{r[0]:.3f}', r[1], f'{r[2]:.2e}', f'{r[3]:.2e}', f'{r[4]:.4f}')
    assert n_checked == 12 * 2 + 5 + 1, n_checked
    assert report[0][0] <= 1., report[:4]



def _stagewise_bf16_backward(model, hp, state, trace, rows, hip_grads, what):
    ''' the BACKWARD of every stage of the bf16 path against the oracle's autograd of the same stage, at full size.
        The loss gradient of this run is restricted to `rows`, so every gradient -- also the parameter gradients -- is the kept rows'
        own.  FFT blocks are compared on the boundaries the fused kernels materialise: gradient in = dL/d(s2_b) (the input of the
        block's second LayerNorm), gradient out = dL/d(s2_{b-1}) for the block below (its LayerNorm backward runs inside this
        block's QKV data-gradient launch, dx_conv1d_lnbwd) or dL/d(block input) for the lowest block; the oracle stage is
        s2_{b-1} -> LayerNorm [FiLM, mask] -> attention -> LayerNorm -> conv k3 -> ReLU -> conv k3 -> + residual = s2_b.
        conv + LayerNorm stages: dL/dy in, dL/dx out (where the data gradient is materialised on its own).
        Yardstick: the oracle is run twice per stage, exact fp32 and bf16-emulating.  Their distance d_r is what bf16 operand
        rounding does to this gradient; the HIP path rounds at the same places but not in the same order (flash-style softmax,
        split sums), so it is asked to sit within max(2 d_r, 2e-3) of the emulating oracle in the mean (relative to the mean
        magnitude) with an absolute cap of 2e-2, and to have <= 1 % of its elements further than 3e-2 of the maximum -- a 5 %
        systematic error in a fused backward epilogue fails both. '''
    cfgs = {'prosody_encoder': hp.prosody_encoder, 'phoneme_encoder': hp.phoneme_encoder, 'frame_decoder': hp.frame_decoder}

    def cpu(t, lfill=None):
        ''' kept rows on the host; with `lfill` (the stage's lengths, kept rows) the rows past the fill end -- unwritten by
            contract, csrc/dx_common.h dx_fill_end -- are dropped '''
        if t is None:
            return None
        t = t.detach().float().cpu()[rows]
        return t if lfill is None or t.dim() != 3 else drop_unwritten(t, lfill)
    report, n_fft, n_conv = [], 0, 0

    def score(name, got, emu, exact, live=None):
        got, emu, exact = got.float(), emu.float(), exact.float()
        d, dr = (got - emu).abs(), (emu - exact).abs()
        refl = emu
        if live is not None:
            d, dr, refl = d[live], dr[live], emu[live]
        if refl.numel() == 0 or float(emu.abs().max()) == 0.:
            assert float(d.max() if d.numel() else 0.) == 0., name
            return
        scale = float(refl.abs().mean()) + 1e-30
        mean, mean_r = float(d.mean()) / scale, float(dr.mean()) / scale
        frac = float((d > 3e-2 * emu.abs().max()).float().mean())
        bound = min(2e-2, max(2. * mean_r, 2e-3))
        report.append((max(mean / bound, frac / 1e-2), name, mean, frac, float(d.max() / emu.abs().max()), mean_r))

    def fft_stage(sv, below, g_in, mode):
        O.OPERAND_DTYPE = mode
        pre, cfg = sv.pre, cfgs[sv.pre.split('.')[0]]
        ls = sv.lengths.cpu()[rows]
        names = [f'{pre}.attention.multi_head_attention.in_proj_weight', f'{pre}.attention.multi_head_attention.in_proj_bias',
                 f'{pre}.attention.multi_head_attention.out_proj.weight', f'{pre}.attention.multi_head_attention.out_proj.bias',
                 f'{pre}.attention.layer_norm.weight', f'{pre}.attention.layer_norm.bias',
                 f'{pre}.feed_forward.convs.0.conv.weight', f'{pre}.feed_forward.convs.0.conv.bias',
                 f'{pre}.feed_forward.convs.2.conv.weight', f'{pre}.feed_forward.convs.2.conv.bias']
        if below is not None:
            names += [f'{below.pre}.feed_forward.layer_norm.weight', f'{below.pre}.feed_forward.layer_norm.bias']
        P = {k: v.detach().clone() for k, v in state.items()}
        for n in names:
            P[n].requires_grad_(True)
        xin = cpu(below.s2 if below is not None else sv.x, ls).requires_grad_(True)
        N = xin.shape[1]
        pad = ~O.valid_mask(ls, N)
        if below is not None:
            x = O.layer_norm(xin, P[f'{below.pre}.feed_forward.layer_norm.weight'], P[f'{below.pre}.feed_forward.layer_norm.bias'])
            fb = cpu(below.film)
            if fb is not None:
                C = fb.shape[1] // 2
                x = fb[:, None, :C] * x + fb[:, None, C:]
            x = x.masked_fill(pad.unsqueeze(2), 0.)
        else:
            x = xin
        a = O.multi_head_attention(P, pre + '.attention.', x, pad, cfg['attn_nb_heads'], 0., False).masked_fill(pad.unsqueeze(2), 0.)
        f = pre + '.feed_forward.'
        hh = torch.relu(O.conv1d_cl(a, P[f + 'convs.0.conv.weight'], P[f + 'convs.0.conv.bias']))
        if mode is not None:
            hh = O._stored_lp(hh)
        s2 = O.conv1d_cl(hh, P[f + 'convs.2.conv.weight'], P[f + 'convs.2.conv.bias']) + a
        grads = torch.autograd.grad(s2, [xin] + [P[n] for n in names], grad_outputs=g_in, allow_unused=True)
        grads = [gr if gr is not None else torch.zeros_like(t) for gr, t in zip(grads, [xin] + [P[n] for n in names])]
        return names, grads, O.valid_mask(ls, N)

    def conv_stage(sv, g_in, mode):
        O.OPERAND_DTYPE = mode
        names = [sv.conv_name + '.conv.weight', sv.conv_name + '.conv.bias', sv.ln_name + '.weight', sv.ln_name + '.bias']
        P = {k: v.detach().clone() for k, v in state.items()}
        for n in names:
            P[n].requires_grad_(True)
        xin = cpu(sv.x, sv.skip.cpu()[rows]).requires_grad_(True)
        N = xin.shape[1]
        y = torch.relu(O.conv1d_cl(xin, P[names[0]], P[names[1]]))
        if y.shape[2] != 128 and mode is not None:
            y = O._stored_lp(y)
        y = O.layer_norm(y, P[names[2]], P[names[3]])
        fs = cpu(sv.film)
        if fs is not None:
            C = fs.shape[1] // 2
            y = fs[:, None, :C] * y + fs[:, None, C:]
        if sv.lengths is not None:
            y = y.masked_fill(~O.valid_mask(sv.lengths.cpu()[rows], N).unsqueeze(2), 0.)
        grads = torch.autograd.grad(y, [xin] + [P[n] for n in names], grad_outputs=g_in)
        return names, list(grads), O.valid_mask(sv.skip.cpu()[rows] + 2, N)

    try:
        for kind, sv, below, g_in, g_out in trace:
            lfill = (sv.lengths if kind == 'fft_block' else sv.skip).cpu()[rows]
            gi = cpu(g_in, lfill)
            if kind == 'fft_block':
                names, g_emu, live = fft_stage(sv, below, gi, torch.bfloat16)
                _, g_ex, _ = fft_stage(sv, below, gi, None)
                tag = sv.pre
                n_fft += 1
            else:
                names, g_emu, live = conv_stage(sv, gi, torch.bfloat16)
                _, g_ex, _ = conv_stage(sv, gi, None)
                tag = sv.conv_name
                n_conv += 1
            if g_out is not None:
                score(f'{tag}: data gradient', cpu(g_out, lfill), g_emu[0], g_ex[0], live)
            for n, ge, gx in zip(names, g_emu[1:], g_ex[1:]):
                score(f'{tag}: d {n}', hip_grads[n], ge, gx)
    finally:
        O.OPERAND_DTYPE = None
    report.sort(reverse=True)
    print(what, f'stage-by-stage bf16 BACKWARD check, {n_fft} FFT-block stages + {n_conv} conv+LayerNorm stages, {len(report)} tensors; '
                'worst (score, tensor, mean error vs the bf16-emulating oracle, that oracle vs exact fp32, share > 3e-2 max, max):')
    for r in report[:10]:
        print('   ', f'{r[0]:.3f}', r[1], f'{r[2]:.2e}', f'{r[5]:.2e}', f'{r[3]:.4f}', f'{r[4]:.2e}')
    assert n_fft == 12 and n_conv == 5, (n_fft, n_conv)
    assert report[0][0] <= 1., report[:6]


def _train_case(mode, batch_size, t_min, n_keep, what, speakers=None, t_max=1000, seed=1234):
    import bench
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.loss import DaftExprtLoss
    from daft_exprt.model import DaftExprt
    hp = bench.make_hparams(batch_size, mode)
    if speakers is not None:
        hp = make_hparams(speakers=list(speakers), batch_size=batch_size, accumulation_steps=1, compute_dtype=mode)
    hp = no_dropout(hp)
    torch.manual_seed(hp.seed)
    model = DaftExprt(hp)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    cb = synthetic_batch(hp, batch_size, seed=seed, t_min=t_min, t_max=t_max, force_first_full=True)
    inputs, targets, _ = model.parse_batch(DEV, cb)
    assert int(inputs[9].max()) == t_max and inputs[0].shape[0] == batch_size
    rows = _keep_rows(inputs, n_keep)
    weights = DaftExprtLoss(0, hp).weights(20000)          # adversarial weight at its maximum: GRL path live
    if mode == 'bf16':
        model._trace_bwd = []
    try:
        hip = _hip_full_batch(model, inputs, targets, weights, rows)
        trace_bwd = model._trace_bwd
    finally:
        model._trace_bwd = None
    if mode == 'bf16':
        _stagewise_bf16(model, hp, state, inputs, rows, what)
        _stagewise_bf16_backward(model, hp, state, trace_bwd, rows, hip[2], what)
        _compare('bf16_emulated', hip, _oracle_slice(hp, state, inputs, rows, batch_size, 20000, torch.bfloat16), what)
    _compare(mode, hip, _oracle_slice(hp, state, inputs, rows, batch_size, 20000), what)
    return model, inputs, targets, weights


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_c2_bench_batch_matches_oracle_slice(mode):
    ''' BASELINE configs[1]: B = 48, 11 speakers, T <= 1000 (utterance 0 = 1000 frames): the bench batch itself (seed 1234) '''
    _train_case(mode, 48, 1, 4, 'C2')


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_c5_long_utterance_batch_matches_oracle_slice(mode):
    ''' BASELINE configs[4]: B = 256, 500 <= T <= 1000, adversarial weight at max '''
    model, inputs, targets, weights = _train_case(mode, 256, 500, 4, 'C5')
    if mode != 'bf16':
        return
    # size-independent properties at this size: rerun reproduces every prediction bit for bit; fixed tiles == balanced tiles
    def step():
        model.zero_grad()
        model._step_id = 3
        terms = model.forward_backward(inputs, targets, weights)
        torch.cuda.synchronize()
        return terms.clone(), model._gflat.clone(), model.last_outputs[3].clone()
    from daft_exprt import ops
    t0, g0, m0 = step()
    t1, g1, m1 = step()
    assert torch.equal(m0, m1) and torch.allclose(t0, t1, rtol=1e-5, atol=0.)      # (loss sums: block partials meet in fp32 atomics)
    assert float((g0 - g1).norm()) <= 2e-4 * float(g0.norm())
    ops.USE_SPLITK = False            # (the phoneme-level GEMMs of this batch fit one round of tiles and take the split-K kernel: other sums)
    try:
        t3, g3, m3 = step()
        model.balanced_tiles = False
        t2, g2, m2 = step()
    finally:
        ops.USE_SPLITK = True
        model.balanced_tiles = True
    assert torch.equal(m3, m2) and float((g3 - g2).norm()) <= 3e-4 * float(g3.norm())
    dm, scale = (m0 - m3).abs(), float(m3.abs().max())     # other summation order at the phoneme level, amplified through 8 FFT blocks
    assert float(dm.mean()) <= 1e-2 * scale and float(dm.max()) <= 0.3 * scale and float((g0 - g3).norm()) <= 5e-2 * float(g3.norm())


def test_c1_single_speaker_variant_matches_oracle():
    ''' BASELINE configs[0] on the HIP path: one speaker => n_speakers = 2, a 1-logit classifier, cross-entropy identically 0
        (SURVEY App. B item 4); B = 8.  The whole batch goes through the oracle. '''
    _train_case('fp32', 8, 1, 8, 'C1', speakers=['LJ'], t_max=500, seed=77)


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_c4_batched_synthesis_matches_oracle_slice(mode):
    ''' BASELINE configs[3]: `inference` on 256 sentences, an 8-sentence slice through the oracle.  The slice keeps row 0 (L_max),
        the longest reference and the longest generated utterance, so pad extents agree everywhere.
        Staged so that the integer path is checked bit-exactly on identical floats: (1) float durations / energy / pitch vs the
        oracle -- 2e-4 in fp32 operand mode, 3e-2 in bf16 mode (the bench's: the exact oracle is the reference's arithmetic);
        (2) the oracle's `get_int_durations` on the HIP path's own float durations == the HIP integer durations, bit for bit;
        (3) the upsampler with PREDICTED durations (training only ever feeds it ground-truth ones): the oracle's
        Gaussian upsampling + positional add + mask on the HIP path's own encoder output and prosody == the decoder input and the
        alignments the HIP path produced (fp32 kernels in both modes: 5e-4); (4) fp32: the oracle's decoder fed with those == the
        HIP mel; bf16: EVERY stage of the synthesis pass (12 FFT blocks, 5 conv + LayerNorm stages, the mel projection -- the
        decoder blocks run on utterance lengths the model itself generated) against the bf16-emulating oracle on that stage's
        own HIP input, the stage-wise yardstick of the training tests. '''
    import bench
    from daft_exprt.data_loader import centre_duration_head, synthetic_inference_batch
    from daft_exprt.model import DaftExprt
    hp = bench.make_hparams(256, mode)
    hp.stats = {f'spk {i}': {'pitch': {'mean': 5.0, 'std': 0.3}} for i in range(hp.n_speakers)}
    torch.manual_seed(hp.seed)
    model = DaftExprt(hp).to(DEV).eval()
    centre_duration_head(model)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cpu_in = synthetic_inference_batch(hp, 256, seed=1234)
    model._trace = []
    try:
        enc_p, dec_p, weights = model.inference(tuple(t.to(DEV) for t in cpu_in), 'add', hp)
        torch.cuda.synchronize()
        trace = model._trace
    finally:
        model._trace = None
    tol = 2e-4 if mode == 'fp32' else 3e-2
    dur, dur_int, energy, pitch, in_len = (t.cpu() for t in enc_p)
    mel, out_len = (t.cpu() for t in dec_p)
    assert mel.shape[0] == 256 and int(out_len.max()) == mel.shape[2] and int(out_len.min()) > 100
    rows = sorted({0, int(cpu_in[8].argmax()), int(out_len.argmax())} | set(range(3, 8)))
    sl = tuple(t[rows] for t in cpu_in)
    symbols, dur_f, en_f, pi_f, in_l, e_ref, p_ref, m_ref, ref_l, spk = sl
    # the HIP path's own encoder output and decoder input, from the stage trace: output of the phoneme encoder's last block,
    # input of the frame decoder's first block
    blocks = [t for t in trace if t[0] == 'fft_block']
    enc_hip = [t for t in blocks if t[1].startswith('phoneme_encoder.')][-1][5][1].float().cpu()[rows]
    dec_in_hip = [t for t in blocks if t[1].startswith('frame_decoder.')][0][2].float().cpu()[rows]
    with torch.no_grad():
        _, enc_film, pp_film, dec_film = O.prosody_encoder(P, hp, e_ref, p_ref, m_ref, spk, ref_l, False)
        enc = O.phoneme_encoder(P, hp, symbols, enc_film, in_l, False)
        o_dur, o_energy, o_pitch = O.prosody_predictor(P, hp, enc, pp_film, in_l, False)
        o_dur = o_dur * dur_f
        thr, o_int = O.get_int_durations(o_dur.clone(), hp)
        if mode == 'fp32':
            assert _rel(dur[rows], thr) <= tol
        else:   # a prediction within rounding distance of the threshold may be zeroed on one side only: compare where both kept it
            both = (dur[rows] != 0) & (thr != 0)
            assert float(((dur[rows] - thr).abs() * both).max() / thr.abs().max()) <= tol
            assert float(((dur[rows] != 0) != (thr != 0)).float().mean()) <= 2e-2
        # (2) integer path on identical floats
        same, h_int = O.get_int_durations(dur[rows].clone(), hp)
        assert torch.equal(same, dur[rows]) and torch.equal(h_int, dur_int[rows])
        assert torch.equal(h_int.sum(1), out_len[rows])
        print(f'C4 {mode}: oracle-on-oracle-floats integer durations differ from HIP in', int((o_int != h_int).sum()), 'of', h_int.numel(), 'symbols')
        o_energy = o_energy * en_f
        o_energy[h_int == 0] = 0.
        o_pitch = o_pitch.clone()
        o_pitch[h_int == 0] = 0.
        o_pitch = O.pitch_shift(o_pitch, pi_f, hp, spk)
        if mode == 'fp32':
            assert _rel(energy[rows], o_energy) <= tol and _rel(pitch[rows], o_pitch) <= tol
        else:   # symbols whose integer duration differs between the two (threshold / rounding flips) are zeroed on one side only
            agree = (o_int == 0) == (h_int == 0)
            for got, ref in ((energy[rows], o_energy), (pitch[rows], o_pitch)):
                assert float(((got - ref).abs() * agree).max() / ref.abs().max()) <= tol
        # (3) upsampling with predicted durations, from the HIP path's own encoder output and prosody (fp32 kernels in both modes)
        x_up, o_w = O.gaussian_upsampling(P, hp, enc_hip if mode == 'bf16' else enc, dur[rows], h_int, energy[rows], pitch[rows], in_l)
        assert x_up.shape[1] == mel.shape[2]
        if mode == 'bf16':
            D = hp.phoneme_encoder['hidden_embed_dim']
            pad = ~O.valid_mask(out_len[rows], x_up.shape[1])
            o_dec_in = (x_up + O.pos_encoding(out_len[rows], D)[:, :x_up.shape[1]]).masked_fill(pad.unsqueeze(2), 0.)
            errs = {'decoder input': _rel(dec_in_hip, o_dec_in), 'weights': _rel(weights.cpu()[rows], o_w)}
            print('C4 bf16 upsampler stage (predicted durations)', errs)
            assert errs['decoder input'] <= 5e-4 and errs['weights'] <= 5e-4, errs
        else:
            o_mel = O.frame_decoder(P, hp, x_up, dec_film, out_len[rows], False)
            errs = {'mel': _rel(mel[rows], o_mel), 'weights': _rel(weights.cpu()[rows], o_w)}
            print('C4 slice', errs)
            assert errs['mel'] <= 5e-4 and errs['weights'] <= 5e-4, errs
    if mode == 'bf16':
        _stagewise_check(trace, hp, P, rows, 'C4 bf16 synthesis')
    for b, t in zip(rows, out_len[rows].tolist()):
        assert not mel[b, :, t:].any()
