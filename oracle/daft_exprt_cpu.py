"""CPU oracle for the Daft-Exprt hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, from scratch and in explicit math on PyTorch-CPU fp32 tensors (fp64 /
Python ints for the integer duration path), the algorithm of the reference hot path
(`/root/reference/src/daft_exprt/model.py`, `loss.py`, `train.py:139-151`,
`extract_features.py:69-111`, `data_loader.py:146-211`, `generate.py:140-239`).
Every function cites the reference lines it follows.

Who may use it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` -- as the checker / the timed CPU baseline only.  The product package
(`ubisoft-laforge-daft-exprt_amd/daft_exprt`) never imports it; the product fails loudly
when the HIP library is missing instead of falling back to this file.

Pinning: the reference ships no tests / golden vectors, so the oracle is pinned
against fixtures produced by importing the reference itself in the build container
(`tools/gen_goldens.py` -> `tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).

Layout conventions: activations are channel-last `(B, N, C)`; parameters live in a flat
dict keyed by the reference `state_dict` names, fp32.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# optional emulation of the HIP path's bf16 operand mode (tests/test_gpu_parity_at_size.py)
# ----------------------------------------------------------------------------------------
# OPERAND_DTYPE = None (default): exact fp32 everywhere -- the restatement that is pinned against the reference fixtures.
# OPERAND_DTYPE = torch.bfloat16: every GEMM that the HIP path runs on bf16 MFMA operands (conv / linear layers with Cin > 1
# except the small fp32 heads, and the two attention contractions) rounds its operands -- activations, weights and, in the
# backward pass, the incoming gradient -- to bf16 and accumulates in fp32; wide (> 128-channel) conv outputs are rounded where
# the HIP path stores them in bf16.  Same functions, same order of operations: only roundings are inserted, so a comparison
# against it isolates the kernels' indexing / masking / reduction logic from the (large, T-dependent) bf16 rounding noise.
OPERAND_DTYPE = None


class _RoundForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _op(x):
    ''' MFMA operand rounding (value); straight-through for the gradient '''
    return x if OPERAND_DTYPE is None else _RoundForward.apply(x, OPERAND_DTYPE)


def _og(y):
    ''' output of an MFMA GEMM: its gradient is an MFMA operand of the backward GEMMs '''
    return y if OPERAND_DTYPE is None else _RoundBackward.apply(y, OPERAND_DTYPE)


def _stored_lp(y):
    ''' tensors the HIP path keeps in the operand type between kernels: rounded value AND rounded gradient '''
    return y if OPERAND_DTYPE is None else _RoundBackward.apply(_RoundForward.apply(y, OPERAND_DTYPE), OPERAND_DTYPE)


def linear_mfma(x, w, b):
    ''' x W^T + b for the layers the HIP path runs on MFMA (in/out projections, mel projection) '''
    return _og(_op(x) @ _op(w).t()) + b


# ----------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------
def valid_mask(lengths, n_max=None):
    ''' True where position < length.  model.py:14-24 (callers negate it). N = max(lengths). '''
    n_max = int(lengths.max()) if n_max is None else n_max
    return torch.arange(n_max)[None, :] < lengths[:, None]


def pos_table(max_len=5000, dim=128, timestep=10000.):
    ''' Sinusoid table, even columns sin, odd columns cos.  model.py:123-130 '''
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2).float() * (-np.log(timestep) / dim))
    table = torch.zeros(max_len, dim)
    table[:, 0::2] = torch.sin(pos * div)
    table[:, 1::2] = torch.cos(pos * div)
    return table


_POS_CACHE = {}


def pos_encoding(lengths, dim=128):
    ''' pos[b, t] = table[t] for t < lengths[b], else 0; width max(lengths).
        model.py:132-150 with x = lengths[:, None] (every call site, model.py:400,499,696). '''
    if dim not in _POS_CACHE:
        _POS_CACHE[dim] = pos_table(dim=dim)
    n_max = int(lengths.max())
    out = _POS_CACHE[dim][:n_max].unsqueeze(0).repeat(lengths.numel(), 1, 1)
    return out * valid_mask(lengths, n_max).unsqueeze(2).float()


def conv1d_cl(x, w, b):
    ''' k-tap, stride-1, zero "same" padding conv on channel-last x (B,N,Cin); w (Cout,Cin,K).
        model.py:86-94 (ConvNorm1D.forward: transpose, nn.Conv1d, transpose). '''
    pad = (w.shape[2] - 1) // 2
    if OPERAND_DTYPE is not None and w.shape[1] > 1:      # Cin = 1 scalar embeddings are fp32 pointwise kernels
        return _og(F.conv1d(_op(x).transpose(1, 2), _op(w), None, padding=pad).transpose(1, 2)) + b
    return F.conv1d(x.transpose(1, 2), w, b, padding=pad).transpose(1, 2)


def layer_norm(x, g, b, eps=1e-5):
    ''' LayerNorm over the last dim, biased variance, eps 1e-5 (torch default used at
        model.py:169,218,347,354,361,534,541). '''
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def dropout(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0.) else x


def grad_reverse(x, lambda_):
    ''' identity forward, -lambda * grad backward.  model.py:27-38 '''
    xd = x.detach()
    return (1. + lambda_) * xd - lambda_ * x if x.requires_grad else x


# ----------------------------------------------------------------------------------------
# FFT block
# ----------------------------------------------------------------------------------------
def multi_head_attention(P, pre, x, pad, nb_heads, p_drop, training):
    ''' MultiHeadAttention.forward model.py:171-193 on top of nn.MultiheadAttention math
        (SURVEY App. A): [q;k;v] = x W_in^T + b_in, q scaled by 1/sqrt(d_h), pad keys -> -inf,
        softmax over keys, dropout on probs, out-proj, Dropout, +x, LayerNorm(128). '''
    B, N, E = x.shape
    d_h = E // nb_heads
    if OPERAND_DTYPE is None:
        qkv = x @ P[pre + 'multi_head_attention.in_proj_weight'].t() + P[pre + 'multi_head_attention.in_proj_bias']
    else:
        qkv = _stored_lp(linear_mfma(x, P[pre + 'multi_head_attention.in_proj_weight'], P[pre + 'multi_head_attention.in_proj_bias']))
    q, k, v = qkv.split(E, dim=-1)

    def heads(t):
        return t.reshape(B, N, nb_heads, d_h).permute(0, 2, 1, 3)  # (B,H,N,d)
    if OPERAND_DTYPE is None:
        q, k, v = heads(q) * (1. / math.sqrt(d_h)), heads(k), heads(v)
        s = q @ k.transpose(-1, -2)  # (B,H,N,N)
    else:   # the HIP kernels contract the bf16 q / k and apply the scale to the fp32 scores
        q, k, v = heads(q), heads(k), heads(v)
        s = _og(q @ k.transpose(-1, -2)) * (1. / math.sqrt(d_h))
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    if OPERAND_DTYPE is None:
        p = dropout(torch.softmax(s, dim=-1), p_drop, training)
        o = (p @ v).permute(0, 2, 1, 3).reshape(B, N, E)
        o = o @ P[pre + 'multi_head_attention.out_proj.weight'].t() + P[pre + 'multi_head_attention.out_proj.bias']
    else:
        # the HIP kernel is flash-style: keys in steps of 32 with a running row maximum; what is rounded to the operand type
        # is exp(s - running max) of that step (<= 1), later rescaled in fp32; the normaliser sums the unrounded values
        m_fin = s.max(dim=-1, keepdim=True).values
        n_steps = (N + 31) // 32
        s_pad = F.pad(s, (0, n_steps * 32 - N), value=float('-inf')).reshape(B, nb_heads, N, n_steps, 32)
        m_run = s_pad.max(dim=-1).values.cummax(dim=-1).values                          # (B,H,N,steps), finite from step 0 on
        m_key = m_run.unsqueeze(-1).expand(-1, -1, -1, -1, 32).reshape(B, nb_heads, N, n_steps * 32)[..., :N]
        p_step = _stored_lp(torch.exp(s - m_key)) * torch.exp(m_key - m_fin)
        p_step = dropout(p_step, p_drop, training)
        norm = torch.exp(s - m_fin).sum(dim=-1, keepdim=True)
        o = _stored_lp(((p_step @ v) / norm).permute(0, 2, 1, 3).reshape(B, N, E))
        o = linear_mfma(o, P[pre + 'multi_head_attention.out_proj.weight'], P[pre + 'multi_head_attention.out_proj.bias'])
    o = dropout(o, p_drop, training)
    return layer_norm(o + x, P[pre + 'layer_norm.weight'], P[pre + 'layer_norm.bias'])


def conv_ff(P, pre, x, film, p_drop, training):
    ''' PositionWiseConvFF.forward model.py:220-237: conv k3 -> ReLU -> conv k3 -> Dropout
        -> +x -> LayerNorm -> FiLM (gamma = film[:, :C], beta = film[:, C:]). '''
    h = torch.relu(conv1d_cl(x, P[pre + 'convs.0.conv.weight'], P[pre + 'convs.0.conv.bias']))
    z = conv1d_cl(h, P[pre + 'convs.2.conv.weight'], P[pre + 'convs.2.conv.bias'])
    u = layer_norm(dropout(z, p_drop, training) + x, P[pre + 'layer_norm.weight'], P[pre + 'layer_norm.bias'])
    if film is not None:
        C = film.shape[1] // 2
        assert C == u.shape[2]  # model.py:232
        u = film[:, None, :C] * u + film[:, None, C:]
    return u


def fft_block(P, pre, x, film, pad, cfg, training):
    ''' FFTBlock.forward model.py:251-264: attention, zero pads, FF(+FiLM), zero pads. '''
    a = multi_head_attention(P, pre + 'attention.', x, pad, cfg['attn_nb_heads'], cfg['attn_dropout'], training)
    a = a.masked_fill(pad.unsqueeze(2), 0.)
    u = conv_ff(P, pre + 'feed_forward.', a, film, cfg['conv_dropout'], training)
    return u.masked_fill(pad.unsqueeze(2), 0.)


# ----------------------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------------------
def film_layout(hp):
    ''' (name, nb_blocks, channels) in projection-column order.  model.py:322-326 '''
    return [('encoder', hp.phoneme_encoder['nb_blocks'], hp.phoneme_encoder['hidden_embed_dim']),
            ('prosody_predictor', hp.local_prosody_predictor['nb_blocks'], hp.local_prosody_predictor['conv_channels']),
            ('decoder', hp.frame_decoder['nb_blocks'], hp.phoneme_encoder['hidden_embed_dim'])]


def prosody_encoder(P, hp, frames_energy, frames_pitch, mel_specs, speaker_ids, output_lengths, training):
    ''' ProsodyEncoder.forward model.py:391-464 '''
    cfg, pre = hp.prosody_encoder, 'prosody_encoder.'
    D = cfg['hidden_embed_dim']
    pos = pos_encoding(output_lengths, D)
    energy = conv1d_cl(frames_energy.unsqueeze(2), P[pre + 'energy_embedding.conv.weight'], P[pre + 'energy_embedding.conv.bias'])
    pitch = conv1d_cl(frames_pitch.unsqueeze(2), P[pre + 'pitch_embedding.conv.weight'], P[pre + 'pitch_embedding.conv.bias'])
    x = mel_specs.transpose(1, 2)
    for conv_idx, ln_idx in ((0, 2), (4, 6), (8, 10)):  # nn.Sequential indices, model.py:341-363
        x = torch.relu(conv1d_cl(x, P[f'{pre}convs.{conv_idx}.conv.weight'], P[f'{pre}convs.{conv_idx}.conv.bias']))
        if x.shape[2] != D:
            x = _stored_lp(x)
        x = layer_norm(x, P[f'{pre}convs.{ln_idx}.weight'], P[f'{pre}convs.{ln_idx}.bias'])
        x = dropout(x, cfg['conv_dropout'], training)
    pad = ~valid_mask(output_lengths)
    x = (x + energy + pitch + pos).masked_fill(pad.unsqueeze(2), 0.)
    for blk in range(cfg['nb_blocks']):
        x = fft_block(P, f'{pre}blocks.{blk}.', x, None, pad, cfg, training)
    prosody_embeddings = x.sum(dim=1) / output_lengths.unsqueeze(1)
    z = prosody_embeddings + P[pre + 'spk_embedding.weight'][speaker_ids]
    gammas = z @ P[pre + 'gammas_predictor.linear_layer.weight'].t() + P[pre + 'gammas_predictor.linear_layer.bias']
    betas = z @ P[pre + 'betas_predictor.linear_layer.weight'].t() + P[pre + 'betas_predictor.linear_layer.bias']
    films, col, blk0 = [], 0, 0
    B = z.shape[0]
    for _, nb_blocks, channels in film_layout(hp):
        width = nb_blocks * channels
        g = gammas[:, col: col + width].reshape(B, nb_blocks, -1)
        bt = betas[:, col: col + width].reshape(B, nb_blocks, -1)
        if hp.post_mult_weight != 0.:
            post = P[pre + 'post_multipliers']
            g = post[0, blk0: blk0 + nb_blocks][None, :, None] * g + 1.
            bt = post[1, blk0: blk0 + nb_blocks][None, :, None] * bt
        else:
            g = g + 1.
        films.append(torch.cat((g, bt), dim=2))
        blk0 += nb_blocks
        col += width
    return (prosody_embeddings, *films)


def speaker_classifier(P, hp, x):
    ''' SpeakerClassifier.forward model.py:285-292 (GRL, 3 Linear, 2 ReLU) '''
    pre = 'speaker_classifier.classifier.'
    x = grad_reverse(x, hp.lambda_reversal)
    x = torch.relu(x @ P[pre + '1.linear_layer.weight'].t() + P[pre + '1.linear_layer.bias'])
    x = torch.relu(x @ P[pre + '3.linear_layer.weight'].t() + P[pre + '3.linear_layer.bias'])
    return x @ P[pre + '5.linear_layer.weight'].t() + P[pre + '5.linear_layer.bias']


def phoneme_encoder(P, hp, symbols, film, input_lengths, training):
    ''' PhonemeEncoder.forward model.py:490-509 '''
    cfg, pre = hp.phoneme_encoder, 'phoneme_encoder.'
    x = P[pre + 'symbols_embedding.weight'][symbols] + pos_encoding(input_lengths, cfg['hidden_embed_dim'])
    pad = ~valid_mask(input_lengths)
    x = x.masked_fill(pad.unsqueeze(2), 0.)
    for blk in range(cfg['nb_blocks']):
        x = fft_block(P, f'{pre}blocks.{blk}.', x, film[:, blk, :], pad, cfg, training)
    return x


def prosody_predictor(P, hp, x, film, input_lengths, training):
    ''' LocalProsodyPredictor.forward model.py:549-575 '''
    cfg, pre = hp.local_prosody_predictor, 'prosody_predictor.'
    for blk in range(cfg['nb_blocks']):
        for conv_idx, ln_idx in ((0, 2), (4, 6)):
            x = _stored_lp(torch.relu(conv1d_cl(x, P[f'{pre}blocks.{blk}.{conv_idx}.conv.weight'], P[f'{pre}blocks.{blk}.{conv_idx}.conv.bias'])))
            x = layer_norm(x, P[f'{pre}blocks.{blk}.{ln_idx}.weight'], P[f'{pre}blocks.{blk}.{ln_idx}.bias'])
            x = dropout(x, cfg['conv_dropout'], training)
        C = film.shape[2] // 2
        assert C == x.shape[2]  # model.py:561
        x = film[:, blk, None, :C] * x + film[:, blk, None, C:]
    pad = ~valid_mask(input_lengths)
    x = x.masked_fill(pad.unsqueeze(2), 0.)
    y = x @ P[pre + 'projection.linear_layer.weight'].t() + P[pre + 'projection.linear_layer.bias']
    y = y.masked_fill(pad.unsqueeze(2), 0.)
    return y[:, :, 0], y[:, :, 1], y[:, :, 2]


def gaussian_upsampling(P, hp, x, durations_float, durations_int, energies, pitch, input_lengths):
    ''' GaussianUpsamplingModule.forward model.py:608-662.  Integer part (cumsum of int64
        durations, T = max total) is exact; the Gaussian part is fp32. '''
    pre = 'gaussian_upsampling.'
    d = conv1d_cl(durations_float.unsqueeze(2), P[pre + 'duration_projection.conv.weight'], P[pre + 'duration_projection.conv.bias'])
    e = conv1d_cl(energies.unsqueeze(2), P[pre + 'energy_projection.conv.weight'], P[pre + 'energy_projection.conv.bias'])
    p = conv1d_cl(pitch.unsqueeze(2), P[pre + 'pitch_projection.conv.weight'], P[pre + 'pitch_projection.conv.bias'])
    x = x + e + p
    r = (x + d) @ P[pre + 'projection.0.linear_layer.weight'].t() + P[pre + 'projection.0.linear_layer.bias']
    ranges = F.softplus(r).squeeze(2)
    pad = ~valid_mask(input_lengths)
    ranges = ranges.masked_fill(pad, 1.)
    means = durations_int.float() / 2
    cumsum = torch.cumsum(durations_int, dim=1)
    means = torch.cat((means[:, :1], means[:, 1:] + cumsum[:, :-1]), dim=1)
    T = int(cumsum.max())
    t = torch.arange(T, dtype=torch.float) + 0.5
    mu, sigma = means.unsqueeze(-1), ranges.unsqueeze(-1)
    # Normal(mu, sigma).log_prob(t): -(t-mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi)
    log_prob = -((t - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi))
    probs = torch.exp(log_prob).masked_fill(pad.unsqueeze(2), 0.)
    weights = probs / (probs.sum(dim=1, keepdim=True) + 1e-20)  # (B, L, T)
    x_up = weights.transpose(1, 2) @ x  # (B, T, D)  == sum_l w[b,l,t] x[b,l,:]
    return x_up, weights


def frame_decoder(P, hp, x, film, output_lengths, training):
    ''' FrameDecoder.forward model.py:689-710; returns (B, n_mel, T). '''
    cfg, pre = hp.frame_decoder, 'frame_decoder.'
    D = hp.phoneme_encoder['hidden_embed_dim']
    pad = ~valid_mask(output_lengths)
    x = (x + pos_encoding(output_lengths, D)).masked_fill(pad.unsqueeze(2), 0.)
    for blk in range(cfg['nb_blocks']):
        x = fft_block(P, f'{pre}blocks.{blk}.', x, film[:, blk, :], pad, cfg, training)
    if OPERAND_DTYPE is None:
        mel = x @ P[pre + 'projection.linear_layer.weight'].t() + P[pre + 'projection.linear_layer.bias']
    else:
        mel = linear_mfma(x, P[pre + 'projection.linear_layer.weight'], P[pre + 'projection.linear_layer.bias'])
    return mel.masked_fill(pad.unsqueeze(2), 0.).transpose(1, 2)


# ----------------------------------------------------------------------------------------
# model entry points
# ----------------------------------------------------------------------------------------
def forward(P, hp, inputs, training=False):
    ''' DaftExprt.forward model.py:755-787 (teacher-forced). `training` switches dropout. '''
    symbols, durations_float, durations_int, symbols_energy, symbols_pitch, input_lengths, \
        frames_energy, frames_pitch, mel_specs, output_lengths, speaker_ids = inputs
    emb, enc_film, pp_film, dec_film = prosody_encoder(P, hp, frames_energy, frames_pitch, mel_specs,
                                                       speaker_ids, output_lengths, training)
    spk_preds = speaker_classifier(P, hp, emb)
    enc = phoneme_encoder(P, hp, symbols, enc_film, input_lengths, training)
    dur, energy, pitch = prosody_predictor(P, hp, enc, pp_film, input_lengths, training)
    x_up, weights = gaussian_upsampling(P, hp, enc, durations_float, durations_int, symbols_energy,
                                        symbols_pitch, input_lengths)
    mel = frame_decoder(P, hp, x_up, dec_film, output_lengths, training)
    post = P['prosody_encoder.post_multipliers'] if hp.post_mult_weight != 0. else 1.
    return spk_preds, [post, enc_film, pp_film, dec_film], [dur, energy, pitch, input_lengths], \
        [mel, output_lengths], weights


def duration_to_integer(float_durations, sampling_rate=22050, filter_length=1024, hop_length=256, centered=True):
    ''' extract_features.py:69-111 restated (nb_samples=None branch).  `float_durations` is a list
        of [begin, end] in seconds (Python doubles).  Raises IndexError when the utterance is
        shorter than one analysis window, ValueError on empty entries -- like the reference. '''
    spans = list(float_durations)
    total = 0  # sum() starts from int 0
    for b, e in spans:
        total = total + (e - b)
    nb_samples = int(total * sampling_rate)
    nb_frames = 1 + int((nb_samples - filter_length) / hop_length)  # int() truncates toward zero
    half = int(filter_length / 2)
    out, assigned, cursor = [], 0, 0
    while assigned + 1 <= nb_frames:
        if cursor >= len(spans):
            raise IndexError('pop from empty list')
        b, e = spans[cursor]
        cursor += 1
        if b == e:
            raise ValueError
        sb, se = int(b * sampling_rate), int(e * sampling_rate)
        # frame centres are half + hop*i, i in [0, nb_frames); count those in (sb, se]
        lo = max(0, -(-(sb + 1 - half) // hop_length))          # first i with centre > sb
        hi = min(nb_frames - 1, (se - half) // hop_length)      # last i with centre <= se
        n = max(0, hi - lo + 1)
        out.append(n)
        assigned += n
    if centered:
        edge = int(filter_length / 2 / hop_length)
        out[0] += edge  # IndexError on an empty list, as in the reference
        if cursor < len(spans):
            out.append(edge)
        else:
            out[-1] += edge
    return out


def get_int_durations(duration_preds, hp):
    ''' DaftExprt.get_int_durations model.py:789-812.  Mutates duration_preds in place
        (thresholding), returns (duration_preds, durations_int int64). '''
    dur_min = hp.filter_length / hp.sampling_rate / 2
    duration_preds[duration_preds < dur_min] = 0.
    out = torch.zeros(duration_preds.shape, dtype=torch.long)
    rows = duration_preds.tolist()  # float32 -> exact Python doubles, like .item()
    for b, row in enumerate(rows):
        end_prev, idx, spans = 0., [], []
        for l, dur in enumerate(row):
            if dur != 0.:
                idx.append(l)
                spans.append([end_prev, end_prev + dur])
                end_prev += dur
        ints = duration_to_integer(spans, hp.sampling_rate, hp.filter_length, hp.hop_length, hp.centered)
        out[b, idx] = torch.tensor(ints, dtype=torch.long)
    return duration_preds, out


def pitch_shift(pitch_preds, pitch_factors, hp, speaker_ids):
    ''' model.py:814-834 (exp / +Hz / log round trip, zeros restored). In place. '''
    zeros = pitch_preds == 0.
    for b in range(pitch_preds.shape[0]):
        stats = hp.stats[f'spk {int(speaker_ids[b])}']['pitch']
        mean, std = stats['mean'], stats['std']
        hz = torch.exp(std * pitch_preds[b] + mean) + pitch_factors[b]
        pitch_preds[b] = (torch.log(hz) - mean) / std
    pitch_preds[zeros] = 0.
    return pitch_preds


def pitch_multiply(pitch_preds, pitch_factors):
    ''' model.py:836-864 (scale the deviation from the voiced mean, zeros restored). In place. '''
    for b in range(pitch_preds.shape[0]):
        row = pitch_preds[b]
        zeros = row == 0.
        mean = row[~zeros].mean()
        row += (row - mean) * pitch_factors[b]
        row[zeros] = 0.
    return pitch_preds


def inference(P, hp, inputs, pitch_transform):
    ''' DaftExprt.inference model.py:866-923 '''
    symbols, dur_factors, energy_factors, pitch_factors, input_lengths, \
        energy_refs, pitch_refs, mel_spec_refs, ref_lengths, speaker_ids = inputs
    with torch.no_grad():
        _, enc_film, pp_film, dec_film = prosody_encoder(P, hp, energy_refs, pitch_refs, mel_spec_refs,
                                                         speaker_ids, ref_lengths, False)
        enc = phoneme_encoder(P, hp, symbols, enc_film, input_lengths, False)
        dur, energy, pitch = prosody_predictor(P, hp, enc, pp_film, input_lengths, False)
        dur, energy, pitch = dur.clone(), energy.clone(), pitch.clone()
        dur *= dur_factors
        dur, dur_int = get_int_durations(dur, hp)
        energy *= energy_factors
        energy[dur_int == 0] = 0.
        pitch[dur_int == 0] = 0.
        if pitch_transform == 'add':
            pitch = pitch_shift(pitch, pitch_factors, hp, speaker_ids)
        elif pitch_transform == 'multiply':
            pitch = pitch_multiply(pitch, pitch_factors)
        else:
            raise NotImplementedError
        x_up, weights = gaussian_upsampling(P, hp, enc, dur, dur_int, energy, pitch, input_lengths)
        output_lengths = dur_int.sum(dim=1)
        assert int(output_lengths.max()) == x_up.shape[1]  # model.py:914
        mel = frame_decoder(P, hp, x_up, dec_film, output_lengths, False)
    return [dur, dur_int, energy, pitch, input_lengths], [mel, output_lengths], weights


# ----------------------------------------------------------------------------------------
# loss, schedules, optimizer
# ----------------------------------------------------------------------------------------
def adversarial_weight(hp, iteration):
    ''' loss.py:22-28 '''
    w = iteration * hp.warmup_steps ** -1.5 * hp.adv_max_weight / hp.warmup_steps ** -0.5
    return min(hp.adv_max_weight, w)


def loss(hp, outputs, targets, iteration):
    ''' DaftExprtLoss.forward loss.py:30-106 -> (total, dict of the 7 weighted terms as tensors) '''
    dur_t, energy_t, pitch_t, mel_t, speaker_ids = targets
    spk_preds, film, enc_preds, dec_preds, _ = outputs
    dur, energy, pitch, input_lengths = enc_preds
    mel, output_lengths = dec_preds
    terms = {}
    terms['speaker_loss'] = adversarial_weight(hp, iteration) * F.cross_entropy(spk_preds, speaker_ids)
    if hp.post_mult_weight != 0.:
        terms['post_mult_loss'] = hp.post_mult_weight * torch.sqrt((film[0] ** 2).sum())
    else:
        terms['post_mult_loss'] = torch.zeros(())
    for name, w, pred, tgt in (('duration_loss', hp.dur_weight, dur, dur_t), ('energy_loss', hp.energy_weight, energy, energy_t),
                               ('pitch_loss', hp.pitch_weight, pitch, pitch_t)):
        terms[name] = w * (((pred - tgt) ** 2).sum(dim=1) / input_lengths).mean()
    denom = hp.n_mel_channels * output_lengths
    terms['mel_spec_l1_loss'] = hp.mel_spec_weight * ((mel - mel_t).abs().sum(dim=(1, 2)) / denom).mean()
    terms['mel_spec_l2_loss'] = hp.mel_spec_weight * (((mel - mel_t) ** 2).sum(dim=(1, 2)) / denom).mean()
    total = sum(terms[k] for k in ('speaker_loss', 'post_mult_loss', 'duration_loss', 'energy_loss', 'pitch_loss',
                                   'mel_spec_l1_loss', 'mel_spec_l2_loss'))
    return total, terms


def learning_rate(hp, iteration):
    ''' train.py:139-151 '''
    if iteration < hp.warmup_steps:
        return (hp.max_learning_rate - hp.initial_learning_rate) / hp.warmup_steps * iteration + hp.initial_learning_rate
    return iteration ** -0.5 * hp.max_learning_rate / hp.warmup_steps ** -0.5


def adam_step(params, grads, state, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6):
    ''' torch.optim.Adam as configured at train.py:299-301 (coupled L2, bias correction,
        denom = sqrt(v_hat) + eps, amsgrad off).  `state` = {'step', 'm': {...}, 'v': {...}}. '''
    state['step'] += 1
    t = state['step']
    b1, b2 = betas
    for name, p in params.items():
        g = grads[name] + weight_decay * p
        m = state['m'][name].mul_(b1).add_(g, alpha=1 - b1)
        v = state['v'][name].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


# ----------------------------------------------------------------------------------------
# collate contracts
# ----------------------------------------------------------------------------------------
def collate_train(items, n_mel=80):
    ''' DaftExprtDataCollate.__call__ data_loader.py:146-211.  items: list of
        [symbols(L) i64, dur_float(L), dur_int(L) i64, sym_energy(L), sym_pitch(L),
         frames_energy(T), frames_pitch(T), mel(n_mel,T), speaker_id, feature_dir, feature_file] '''
    lens = torch.tensor([len(it[0]) for it in items], dtype=torch.long)
    input_lengths, order = torch.sort(lens, dim=0, descending=True)
    B, L, T = len(items), int(input_lengths[0]), max(it[7].shape[1] for it in items)
    symbols = torch.zeros(B, L, dtype=torch.long)
    dur_f, dur_i = torch.zeros(B, L), torch.zeros(B, L, dtype=torch.long)
    s_energy, s_pitch = torch.zeros(B, L), torch.zeros(B, L)
    f_energy, f_pitch, mel = torch.zeros(B, T), torch.zeros(B, T), torch.zeros(B, n_mel, T)
    output_lengths, speaker_ids = torch.zeros(B, dtype=torch.long), torch.zeros(B, dtype=torch.long)
    dirs, files = [], []
    for row, src in enumerate(order.tolist()):
        it = items[src]
        l, t = len(it[0]), it[7].shape[1]
        symbols[row, :l], dur_f[row, :l], dur_i[row, :l] = it[0], it[1], it[2]
        s_energy[row, :l], s_pitch[row, :l] = it[3], it[4]
        f_energy[row, :t], f_pitch[row, :t], mel[row, :, :t] = it[5], it[6], it[7]
        output_lengths[row], speaker_ids[row] = t, it[8]
        dirs.append(it[9])
        files.append(it[10])
    return symbols, dur_f, dur_i, s_energy, s_pitch, input_lengths, f_energy, f_pitch, mel, \
        output_lengths, speaker_ids, dirs, files


def random_params(hp, seed=0):
    ''' Parameter dict with the reference's tensor names/shapes and its init
        families (xavier-uniform with the layer's gain; torch defaults elsewhere).
        Only used to drive the oracle on synthetic weights; goldens use `fill_params`. '''
    g = torch.Generator().manual_seed(seed)
    P = {}

    def xavier(shape, gain=1.):
        fan_out, fan_in = shape[0], shape[1]
        rf = 1
        for s in shape[2:]:
            rf *= s
        bound = gain * math.sqrt(6. / ((fan_in + fan_out) * rf))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def bias(n, fan_in):
        return (torch.rand(n, generator=g) * 2 - 1) / math.sqrt(fan_in)

    for name, shape in param_shapes(hp).items():
        if name.endswith('layer_norm.weight') or (len(shape) == 1 and name.endswith('.weight')):
            P[name] = torch.ones(shape)
        elif name.endswith('in_proj_bias') or name.endswith('out_proj.bias') or name.endswith('layer_norm.bias') \
                or (len(shape) == 1 and '.conv.' not in name and 'linear_layer' not in name):
            P[name] = torch.zeros(shape)
        elif len(shape) == 1:
            P[name] = bias(shape[0], 64)
        else:
            P[name] = xavier(shape, math.sqrt(2.) if ('convs.0' in name or 'classifier.1' in name) else 1.)
    return P


def param_shapes(hp):
    ''' Ordered {state_dict name: shape}; order = nn.Module registration order of the
        reference (model.py:718-725 and the sub-module constructors). '''
    S = {}
    n_mel, D = hp.n_mel_channels, hp.prosody_encoder['hidden_embed_dim']

    def fft_blocks(pre, cfg, E):
        for b in range(cfg['nb_blocks']):
            a = f'{pre}blocks.{b}.attention.'
            S[a + 'multi_head_attention.in_proj_weight'] = (3 * E, E)
            S[a + 'multi_head_attention.in_proj_bias'] = (3 * E,)
            S[a + 'multi_head_attention.out_proj.weight'] = (E, E)
            S[a + 'multi_head_attention.out_proj.bias'] = (E,)
            S[a + 'layer_norm.weight'] = (E,)
            S[a + 'layer_norm.bias'] = (E,)
            f = f'{pre}blocks.{b}.feed_forward.'
            C, K = cfg['conv_channels'], cfg['conv_kernel']
            S[f + 'convs.0.conv.weight'] = (C, E, K)
            S[f + 'convs.0.conv.bias'] = (C,)
            S[f + 'convs.2.conv.weight'] = (E, C, K)
            S[f + 'convs.2.conv.bias'] = (E,)
            S[f + 'layer_norm.weight'] = (E,)
            S[f + 'layer_norm.bias'] = (E,)

    pe, cfg = 'prosody_encoder.', hp.prosody_encoder
    C, K = cfg['conv_channels'], cfg['conv_kernel']
    if hp.post_mult_weight != 0.:
        S[pe + 'post_multipliers'] = (2, sum(nb for _, nb, _ in film_layout(hp)))
    for nm in ('energy_embedding', 'pitch_embedding'):
        S[f'{pe}{nm}.conv.weight'] = (D, 1, K)
        S[f'{pe}{nm}.conv.bias'] = (D,)
    for idx, (cin, cout) in zip((0, 4, 8), ((n_mel, C), (C, C), (C, D))):
        S[f'{pe}convs.{idx}.conv.weight'] = (cout, cin, K)
        S[f'{pe}convs.{idx}.conv.bias'] = (cout,)
        S[f'{pe}convs.{idx + 2}.weight'] = (cout,)
        S[f'{pe}convs.{idx + 2}.bias'] = (cout,)
    fft_blocks(pe, cfg, D)
    S[pe + 'spk_embedding.weight'] = (hp.n_speakers, D)
    nb_film = sum(nb * ch for _, nb, ch in film_layout(hp))
    for nm in ('gammas_predictor', 'betas_predictor'):
        S[f'{pe}{nm}.linear_layer.weight'] = (nb_film, D)
        S[f'{pe}{nm}.linear_layer.bias'] = (nb_film,)
    sc = 'speaker_classifier.classifier.'
    for idx, (i, o) in zip((1, 3, 5), ((D, D), (D, D), (D, hp.n_speakers - 1))):
        S[f'{sc}{idx}.linear_layer.weight'] = (o, i)
        S[f'{sc}{idx}.linear_layer.bias'] = (o,)
    E = hp.phoneme_encoder['hidden_embed_dim']
    S['phoneme_encoder.symbols_embedding.weight'] = (hp.n_symbols, E)
    fft_blocks('phoneme_encoder.', hp.phoneme_encoder, E)
    pp, cfg = 'prosody_predictor.', hp.local_prosody_predictor
    for b in range(cfg['nb_blocks']):
        cin = E if b == 0 else cfg['conv_channels']
        for idx, ci in ((0, cin), (4, cfg['conv_channels'])):
            S[f'{pp}blocks.{b}.{idx}.conv.weight'] = (cfg['conv_channels'], ci, cfg['conv_kernel'])
            S[f'{pp}blocks.{b}.{idx}.conv.bias'] = (cfg['conv_channels'],)
            S[f'{pp}blocks.{b}.{idx + 2}.weight'] = (cfg['conv_channels'],)
            S[f'{pp}blocks.{b}.{idx + 2}.bias'] = (cfg['conv_channels'],)
    S[pp + 'projection.linear_layer.weight'] = (3, cfg['conv_channels'])
    S[pp + 'projection.linear_layer.bias'] = (3,)
    gu, Kg = 'gaussian_upsampling.', hp.gaussian_upsampling_module['conv_kernel']
    for nm in ('duration_projection', 'energy_projection', 'pitch_projection'):
        S[f'{gu}{nm}.conv.weight'] = (E, 1, Kg)
        S[f'{gu}{nm}.conv.bias'] = (E,)
    S[gu + 'projection.0.linear_layer.weight'] = (1, E)
    S[gu + 'projection.0.linear_layer.bias'] = (1,)
    fft_blocks('frame_decoder.', hp.frame_decoder, E)
    S['frame_decoder.projection.linear_layer.weight'] = (n_mel, E)
    S['frame_decoder.projection.linear_layer.bias'] = (n_mel,)
    return S


def collate_inference(symbol_seqs, dur_factors, energy_factors, pitch_factors, pitch_transform, refs, speaker_ids,
                      file_names, n_mel=80):
    ''' generate.collate_tensors generate.py:140-239 after the text -> symbol-id step: sort by
        symbol count (descending), pad durations / energy factors with 1, pitch factors with
        0 ('add') or 1 ('multiply'), zero-pad references.  `refs` = list of (energy, pitch, mel). '''
    n = len(symbol_seqs)
    lens = torch.tensor([len(s) for s in symbol_seqs], dtype=torch.long)
    input_lengths, order = torch.sort(lens, dim=0, descending=True)
    L = int(input_lengths[0])
    neutral_pitch = 0. if pitch_transform == 'add' else 1.
    symbols = torch.zeros(n, L, dtype=torch.long)
    dur_f, en_f, pi_f = torch.ones(n, L), torch.ones(n, L), torch.full((n, L), neutral_pitch)
    T = max(r[2].shape[1] for r in refs)
    e_ref, p_ref, m_ref = torch.zeros(n, T), torch.zeros(n, T), torch.zeros(n, n_mel, T)
    ref_lengths, spk, names = torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long), []
    for row, src in enumerate(order.tolist()):
        l = len(symbol_seqs[src])
        symbols[row, :l] = torch.as_tensor(symbol_seqs[src], dtype=torch.long)
        if dur_factors[src] is not None:
            dur_f[row, :l] = torch.as_tensor(dur_factors[src], dtype=torch.float)
        if energy_factors[src] is not None:
            en_f[row, :l] = torch.as_tensor(energy_factors[src], dtype=torch.float)
        pi_f[row, :l] = neutral_pitch if pitch_factors[src] is None else torch.as_tensor(pitch_factors[src], dtype=torch.float)
        e, p, m = (torch.as_tensor(a).float() for a in refs[src])
        t = m.shape[1]
        e_ref[row, :t], p_ref[row, :t], m_ref[row, :, :t] = e, p, m
        ref_lengths[row], spk[row] = t, speaker_ids[src]
        names.append(file_names[src])
    return symbols, dur_f, en_f, pi_f, input_lengths, e_ref, p_ref, m_ref, ref_lengths, spk, names
