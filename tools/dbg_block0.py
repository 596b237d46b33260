"""Development aid: block 0 of the phoneme encoder, HIP bf16 vs the bf16-emulating oracle, tensor by tensor."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')]
from oracle import daft_exprt_cpu as O   # noqa: E402
from tests.util import make_hparams, no_dropout   # noqa: E402
from daft_exprt.data_loader import synthetic_batch
from daft_exprt.model import DaftExprt
B, tmax, seed = 48, 1000, 1234
hp = no_dropout(make_hparams(speakers=[f's{i}' for i in range(11)], batch_size=B, accumulation_steps=1, compute_dtype='bf16'))
torch.manual_seed(hp.seed)
model = DaftExprt(hp)
P = {k: v.detach().clone() for k, v in model.state_dict().items()}
model = model.to('cuda:0').train()
cb = synthetic_batch(hp, B, seed=seed, t_max=tmax, force_first_full=True)
inputs, targets, _ = model.parse_batch('cuda:0', cb)
with torch.no_grad():
    outs, S = model._forward(inputs, True, True)
torch.cuda.synchronize()
cin = tuple(t.cpu() for t in inputs)


def stats(name, a, b, valid):
    a, b = a.detach().float().cpu()[valid], b.detach().float()[valid]
    d = (a - b).abs()
    print(f'{name:14s} max {float(d.max() / b.abs().max()):.2e}  mean {float(d.mean() / b.abs().mean()):.2e}  frac>1e-3*max {float((d > 1e-3 * b.abs().max()).float().mean()):.4f}')


O.OPERAND_DTYPE = torch.bfloat16
with torch.no_grad():
    pre = 'phoneme_encoder.blocks.0.'
    x = P['phoneme_encoder.symbols_embedding.weight'][cin[0]] + O.pos_encoding(cin[5], 128)
    valid = O.valid_mask(cin[5])
    pad = ~valid
    x = x.masked_fill(pad.unsqueeze(2), 0.)
    s = S.enc[0]
    stats('x', s.x, x, valid)
    E, H = 128, 2
    qkv = O._stored_lp(O.linear_mfma(x, P[pre + 'attention.multi_head_attention.in_proj_weight'], P[pre + 'attention.multi_head_attention.in_proj_bias']))
    stats('qkv', s.qkv, qkv, valid)
    q, k, v = qkv.split(E, dim=-1)
    heads = lambda t: t.reshape(B, -1, H, E // H).permute(0, 2, 1, 3)
    sc = (heads(q) @ heads(k).transpose(-1, -2)) / math.sqrt(E // H)
    sc = sc.masked_fill(pad[:, None, None, :], float('-inf'))
    p = torch.softmax(sc, dim=-1)
    lse = torch.logsumexp(sc, dim=-1)       # (B,H,N)
    stats('lse', s.lse.permute(0, 2, 1), lse.permute(0, 2, 1), valid)
    a_full = O.multi_head_attention(P, pre + 'attention.', x, pad, 2, 0., False).masked_fill(pad.unsqueeze(2), 0.)
    stats('a (bf16 copy)', s.a, O._op(a_full), valid)
    u_full = O.conv_ff(P, pre + 'feed_forward.', a_full, O.prosody_encoder(P, hp, cin[6], cin[7], cin[8], cin[10], cin[9], True)[1][:, 0, :], 0., False).masked_fill(pad.unsqueeze(2), 0.)
    stats('block-0 out', S.enc[1].x, O._op(u_full), valid)
    enc = O.phoneme_encoder(P, hp, cin[0], O.prosody_encoder(P, hp, cin[6], cin[7], cin[8], cin[10], cin[9], True)[1], cin[5], False)
    stats('encoder out', S.enc_out, enc, valid)
    y = O.prosody_predictor(P, hp, enc, O.prosody_encoder(P, hp, cin[6], cin[7], cin[8], cin[10], cin[9], True)[2], cin[5], False)
    stats('duration', outs[2][0], y[0], valid)
    y2 = O.prosody_predictor(P, hp, S.enc_out.cpu(), O.prosody_encoder(P, hp, cin[6], cin[7], cin[8], cin[10], cin[9], True)[2], cin[5], False)
    stats('duration | HIP enc', outs[2][0], y2[0], valid)
    a_h = O.layer_norm(s.s1.cpu(), P[pre + 'attention.layer_norm.weight'], P[pre + 'attention.layer_norm.bias'])
    stats('a | HIP s1', s.a, O._op(a_h), valid)
    h = torch.relu(O.conv1d_cl(s.a.float().cpu(), P[pre + 'feed_forward.convs.0.conv.weight'], P[pre + 'feed_forward.convs.0.conv.bias']))
    stats('h | HIP a', s.h, O._op(h), valid)
    z = O.conv1d_cl(s.h.float().cpu(), P[pre + 'feed_forward.convs.2.conv.weight'], P[pre + 'feed_forward.convs.2.conv.bias'])
    stats('s2 | HIP h,a', s.s2, z + a_h, valid)
O.OPERAND_DTYPE = None
