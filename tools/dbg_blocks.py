"""Development aid: where does the bf16 HIP forward leave the bf16-emulating oracle?  Per-block errors of the phoneme encoder.
usage: python tools/dbg_blocks.py BATCH T_MAX SEED"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')]
from oracle import daft_exprt_cpu as O   # noqa: E402
from tests.util import make_hparams, no_dropout   # noqa: E402

B, tmax, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
from daft_exprt.data_loader import synthetic_batch
from daft_exprt.model import DaftExprt
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
rel = lambda a, b: float((a.detach().float().cpu() - b.detach().float()).abs().max() / (b.abs().max() + 1e-12))
for emul in (torch.bfloat16, None):
    O.OPERAND_DTYPE = emul
    with torch.no_grad():
        emb, enc_film, pp_film, dec_film = O.prosody_encoder(P, hp, cin[6], cin[7], cin[8], cin[10], cin[9], True)
        cfg, pre = hp.phoneme_encoder, 'phoneme_encoder.'
        x = P[pre + 'symbols_embedding.weight'][cin[0]] + O.pos_encoding(cin[5], 128)
        pad = ~O.valid_mask(cin[5])
        x = x.masked_fill(pad.unsqueeze(2), 0.)
        print('emulation', emul, 'films', rel(outs[1][0], enc_film))
        for blk in range(4):
            s = S.enc[blk]
            print(f'  block {blk}: input {rel(s.x, x):.2e}', end='')
            a = O.multi_head_attention(P, f'{pre}blocks.{blk}.attention.', x, pad, cfg['attn_nb_heads'], 0., False).masked_fill(pad.unsqueeze(2), 0.)
            print(f'  attn+LN {rel(s.a, a):.2e}', end='')
            # feed the oracle's FF with the HIP path's own attention output to isolate the FF
            u_iso = O.conv_ff(P, f'{pre}blocks.{blk}.feed_forward.', s.a.float().cpu(), enc_film[:, blk, :], 0., False).masked_fill(pad.unsqueeze(2), 0.)
            x = O.conv_ff(P, f'{pre}blocks.{blk}.feed_forward.', a, enc_film[:, blk, :], 0., False).masked_fill(pad.unsqueeze(2), 0.)
            nxt = S.enc[blk + 1].x if blk < 3 else S.enc_out
            print(f'  block out {rel(nxt, x):.2e}  FF alone (HIP attention output in) {rel(nxt, u_iso):.2e}')
        d, e, p = O.prosody_predictor(P, hp, S.enc_out.float().cpu(), pp_film, cin[5], False)
        print('  predictor alone (HIP encoder output in):', rel(outs[2][0], d), rel(outs[2][1], e), rel(outs[2][2], p))
O.OPERAND_DTYPE = None
