"""Training loss of Daft-Exprt on one fused HIP kernel family.

Surface of the reference `DaftExprtLoss` (`src/daft_exprt/loss.py:6-106`): `DaftExprtLoss(gpu, hparams)`,
`update_adversarial_weight(iteration)`, `forward(outputs, targets, iteration) -> (loss, individual_loss)`
with the 7 keys `speaker_loss, post_mult_loss, duration_loss, energy_loss, pitch_loss, mel_spec_l1_loss,
mel_spec_l2_loss`.  The 7 terms AND their gradients come out of one pass of `dx_loss_fwd_bwd`
(the reference issues ~25 ATen ops and 7 `.item()` syncs here; this makes one D2H copy of 8 floats).
"""
import torch
from torch import nn

from daft_exprt import ops

KEYS = ('speaker_loss', 'post_mult_loss', 'duration_loss', 'energy_loss', 'pitch_loss', 'mel_spec_l1_loss', 'mel_spec_l2_loss')


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spk, dur, energy, pitch, mel, post, targets, lengths, weights):
        dur_t, energy_t, pitch_t, mel_t, spk_ids = targets
        in_len, out_len = lengths
        g = {'d_dur': torch.empty_like(dur), 'd_energy': torch.empty_like(energy), 'd_pitch': torch.empty_like(pitch),
             'd_mel': torch.empty_like(mel), 'd_spk': torch.empty_like(spk)}
        d_post = torch.zeros_like(post) if post is not None else None
        terms = ops.loss_fwd_bwd(dur.contiguous(), energy.contiguous(), pitch.contiguous(), dur_t, energy_t, pitch_t, in_len,
                                 mel.contiguous(), mel_t.contiguous(), out_len, spk.contiguous(), spk_ids,
                                 post.detach() if post is not None else None, weights, grads=g, d_post_mult=d_post)
        ctx.g, ctx.d_post = g, d_post
        ctx.mark_non_differentiable(terms)
        return terms[7].clone(), terms

    @staticmethod
    def backward(ctx, gout, _):
        g = ctx.g
        dp = ctx.d_post * gout if ctx.d_post is not None else None
        return g['d_spk'] * gout, g['d_dur'] * gout, g['d_energy'] * gout, g['d_pitch'] * gout, g['d_mel'] * gout, dp, \
            None, None, None


class DaftExprtLoss(nn.Module):
    def __init__(self, gpu, hparams):
        super(DaftExprtLoss, self).__init__()
        self.nb_channels = hparams.n_mel_channels
        self.warmup_steps = hparams.warmup_steps
        self.adv_max_weight = hparams.adv_max_weight
        self.post_mult_weight = hparams.post_mult_weight
        self.dur_weight = hparams.dur_weight
        self.energy_weight = hparams.energy_weight
        self.pitch_weight = hparams.pitch_weight
        self.mel_spec_weight = hparams.mel_spec_weight

    def update_adversarial_weight(self, iteration):
        ''' linear ramp to `adv_max_weight` over `warmup_steps` (`loss.py:22-28`) '''
        ramp = iteration * self.warmup_steps ** -1.5 * self.adv_max_weight / self.warmup_steps ** -0.5
        return min(self.adv_max_weight, ramp)

    def weights(self, iteration):
        return (self.update_adversarial_weight(iteration), self.post_mult_weight, self.dur_weight, self.energy_weight,
                self.pitch_weight, self.mel_spec_weight)

    def forward(self, outputs, targets, iteration):
        speaker_preds, film_params, encoder_preds, decoder_preds, _ = outputs
        post = film_params[0] if (self.post_mult_weight != 0. and torch.is_tensor(film_params[0])) else None
        dur, energy, pitch, input_lengths = encoder_preds
        mel, output_lengths = decoder_preds
        loss, terms = _LossFn.apply(speaker_preds, dur, energy, pitch, mel, post, tuple(targets),
                                    (input_lengths, output_lengths), self.weights(iteration))
        values = terms.tolist()   # the single D2H sync of the step (the reference does 7 `.item()` calls, loss.py:102-104)
        return loss, dict(zip(KEYS, values[:7]))
