"""Fused Adam over the model's flat parameter buffer (one kernel for all 14.7 M parameters).

Same arithmetic as `torch.optim.Adam(params, betas, eps, weight_decay, amsgrad=False)` as configured by the
reference trainer (`src/daft_exprt/train.py:299-301`): coupled L2 (`g += wd * p`), bias correction,
`denom = sqrt(v_hat) + eps`; `clip_grad_norm_` (`train.py:399`) is folded into the same launch.
`state_dict()` / `load_state_dict()` speak torch.optim.Adam's checkpoint format (per-parameter `exp_avg`,
`exp_avg_sq`, `step`) so optimizer states round-trip with reference checkpoints (`train.py:73-78, 122-128`).
"""
import numpy as np
import torch

from daft_exprt import ops


class FusedAdam(object):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6, grad_clip_thresh=float('inf')):
        self.model = model
        flat = model.flat_parameters()
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.grad_norm_sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.step_count = 0
        self.grad_clip_thresh = grad_clip_thresh
        self.fuse_pack = True   # False: flat Adam launch, operand copies refreshed by the pack launches of the next step (tests compare the two)
        self.param_groups = [{'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': weight_decay, 'amsgrad': False,
                              'params': list(range(len(model._table)))}]

    def step(self):
        ''' returns the device scalar sum(grad^2) (sqrt = the gradient norm the reference logs) '''
        g = self.param_groups[0]
        self.step_count += 1
        flat, gflat = self.model.flat_parameters(), self.model.flat_gradients()
        table = self.model.adam_pack_table() if (self.fuse_pack and self.grad_clip_thresh == float('inf') and flat.is_cuda) else None
        if table is not None:
            # Adam + the refresh of every MFMA operand copy of the GEMM weights in ONE launch (the copies the next forward / backward pass
            # reads: forward, data-gradient and fragment-order packings): the weights are read once instead of three times, and the 5 pack
            # launches at the head of the next step disappear
            ops.H.check(ops.H.lib().dx_fill_zero(ops.H.ptr(self.grad_norm_sq), 4, ops.H.stream()))
            ops.adam_pack_step(flat, gflat, self.exp_avg, self.exp_avg_sq, table, self.model.cd, g['lr'], g['betas'], g['eps'],
                               g['weight_decay'], self.step_count, norm_accum=self.grad_norm_sq)
            self.model.mark_updated()
            self.model.packs_are_current()
            return self.grad_norm_sq
        if self.grad_clip_thresh == float('inf'):   # the norm is only logged (train.py:399): summed inside the Adam launch, no pass of its own
            ops.H.check(ops.H.lib().dx_fill_zero(ops.H.ptr(self.grad_norm_sq), 4, ops.H.stream()))
            ops.adam_step(flat, gflat, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'], g['eps'], g['weight_decay'],
                          self.step_count, None, self.grad_clip_thresh, norm_accum=self.grad_norm_sq)
        else:
            ops.sumsq(gflat, self.grad_norm_sq)
            ops.adam_step(flat, gflat, self.exp_avg, self.exp_avg_sq, g['lr'], g['betas'], g['eps'], g['weight_decay'],
                          self.step_count, self.grad_norm_sq, self.grad_clip_thresh)
        self.model.mark_updated()
        return self.grad_norm_sq

    # ---- per-section form: the Adam update of a gradient bucket runs as soon as that bucket is final (behind its all-reduce), on the
    # stream the caller chose, while the backward pass of the other modules is still running.  Only with an infinite clipping
    # threshold (the reference's setting, train.py:399: the norm is logged, never applied) -- a finite one needs the whole norm first.
    def sectioned(self):
        return self.grad_clip_thresh == float('inf')

    def begin_step(self):
        self.step_count += 1
        self._all_packed, self._covered = True, 0
        ops.H.check(ops.H.lib().dx_fill_zero(ops.H.ptr(self.grad_norm_sq), 4, ops.H.stream()))

    def step_slice(self, off, n):
        g = self.param_groups[0]
        flat, gflat = self.model.flat_parameters(), self.model.flat_gradients()
        table = self.model.adam_pack_table((off, n)) if (self.fuse_pack and flat.is_cuda) else None
        self._covered += n
        if table is not None:      # the slice's Adam + the refresh of the operand copies of the GEMM weights inside it, one launch
            ops.adam_pack_step(flat, gflat, self.exp_avg, self.exp_avg_sq, table, self.model.cd, g['lr'], g['betas'], g['eps'],
                               g['weight_decay'], self.step_count, norm_accum=self.grad_norm_sq)
            return
        if not (self.fuse_pack and flat.is_cuda and self.model._packed):   # (a slice WITHOUT GEMM weights has no copies to refresh)
            self._all_packed = False
        ops.adam_step(flat[off: off + n], gflat[off: off + n], self.exp_avg[off: off + n], self.exp_avg_sq[off: off + n], g['lr'],
                      g['betas'], g['eps'], g['weight_decay'], self.step_count, None, float('inf'), norm_accum=self.grad_norm_sq)

    def end_step(self):
        self.model.mark_updated()
        if self._all_packed and self._covered == self.model.flat_parameters().numel():
            self.model.packs_are_current()
        return self.grad_norm_sq

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def state_dict(self):
        state = {}
        for idx, (name, shape, _) in enumerate(self.model._table):
            off, n = self.model._offsets[name]
            state[idx] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.exp_avg[off: off + n].view(shape).clone(),
                          'exp_avg_sq': self.exp_avg_sq[off: off + n].view(shape).clone()}
        groups = [{k: v for k, v in self.param_groups[0].items()}]
        return {'state': state if self.step_count else {}, 'param_groups': groups}

    def load_state_dict(self, sd):
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in sd['param_groups'][0]:
                self.param_groups[0][k] = sd['param_groups'][0][k]
        self.param_groups[0]['betas'] = tuple(self.param_groups[0]['betas'])
        steps = set()
        for idx, (name, shape, _) in enumerate(self.model._table):
            st = sd['state'].get(idx)
            if st is None:
                continue
            off, n = self.model._offsets[name]
            self.exp_avg[off: off + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off: off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if steps:
            assert len(steps) == 1, 'per-parameter step counts differ'
            self.step_count = steps.pop()
