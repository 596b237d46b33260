"""Trainer of the MI355X-native Daft-Exprt.

Keeps the reference trainer's surface (`src/daft_exprt/train.py`): `update_learning_rate`, `save_checkpoint` /
`load_checkpoint` (same checkpoint dict keys, `train.py:73-78`; `module.`-prefixed state dicts accepted),
`train(gpu, hparams, log_file)`, `launch_training(...)` and the CLI flags `--data_set_dir --config_file
--benchmark_dir --log_file --world_size --rank --multiprocessing_distributed --master` (`train.py:612-634`).

Step loop = `train.py:368-401, 475-494` with the device work restructured for MI355X:
  * `model.forward_backward` (forward + fused 7-term loss + hand-written backward, no autograd graph);
  * gradient all-reduce over RCCL/xGMI overlapped with backward, once per optimizer step (`parallel.GradReducer`);
  * fused Adam over the flat parameter buffer, bucket by bucket behind each bucket's all-reduce on an optimizer stream, the
    logged gradient norm summed on the way (`optim.FusedAdam`, `Trainer`);
  * loss terms / gradient norm stay on the device; the scalars of iteration i reach the host (pinned copy on a side stream)
    after iteration i + 1 has been enqueued, so logging every iteration like the reference (`train.py:404-423`; it does
    8 `.item()` syncs per micro-batch, `loss.py:102-104`, `train.py:382`) costs no device idle time;
  * no per-iteration `dist.barrier()` (the reference's only paces log output, and its rank-local NaN test in front of it can
    deadlock): the gradient all-reduce keeps the ranks in step; validation and checkpoints keep their barriers.
Validation scoring and the best-model checkpoint (`train.py:427-456`) are kept; validation figures and the
benchmark-sentence synthesis (MFA, librosa, Griffin-Lim) are outside the accelerated path.
"""
import argparse
import json
import logging
import math
import os
import time

import torch
import torch.distributed as dist

from daft_exprt.data_loader import DaftExprtDataCollate, GroupedBatch, SyntheticUtterances, group_host_batches, \
    group_micro_batches, prepare_data_loaders
from daft_exprt.hparams import HyperParams
from daft_exprt.loss import DaftExprtLoss, KEYS
from daft_exprt.model import DaftExprt
from daft_exprt import config, ops, streams
from daft_exprt.optim import FusedAdam
from daft_exprt.parallel import GradReducer

_logger = logging.getLogger(__name__)
FEATURES_HPARAMS = ['centered', 'cutoff', 'f0_interval', 'filter_length', 'hop_length', 'language', 'mel_fmax', 'mel_fmin',
                    'min_clipping', 'max_f0', 'min_f0', 'n_mel_channels', 'order', 'sampling_rate', 'symbols', 'uv_cost',
                    'uv_interval']   # extract_features.py:26-28


def update_learning_rate(hparams, iteration):
    ''' linear warm-up from `initial_learning_rate` to `max_learning_rate`, then inverse-sqrt decay (`train.py:139-151`) '''
    lo, hi, warm = hparams.initial_learning_rate, hparams.max_learning_rate, hparams.warmup_steps
    if iteration < warm:
        return (hi - lo) / warm * iteration + lo
    return iteration ** -0.5 * hi / warm ** -0.5


def save_checkpoint(model, optimizer, hparams, learning_rate, iteration, best_val_loss=None, filepath=None):
    ''' same dict as `train.py:73-78`; state_dict keys carry the `module.` prefix when trained data-parallel so that
        reference consumers (`synthesize.py:43`, `fine_tune.py:40`) strip it as usual '''
    os.makedirs(os.path.dirname(filepath), exist_ok=True)
    _logger.info(f'Saving model and optimizer state at iteration "{iteration}" to "{filepath}"')
    prefix = 'module.' if dist.is_available() and dist.is_initialized() else ''   # what a DDP-wrapped model's state_dict carries
    state = {prefix + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    config = {k: v for k, v in vars(hparams).items()}
    torch.save({'iteration': iteration, 'learning_rate': learning_rate, 'best_val_loss': best_val_loss,
                'state_dict': state, 'optimizer': optimizer.state_dict(), 'config_params': config}, filepath)


def load_checkpoint(checkpoint_path, gpu, model, optimizer, hparams):
    ''' `train.py:81-136`: feature-extraction hyper-parameters must match (assert), other differences are warnings '''
    assert os.path.isfile(checkpoint_path), _logger.error(f'Checkpoint "{checkpoint_path}" does not exist')
    _logger.info(f'Loading checkpoint "{checkpoint_path}"')
    ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
    hp_ckpt = HyperParams(verbose=False, **ckpt['config_params'])
    for param in list(vars(hparams)):
        mine, theirs = getattr(hparams, param), getattr(hp_ckpt, param, None)
        if param in FEATURES_HPARAMS:
            assert mine == theirs, _logger.error(f'Parameter "{param}" is different between current config and the one used '
                                                 f'in checkpoint -- Was {theirs} in checkpoint and now is {mine}')
        elif not hasattr(hp_ckpt, param):
            _logger.warning(f'Parameter "{param}" exists in the current training config but did not exist in checkpoint config')
        elif mine != theirs:
            _logger.warning(f'Parameter "{param}" has changed -- Was {theirs} in checkpoint and now is {mine}')
    state = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in ckpt['state_dict'].items()}
    try:
        model.load_state_dict(state)
    except RuntimeError as e:
        _logger.error(f'Error when trying to load the checkpoint -- "{e}"\n')
    if len(ckpt['optimizer']['param_groups']) != len(optimizer.param_groups):
        _logger.warning('The optimizer in the loaded checkpoint does not have the same number of parameters '
                        'as the blank optimizer -- Creating a new optimizer.')
    else:
        optimizer.load_state_dict(ckpt['optimizer'])
    _logger.info(f'Loaded checkpoint "{checkpoint_path}" from iteration "{ckpt["iteration"]}"\n')
    return model, optimizer, ckpt['iteration'], ckpt['learning_rate'], ckpt['best_val_loss']


class Trainer(object):
    ''' the device-side body of one optimizer step (`train.py:368-401`) for one rank.

        The optimizer works per gradient BUCKET (the contiguous slices of `parallel.GradReducer`, in backward order): as soon as the
        backward pass reports a bucket final -- and, with several ranks, its RCCL all-reduce has finished -- the Adam update of
        that slice runs on a third stream (`_opt_stream`) under the rest of the backward pass; only the last bucket's update
        (the prosody encoder's pre-net, 15 MB) is left after the last backward kernel.  This needs the reference's infinite
        clipping threshold (`train.py:399` logs the norm, never applies it); with a finite one the whole-buffer step runs after
        the backward pass as before (also the default on one GPU, see `sectioned`). '''
    def __init__(self, model, hparams, world_size=1):
        self.model, self.hp, self.world = model, hparams, world_size
        self.criterion = DaftExprtLoss(0, hparams)
        self.optimizer = FusedAdam(model, betas=hparams.betas, eps=hparams.epsilon, weight_decay=hparams.weight_decay,
                                   grad_clip_thresh=hparams.grad_clip_thresh)
        self.reducer = GradReducer(model)     # also on one rank: its bucket table drives the per-section optimizer
        if self.reducer.active:
            if model.flat_parameters().is_cuda and config.WGRAD_SIDE_STREAM:
                # launch stream, weight-gradient stream and RCCL's stream on three different hardware queues (probed, `streams.py`)
                self.reducer.pick_group([torch.cuda.current_stream(), model.ensure_side_stream()])
            self.reducer.broadcast_parameters()
        model.always_repack = False   # parameters only change through self.optimizer
        self.terms = None
        # the `accumulation_steps` micro-batches of an optimizer step as ONE pass over their concatenation, every utterance keeping the
        # padded length of its own micro-batch as a hard sequence end (`data_loader.GroupedBatch`): the reference's accumulation
        # (`train.py:379-401`, default 16 x 3) at the launch count and tile efficiency of one batch of 48.  hparams.group_micro_batches
        # = False keeps one pass per micro-batch.
        self.group = bool(getattr(hparams, 'group_micro_batches', True))
        self._groups = {}
        # per-bucket Adam behind each bucket's all-reduce: default on with several ranks (it hides the optimizer pass and the wait for
        # the last all-reduce); on ONE GPU the slice updates only compete with the backward kernels for HBM (measured 8.29 vs 8.08 ms
        # per step), so the whole-buffer step (one launch, gradient norm summed on the way) stays the default there
        mode = config.sectioned_adam()
        assert self.reducer.world == world_size, f'Trainer(world_size={world_size}) inside a process group of {self.reducer.world} ranks'
        # the per-bucket update orders itself behind the collective through `work.wait()`, which is a STREAM wait only with RCCL
        # (backend "nccl"); gloo's wait blocks the host inside the backward hook and would stall kernel issue for the rest of the
        # backward pass, so any other backend keeps the whole-buffer step
        stream_ordered = self.reducer.active and dist.get_backend() == 'nccl' and \
            not int(os.environ.get('TORCH_NCCL_BLOCKING_WAIT', '0') or 0)
        self.sectioned = stream_ordered if mode == 'auto' else bool(int(mode))
        self._opt_stream = self._sec_event = None
        self._done = set()
        self.diag = self._diag_comm = None          # a list: step_eager appends (event behind the last backward kernel, event behind the last all-reduce)
        # whole-step hipGraphs, keyed on the addresses and shapes of the step's input tensors (`CapturedStep`): one rank only -- a
        # captured RCCL collective is untested here -- and only with the whole-buffer optimizer
        # DX_STEP_GRAPH: 0 (default) = never, 1 = every repeating step is captured, auto = only steps small enough to be bound by the
        # host's launch rate (`CapturedStep.AUTO_ROWS`).  Opt-in because, measured on MI355X / ROCm 7.0, a replay only TIES the eager step
        # from B = 16 up (7.73 vs 7.69 ms at B = 48, 13.2 vs 13.0 ms for 16 x 3): the dispatch boundaries cost the same ~2 us on the GPU
        # side whether the host or the graph executor feeds them, and at these sizes the host keeps ahead of the device.  It wins below
        # (3.88 vs 4.05 ms at B = 8).
        use_graph = config.step_graph()
        self.captured = CapturedStep(self, auto=(use_graph == 'auto')) if (not self.reducer.active and use_graph != '0' and
                                                                           model.flat_parameters().is_cuda) else None

    def _section_done(self, name):
        ''' backward hook (runs with the weight-gradient side stream current, after it has caught up with the compute stream):
            all-reduce of the bucket this section closes, then its Adam update behind the collective on the optimizer stream '''
        hit = self.reducer.section_done(name)
        if hit is None or not self._sectioned_now:
            return
        off, n, work = hit
        cur = torch.cuda.current_stream()
        self._sec_event.record(cur)
        with torch.cuda.stream(self._opt_stream):
            self._opt_stream.wait_event(self._sec_event)      # the bucket's gradients are final on the stream that reported them
            if work is not None:
                work.wait()                                    # stream-ordered: the optimizer stream waits for the collective
            if self.diag is not None:                          # (bench.py --gpus N: when did this bucket's collective end?)
                self._diag_comm = torch.cuda.Event(enable_timing=True)
                self._diag_comm.record(self._opt_stream)
            self.optimizer.step_slice(off, n)
        self._done.add(name)

    def _grouped(self, micro_batches):
        ''' the GroupedBatch of these micro-batches, or None when they cannot run as one pass (micro-batches of different sizes: the
            group's per-utterance mean would weigh an utterance 1 / B_total instead of the reference's 1 / (accum B_k)).
            Resident batches are grouped once: the cache is keyed on the address AND the version counter of every source tensor (inputs
            and targets), so a caller that refills its staging buffers in place gets a fresh group '''
        if len({mb[0][0].shape[0] for mb in micro_batches}) != 1:
            return None
        srcs = [t for mb in micro_batches for part in mb for t in part]
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        hit = self._groups.get(key)
        if hit is not None and all(x is y for x, y in zip(hit[0], srcs)):
            return hit[1]
        g = group_micro_batches(micro_batches)
        if len(self._groups) >= 8:
            self._groups.pop(next(iter(self._groups)))
        self._groups[key] = (srcs, g)     # (keeps the source tensors alive: their addresses are part of the key)
        return g

    def step(self, micro_batches, iteration):
        ''' micro_batches: list of (inputs, targets) already on the device (len = accumulation_steps), or one
            `data_loader.GroupedBatch` in a list (what `train()` builds on the host before the H2D copy).
            Returns (terms (8,) device tensor summed over micro-batches / accumulation_steps, grad_norm_sq device scalar).
            The tensors may live in buffers a later call overwrites (captured steps reuse theirs): consume them -- or enqueue the
            copy that does -- before the next call. '''
        if self.group and len(micro_batches) > 1 and self.model.flat_parameters().is_cuda:
            g = self._grouped(micro_batches)                   # (outside a capture: the concatenation is input staging)
            micro_batches = [g] if g is not None else micro_batches
        if self.captured is not None and not self.reducer.active and not self.sectioned:
            return self.captured.step(micro_batches, iteration)
        return self.step_eager(micro_batches, iteration)

    def step_eager(self, micro_batches, iteration):
        ''' the step as individual launches (also what a capture records) '''
        hp, model = self.hp, self.model
        if self.group and len(micro_batches) > 1 and model.flat_parameters().is_cuda:
            g = self._grouped(micro_batches)
            micro_batches = [g] if g is not None else micro_batches     # ragged micro-batches: one pass each, as the reference
        accum = len(micro_batches)
        lr = update_learning_rate(hp, iteration)
        self.optimizer.param_groups[0]['lr'] = lr
        weights = self.criterion.weights(iteration)
        scale = 1. / (accum * self.world)   # loss / accumulation_steps (train.py:379) and the DDP mean over ranks
        self._sectioned_now = self.sectioned and self.optimizer.sectioned()
        if self._sectioned_now:
            if self._opt_stream is None:
                # on hardware queues of their own (probed): beside the launch stream and the weight-gradient stream
                side = model.ensure_side_stream() if config.WGRAD_SIDE_STREAM else None
                beside = [torch.cuda.current_stream()] + ([side] if side is not None else [])
                self._opt_stream, self._sec_event = streams.pick(beside, what='optimizer stream'), torch.cuda.Event()
            self.optimizer.begin_step()     # step count, zeroed norm accumulator: on the compute stream, ahead of every slice update
            self._done = set()
        total = None
        for k, mb in enumerate(micro_batches):
            inputs, targets = mb
            last = k == accum - 1
            hook = self._section_done if (last and (self.reducer.active or self._sectioned_now)) else None
            terms = model.forward_backward(inputs, targets, weights, grad_scale=scale, section_done=hook,
                                           bounds=mb.bounds if isinstance(mb, GroupedBatch) else None)
            total = terms if total is None else ops.add_(total, terms)
        main = torch.cuda.current_stream()
        if self.diag is not None:                   # the last backward kernel of the launch stream is queued: an event behind it
            bwd_end = torch.cuda.Event(enable_timing=True)
            bwd_end.record(main)
        if self._sectioned_now:
            assert self._done == set(sec for sec, _, _ in self.reducer.buckets), 'a gradient bucket was never reported'
            main.wait_stream(self._opt_stream)      # every slice update (and with it every all-reduce) is behind us
            self.reducer._works = []
            gnorm_sq = self.optimizer.end_step()
        else:
            self.reducer.wait()
            if self.diag is not None:
                self._diag_comm = torch.cuda.Event(enable_timing=True)
                self._diag_comm.record(main)        # (whole-buffer optimizer: the launch stream has waited for every collective)
            gnorm_sq = self.optimizer.step()
        if self.diag is not None and self._diag_comm is not None:
            self.diag.append((bwd_end, self._diag_comm))   # elapsed(bwd_end -> end of the last all-reduce) = communication the backward did not hide
            self._diag_comm = None
        model.zero_grad()
        if accum > 1:
            ops.scale_(total, 1. / accum)
        self.terms = total
        return self.terms, gnorm_sq


class CapturedStep(object):
    ''' `Trainer.step` as ONE hipGraph launch per optimizer step.

        A step is ~330 C-ABI calls from Python (~3.3 ms of host time at B = 48, ~10 ms for the reference's 16 x 3 schedule, whose
        kernels are a third the size): the phoneme-level stretches of a B = 48 step and the WHOLE 16 x 3 step are host-bound.  A
        captured step costs the host one small launch (the step block) + one graph launch.

        What makes the step capturable: every per-step scalar lives in device memory (`DxStepScalars`: dropout salt, learning
        rate, Adam's bias corrections, the adversarial loss weight -- `ops.STEP_PTR`), there is no host sync inside the step, the
        side-stream weight gradients fork from and join the launch stream inside the step, and every buffer the step allocates
        comes from the graph's private pool.  A graph is keyed on the ADDRESSES and shapes of the step's input tensors (it reads
        them in place: no staging copies; the graph pins them, so a key can only come back when the CALLER re-uses the same tensors)
        and on the accumulation count; a key is captured the second time it is seen.  What replays are RESIDENT batches (bench, tests,
        a caller that stages every batch into fixed input buffers); batches from a loader get fresh allocations and simply stay on
        the eager path.  At most `max_graphs` graphs are kept, each
        owning its activations (~4 GB at B = 48, T = 1000); a full cache only gives up its least recently replayed graph when that
        graph has been idle for 4 x max_graphs steps (a round-robin over more keys than slots would otherwise re-capture every step).

        Replays are bit-identical to eager steps with the same step ids (`tests/test_gpu_captured_step.py`). '''
    AUTO_ROWS = 12288     # auto mode: capture only while B x T_max of every micro-batch is below this many frame rows

    def __init__(self, trainer, max_graphs=None, capture_after=2, auto=False):
        self.tr, self.auto = trainer, auto
        self.max_graphs = config.STEP_GRAPH_MAX if max_graphs is None else max_graphs
        self.capture_after = capture_after
        self.cache, self.seen = {}, {}
        self.block = None            # DxStepScalars on the device
        self.replays = self.captures = self.eager_steps = self.ticks = 0
        self.broken = None

    @staticmethod
    def key(micro_batches):
        # (targets too: the loss reads them in place -- normally views of the inputs, but nothing forces a caller to pass those)
        return tuple((t.data_ptr(), tuple(t.shape)) for mb in micro_batches for t in CapturedStep._tensors(mb))

    @staticmethod
    def _tensors(mb):
        ''' every tensor a captured step reads in place: inputs, targets and -- grouped micro-batches -- the per-utterance bounds '''
        return tuple(mb[0]) + tuple(mb[1]) + (tuple(mb.bounds) if isinstance(mb, GroupedBatch) else ())

    def step(self, micro_batches, iteration):
        tr = self.tr
        self.ticks += 1
        if self.broken is not None or ops.PROBE is not None or tr.model._trace is not None or tr.model._trace_bwd is not None or \
                (self.auto and max(mb[0][8].shape[0] * mb[0][8].shape[2] for mb in micro_batches) >= self.AUTO_ROWS):
            self.eager_steps += 1
            return tr.step_eager(micro_batches, iteration)
        key = self.key(micro_batches)
        ent = self.cache.get(key)
        if ent is None:
            n = self.seen[key] = self.seen.get(key, 0) + 1
            if len(self.seen) > 4096:
                self.seen = {key: n}
            full = len(self.cache) >= self.max_graphs
            if n < self.capture_after or (full and self.ticks - min(e['tick'] for e in self.cache.values()) < 4 * self.max_graphs):
                self.eager_steps += 1
                return tr.step_eager(micro_batches, iteration)
            ent = self._capture(key, micro_batches, iteration)
            if ent is None:
                self.eager_steps += 1
                return tr.step_eager(micro_batches, iteration)
        return self._replay(ent, len(micro_batches), iteration)

    def _scalars(self, iteration):
        ''' this iteration's scalars into the step block (one single-thread launch in front of the graph) '''
        tr = self.tr
        opt, model = tr.optimizer, tr.model
        lr = update_learning_rate(tr.hp, iteration)
        opt.param_groups[0]['lr'] = lr
        opt.step_count += 1
        ops.step_scalars_set(self.block, model.step_salt(), lr, opt.param_groups[0]['betas'], opt.step_count,
                             tr.criterion.weights(iteration)[0])

    def _replay(self, ent, accum, iteration):
        tr = self.tr
        self._scalars(iteration)
        tr.model._step_id += accum             # what the captured forward passes would have counted
        ent['graph'].replay()
        tr.model.mark_updated()                # the parameters changed on the device: an eager call that follows re-packs its operand copies
        self.replays += 1
        ent['tick'] = self.ticks
        tr.terms, tr.model.last_outputs = ent['terms'], ent['outputs']
        return ent['terms'], ent['gnorm_sq']

    def prepare(self, micro_batches, iteration=1):
        ''' set-up call: make the graph of this step exist NOW (an eager step first, so that every kernel is loaded and the host-side
            caches are built -- a capture executes nothing -- then the capture), instead of on the second time `step` sees the key.
            Returns True when a graph for the key is cached afterwards. '''
        key = self.key(micro_batches)
        if key not in self.cache and self.broken is None and len(self.cache) < self.max_graphs:
            self.tr.step_eager(micro_batches, iteration)
            self._capture(key, micro_batches, iteration)
        return key in self.cache

    def _capture(self, key, micro_batches, iteration):
        tr = self.tr
        model, opt = tr.model, tr.optimizer
        dev = model.flat_parameters().device
        if self.block is None:
            self.block = torch.zeros(48, dtype=torch.uint8, device=dev)
        while len(self.cache) >= self.max_graphs:           # least recently replayed graph (and its pool) goes
            del self.cache[min(self.cache, key=lambda k: self.cache[k]['tick'])]
        # host-side state the captured launches advance WITHOUT executing: put back afterwards
        step_id, step_count, lr = model._step_id, opt.step_count, opt.param_groups[0]['lr']
        eager_ws, model._wgrad_ws = model._wgrad_ws, None    # the graph gets scratch of its own (it must outlive every replay)
        own_ws = {}
        model.mark_updated()                                  # the graph always begins by re-packing the operand copies
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        # no cyclic garbage collection while the stream is capturing: a collected hipGraph / event of an EARLIER trainer runs runtime calls
        # in its destructor that are illegal during a capture (the process aborted inside `Garbage-collecting` in the test suite)
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            ops.STEP_PTR, ops.ATTN_WS_OWNER, model._capture_step0 = self.block.data_ptr(), own_ws, step_id
            with torch.cuda.graph(graph):
                terms, gnorm_sq = tr.step_eager(micro_batches, iteration)
        except Exception as e:   # noqa: BLE001 -- a capture that cannot be recorded must not take the training run down
            self.broken = f'{type(e).__name__}: {e}'
            _logger.warning(f'step capture failed, staying on eager launches: {self.broken}')
            torch.cuda.synchronize(dev)
            # stream-side state the aborted capture touched: events recorded on a capturing stream are invalid outside it, the queued
            # weight gradients and their pinned operands belong to launches that never ran
            model._wgrad_pending, model._side_deferred = [], None
            model._wgrad_keep.clear()
            model._hop = torch.cuda.Event() if model._hop is not None else None
            tr._sec_event = torch.cuda.Event() if tr._sec_event is not None else None
            opt._all_packed, opt._covered = True, 0
            return None
        finally:
            if gc_was_on:
                gc.enable()
            ops.STEP_PTR, ops.ATTN_WS_OWNER, ops.H.AFTER_LAUNCH, model._side_deferred = None, None, None, None
            graph_ws, model._wgrad_ws = model._wgrad_ws, eager_ws
            model._step_id, opt.step_count, opt.param_groups[0]['lr'] = step_id, step_count, lr
            model.mark_updated()
        self.captures += 1
        ent = self.cache[key] = {'graph': graph, 'terms': terms, 'gnorm_sq': gnorm_sq, 'tick': self.ticks, 'outputs': model.last_outputs,
                                 'keep': (own_ws, graph_ws, [t for mb in micro_batches for t in self._tensors(mb)])}
        return ent


def validate(gpu, model, criterion, val_loader, hparams):
    ''' `train.py:193-233`: eval mode, no grad, criterion at iteration 0 (adversarial weight 0), batch
        means of the total and of the five reconstruction terms.  Returns (val_loss, val_indiv_loss); the per-batch
        targets / outputs the reference also returns only feed its TensorBoard figures (out of scope). '''
    val_loss, n = 0., 0
    indiv = {k: 0. for k in KEYS[2:]}
    model.eval()
    with torch.no_grad():
        for batch in val_loader:
            inputs, targets, _ = model.parse_batch(gpu, batch)
            loss, terms = criterion(model(inputs), targets, iteration=0)
            val_loss += float(loss)
            for k in indiv:
                indiv[k] += terms[k]
            n += 1
    model.train()
    return val_loss / max(n, 1), {k: v / max(n, 1) for k, v in indiv.items()}


def _loaders(hparams, rank, world, distributed):
    ''' (train loader, validation loader or None): on-disk features when `training_files` exists, else the seeded synthetic
        utterances.  The sampler keys on the live process group (one predicate everywhere), not on the hparams flag. '''
    collate = DaftExprtDataCollate(hparams)
    if os.path.isfile(str(hparams.training_files)):
        train_loader, _, val_loader, _ = prepare_data_loaders(hparams, num_workers=8, distributed=distributed)
        return train_loader, val_loader
    n_items = getattr(hparams, 'synthetic_items', hparams.batch_size * hparams.accumulation_steps * world * 8)
    ds = SyntheticUtterances(hparams, n_items, seed=hparams.seed, force_first_full=False)
    idx = list(range(rank, n_items, world))   # DistributedSampler(shuffle=False) striding (data_loader.py:232)
    subset = torch.utils.data.Subset(ds, idx)
    # an utterance costs ~2 ms of numpy (100 ms per batch of 48): 16 processes feed an 8 ms step; with several ranks on one host the
    # ranks share its cores.  The workers come from a fork SERVER: forking a process that already holds HIP / RCCL threads is fragile
    workers = int(getattr(hparams, 'synthetic_workers', max(1, min(16, (os.cpu_count() or 16) // max(1, world)))))
    return torch.utils.data.DataLoader(subset, batch_size=hparams.batch_size, shuffle=False, drop_last=True, collate_fn=collate,
                                       num_workers=workers, pin_memory=True, persistent_workers=workers > 0,
                                       prefetch_factor=4 if workers > 0 else None,
                                       multiprocessing_context='forkserver' if workers > 0 else None), None


def train(gpu, hparams, log_file):
    ''' one rank of the training job (`train.py:236-494`).  Like the reference, the process group exists whenever
        `multiprocessing_distributed` is set -- also with a single GPU -- and everything downstream (sampler, gradient
        reducer, `module.` checkpoint prefix, barriers) keys on that one fact. '''
    distributed = bool(getattr(hparams, 'multiprocessing_distributed', False))
    if distributed:
        hparams.rank = hparams.rank * hparams.ngpus_per_node + gpu
        os.environ.setdefault('NCCL_DEBUG', 'VERSION')     # RCCL prints its version banner once: the log shows which library ran
        dist.init_process_group(backend=hparams.dist_backend, init_method=hparams.dist_url, world_size=hparams.world_size,
                                rank=hparams.rank)
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    os.makedirs(os.path.dirname(os.path.abspath(log_file)), exist_ok=True)
    logging.basicConfig(handlers=[logging.StreamHandler(), logging.FileHandler(log_file)],
                        format='%(asctime)s [%(levelname)s] %(message)s', datefmt='%Y-%m-%d %H:%M:%S',
                        level=logging.INFO if rank == 0 else logging.ERROR)
    torch.cuda.set_device(gpu)
    torch.manual_seed(hparams.seed)
    model = DaftExprt(hparams).cuda(gpu)
    model.set_rank(rank)                      # per-rank dropout streams
    model.train()
    trainer = Trainer(model, hparams, world)
    _logger.info(trainer.reducer.describe())   # rank 0: world size, backend and the bucket sizes (proof of the RCCL world in a log)
    criterion = trainer.criterion
    iteration, best_val_loss = 1, float('inf')
    if hparams.checkpoint != '':
        model, trainer.optimizer, iteration, _, best_val_loss = load_checkpoint(hparams.checkpoint, gpu, model, trainer.optimizer, hparams)
        iteration += 1
    loader, val_loader = _loaders(hparams, rank, world, distributed)
    _logger.info(f'Batch size: {hparams.batch_size * hparams.accumulation_steps * world:_}')
    metrics_path = os.path.join(os.path.dirname(log_file), 'metrics.jsonl')
    ckpt_dir = os.path.join(hparams.output_directory, 'checkpoints')
    start, total_time = time.time(), 0.
    model.zero_grad()
    micro = []
    # Per-iteration reporting (`train.py:404-423`) WITHOUT stalling the device: the 9 scalars of iteration i (7 loss terms, total,
    # gradient norm) are copied to pinned memory on a side stream behind step i, and read by the host after step i + 1 has been
    # enqueued -- the device always has the next step queued while the host formats the log line of the previous one.  The
    # reference's per-iteration `dist.barrier()` (it only paces the ranks' log output) and the round-1 NaN all-reduce are gone:
    # nothing in the step depends on them, and the gradient all-reduce already keeps the ranks in lockstep.
    stats_host = torch.empty(9, dtype=torch.float32).pin_memory()
    # (both helper streams on hardware queues other than the launch stream's: probed, `streams.pick`)
    log_stream, stats_ready, step_done = streams.pick([torch.cuda.current_stream(gpu)], what='log stream'), torch.cuda.Event(), torch.cuda.Event()
    pending = None        # (iteration, lr, valid frames) of the step whose scalars are in flight

    def report(now):
        ''' log the pending iteration; returns False when its loss was NaN (the reference then logs nothing, train.py:404) '''
        nonlocal pending, total_time, start
        if pending is None:
            return True
        it_p, lr_p, frames_p = pending
        pending = None
        stats_ready.synchronize()            # waits for the END OF THE PREVIOUS step only
        values = stats_host.tolist()
        tot_loss, grad_norm = values[7], values[8]
        duration = now - start               # host-enqueue interval of the iteration (the device runs one step behind the host)
        start = now
        if not math.isfinite(tot_loss):
            return False
        if rank == 0:
            total_time += duration
            _logger.info(f'Train loss [{it_p}]: {tot_loss:.6f} Grad Norm {grad_norm:.6f} {duration:.2f}s/it (LR {lr_p:.6f})')
            with open(metrics_path, 'a') as f:   # scalar names of DaftExprtLogger.log_training (logger.py:26-32)
                rec = {'iteration': it_p, 'DaftExprt.optimization/grad_norm': grad_norm,
                       'DaftExprt.optimization/learning_rate': lr_p, 'DaftExprt.optimization/duration': duration,
                       'DaftExprt.training/loss': tot_loss, 'valid_frames': frames_p}
                rec.update({f'DaftExprt.training/{k}': v for k, v in zip(KEYS, values[:7])})
                f.write(json.dumps(rec) + '\n')
        return True

    copy_stream = streams.pick([torch.cuda.current_stream(gpu)], what='copy stream')

    group = trainer.group and hparams.accumulation_steps > 1

    held = []     # micro-batches of an unfinished group: carried over an epoch boundary like the reference's accumulation counter (train.py:391-397)

    def host_units():
        ''' collate outputs, one per forward / backward pass: the loader's batches, or -- grouped micro-batches -- the
            `accumulation_steps` batches of an optimizer step merged on the host (`group_host_batches`) '''
        for batch in loader:
            if not group:
                yield batch, None
                continue
            held.append(batch)
            if len(held) == hparams.accumulation_steps:
                merged, nmax, sizes = group_host_batches(held)
                # (the train loader drops its last partial batch, data_loader.py:238; a ragged group would weigh utterances 1 / B_total
                # instead of the reference's 1 / (accum B_k) -- ADVICE r5)
                assert len(set(sizes)) == 1, f'micro-batches of different sizes {sizes}: set hparams.group_micro_batches = False'
                held.clear()
                yield merged, (nmax, sizes)

    def device_batches():
        ''' the loader's batches one ahead of the step that consumes them: the H2D copies of batch i + 1 (15.8 MB at B = 48) run on
            a copy stream under step i instead of in front of step i + 1 on the compute stream '''
        ahead = None
        for batch, grouped in host_units():
            frames = int(batch[9].sum())     # host tensor (collate output): no device round trip
            with torch.cuda.stream(copy_stream):
                inputs, targets, _ = model.parse_batch(gpu, batch)
                unit = (inputs, targets)
                if grouped is not None:
                    (nmax_in, nmax_out), sizes = grouped
                    up = lambda t: t.to(inputs[5].device, non_blocking=True)
                    nmax_in, nmax_out = up(nmax_in), up(nmax_out)
                    bounds = (torch.clamp(torch.minimum(inputs[5], nmax_in - 2), min=0), nmax_in,
                              torch.clamp(torch.minimum(inputs[9], nmax_out - 2), min=0), nmax_out)
                    unit = GroupedBatch(inputs, targets, bounds, len(sizes), sizes)
                ready = torch.cuda.Event()
                ready.record(copy_stream)
            if ahead is not None:
                yield ahead
            ahead = (unit, frames, ready)
        if ahead is not None:
            yield ahead

    while iteration <= hparams.nb_iterations:
        for unit, frames, ready in device_batches():
            main = torch.cuda.current_stream()
            main.wait_event(ready)
            for t in tuple(unit[0]) + (tuple(unit.bounds) if isinstance(unit, GroupedBatch) else ()):
                t.record_stream(main)        # allocated on the copy stream, consumed on the compute stream
            micro.append((unit, frames))
            if len(micro) < (1 if group else hparams.accumulation_steps):
                continue
            terms, gnorm_sq = trainer.step([u for u, _ in micro], iteration)
            frames = sum(f for _, f in micro)
            micro = []
            lr = trainer.optimizer.param_groups[0]['lr']
            report(time.time())              # the PREVIOUS iteration's scalars (this one is already queued on the device)
            stats_dev = torch.cat((terms, gnorm_sq.sqrt()))
            step_done.record()
            with torch.cuda.stream(log_stream):
                log_stream.wait_event(step_done)
                stats_host.copy_(stats_dev, non_blocking=True)
                stats_ready.record(log_stream)
            stats_dev.record_stream(log_stream)
            pending = (iteration, lr, frames)
            periodic = (val_loader is not None and iteration % hparams.iters_check_for_model_improvement == 0) or \
                iteration % hparams.iters_per_checkpoint == 0 or iteration >= hparams.nb_iterations
            if periodic:
                torch.cuda.synchronize()
                report(time.time())          # validation / checkpoint / end of training: catch up first
            # ---- model evaluation (`train.py:427-456`): every rank scores the whole validation set, rank 0 keeps the best
            if val_loader is not None and iteration % hparams.iters_check_for_model_improvement == 0:
                _logger.info('Validating....')
                val_loss, val_indiv = validate(gpu, model, criterion, val_loader, hparams)
                if rank == 0:
                    _logger.info(f'Validation loss {iteration}: {val_loss:.6f} ')
                    remaining = int((hparams.nb_iterations - iteration) * (total_time / hparams.iters_check_for_model_improvement))
                    _logger.info(f'estimated required time = {remaining // 86400:02}:{remaining // 3600 % 24:02}:'
                                 f'{remaining // 60 % 60:02}:{remaining % 60:02}')
                    total_time = 0.
                    with open(metrics_path, 'a') as f:   # scalar names of DaftExprtLogger.log_validation (logger.py:41-45)
                        rec = {'iteration': iteration, 'DaftExprt.validation/loss': val_loss}
                        rec.update({f'DaftExprt.validation/{k}': v for k, v in val_indiv.items()})
                        f.write(json.dumps(rec) + '\n')
                    if val_loss < best_val_loss:
                        _logger.info('Congrats!!! A new best model. You are the best!')
                        best_val_loss = val_loss
                        save_checkpoint(model, trainer.optimizer, hparams, lr, iteration, best_val_loss,
                                        os.path.join(ckpt_dir, 'DaftExprt_best'))
                if distributed:
                    dist.barrier()
            if iteration % hparams.iters_per_checkpoint == 0:
                if rank == 0:
                    save_checkpoint(model, trainer.optimizer, hparams, lr, iteration, best_val_loss,
                                    os.path.join(ckpt_dir, f'DaftExprt_{iteration}'))
                if distributed:
                    dist.barrier()
            iteration += 1
            if periodic:
                start = time.time()          # validation / checkpoint time is not an iteration's duration (train.py:486)
            if iteration > hparams.nb_iterations:
                break
    if distributed:
        dist.destroy_process_group()


def launch_training(data_set_dir, config_file, benchmark_dir, log_file, world_size=1, rank=0,
                    multiprocessing_distributed=True, master='tcp://localhost:54321'):
    ''' `train.py:497-610`: rebuild HyperParams from the JSON config, one process per GPU '''
    with open(config_file) as f:
        config = json.load(f)
    hparams = HyperParams(verbose=False, **config)
    ngpus = torch.cuda.device_count()
    hparams.data_set_dir, hparams.config_file, hparams.benchmark_dir = data_set_dir, config_file, benchmark_dir
    hparams.rank, hparams.ngpus_per_node, hparams.dist_url = rank, ngpus, master.replace('localhost', '127.0.0.1')
    hparams.multiprocessing_distributed = multiprocessing_distributed
    hparams.world_size = ngpus * world_size if multiprocessing_distributed else 1
    torch.manual_seed(hparams.seed)
    if multiprocessing_distributed:       # one process per GPU, also when there is a single one (`train.py:604-608`)
        torch.multiprocessing.spawn(train, nprocs=ngpus, args=(hparams, log_file))
    else:
        train(0, hparams, log_file)


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--data_set_dir', type=str, required=True)
    parser.add_argument('--config_file', type=str, required=True)
    parser.add_argument('--benchmark_dir', type=str, required=True)
    parser.add_argument('--log_file', type=str, required=True)
    parser.add_argument('--world_size', type=int, default=1)
    parser.add_argument('--rank', type=int, default=0)
    parser.add_argument('--multiprocessing_distributed', action='store_true')
    parser.add_argument('--master', type=str, default='tcp://localhost:54321')
    args = parser.parse_args()
    launch_training(args.data_set_dir, args.config_file, args.benchmark_dir, args.log_file, args.world_size, args.rank,
                    args.multiprocessing_distributed, args.master)
