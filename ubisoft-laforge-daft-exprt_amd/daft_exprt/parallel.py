"""Single-node data parallelism: one process per MI355X, RCCL (torch.distributed backend "nccl") over xGMI.

What the reference does (`src/daft_exprt/train.py:246-251, 293, 391`): wraps the model in
DistributedDataParallel, which broadcasts rank-0 parameters once and all-reduces ~4 gradient buckets
(<= 25 MiB) on EVERY micro-batch backward (no `no_sync()`).

What this build does instead, with the same mathematics (sum of per-rank mean gradients / world):
  * parameters and gradients are two flat fp32 buffers, so a "bucket" is a contiguous slice -- no
    flatten/unflatten copies;
  * the hand-written backward reports each top-level module as soon as its gradients are final
    (frame_decoder -> gaussian_upsampling -> prosody_predictor -> phoneme_encoder -> speaker_classifier ->
    prosody_encoder); the reducer immediately issues an asynchronous all-reduce for that slice, which RCCL runs
    on its own stream while the remaining backward kernels keep the compute stream busy;
  * with gradient accumulation only the LAST micro-batch communicates (one 58.9 MB all-reduce per optimizer
    step instead of one per micro-batch);
  * the 1/world mean is folded into the loss gradient scale, so no extra pass over the gradients.
xGMI is point-to-point (7 links per GPU): a few large messages per step (3.4-30 MB) keep every link busy;
tiny sections are merged with their neighbour so that nothing below ~1 MB goes out on its own.
"""
import torch
import torch.distributed as dist

from daft_exprt import config


class GradReducer(object):
    def __init__(self, model, process_group=None, min_bucket_elems=1 << 18):
        self.model, self.group = model, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # the collectives are issued when there is more than one rank -- or, with DX_FORCE_DIST=1, also inside a ONE-rank process group:
        # a 1-GPU box can then drive the complete multi-rank code path (RCCL communicator, asynchronous all-reduce per bucket,
        # stream-ordered `work.wait()`, per-bucket Adam) on real RCCL (tests/test_gpu_ddp.py)
        self.active = dist.is_initialized() and (self.world > 1 or config.force_dist())
        slices = model.section_slices()
        # buckets in backward order (reverse registration order); merge small sections into the next one
        order = [s for s in reversed(model.SECTIONS) if s in slices]
        self.buckets, pending = [], None
        for sec in order:
            off, n = slices[sec]
            if pending is not None:
                off, n = off, n + pending[1]     # sections are adjacent: pending sits right after this one
                assert off + n == pending[0] + pending[1]
            if n < min_bucket_elems and sec != order[-1]:
                pending = (off, n)
                continue
            self.buckets.append((sec, off, n))
            pending = None
        self._ready_after = {sec: (off, n) for sec, off, n in self.buckets}
        self._works = []

    def pick_group(self, beside, tries=4):
        ''' make sure the collectives of this reducer execute BESIDE the streams in `beside` (launch stream, weight-gradient stream).
            A process group launches on a stream of its own, taken from torch's pool when its communicator is built; if that stream
            shares a hardware queue with the launch stream, every all-reduce would sit in line with the backward kernels instead of
            under them.  The probe is `streams.collective_runs_beside`; a group that fails it is replaced by a NEW group over the same
            ranks (new communicator, next pool stream), at most `tries` times.  All ranks take the same decisions (MIN over ranks).
            RCCL ("nccl") only; collective over all ranks. '''
        from daft_exprt import streams
        self.queue_probe = None
        if not self.active or dist.get_backend(self.group) != 'nccl' or not config.STREAM_PROBE:
            return None
        dev = beside[0].device
        group = self.group
        for attempt in range(tries):
            ok = all([streams.collective_runs_beside(b, group) for b in beside])
            flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()):
                self.group, self.queue_probe = group, f'own hardware queue (group {attempt + 1})'
                return True
            if attempt + 1 < tries:
                ranks = dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)
                rejected, group = group, dist.new_group(ranks=ranks, backend='nccl')    # (collective over the whole world: every rank gets here)
                if rejected is not None and rejected is not self.group and rejected is not dist.group.WORLD:
                    dist.destroy_process_group(rejected)     # a communicator that failed the probe: its streams and buffers go
        self.group, self.queue_probe = group, f'SHARES a hardware queue with the launch or weight-gradient stream after {tries} groups'
        return False

    def broadcast_parameters(self, src=0):
        ''' rank-0 parameters to everyone, once (DDP constructor semantics, train.py:293) '''
        if self.active:
            dist.broadcast(self.model.flat_parameters(), src=src, group=self.group)
            self.model.mark_updated()

    def section_done(self, name):
        ''' backward hook: launch the all-reduce of the bucket closed by this section.  Returns (offset, numel, work) of that bucket
            (work None on one rank) or None when the section only joins a later bucket. '''
        if name not in self._ready_after:
            return None
        off, n = self._ready_after[name]
        if not self.active:
            return off, n, None
        g = self.model.flat_gradients()[off: off + n]
        work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append(work)
        return off, n, work

    def describe(self):
        ''' one line for the rank-0 log: world size, backend and the bucket sizes in backward order '''
        backend = dist.get_backend(self.group) if dist.is_initialized() else 'none'
        sizes = ', '.join(f'{sec} {n * 4 / 1e6:.1f} MB' for sec, _, n in self.buckets)
        probe = getattr(self, 'queue_probe', None)
        return f'gradient all-reduce: world {self.world}, backend {backend}, {len(self.buckets)} buckets in backward order: {sizes}' + \
            (f'; collectives on their {probe}' if probe else '')

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def all_reduce_now(self):
        ''' non-overlapped variant (used when gradients were produced through the autograd bridge) '''
        if self.active:
            dist.all_reduce(self.model.flat_gradients(), op=dist.ReduceOp.SUM, group=self.group)
