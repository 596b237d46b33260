"""HIP streams that really run beside each other.

The runtime maps every HIP stream onto one of a few hardware queues (`GPU_MAX_HW_QUEUES`, 4 per device by default) and kernels of
two streams that share a queue execute in submission order.  torch hands streams out of a pool of 32 per priority, so WHICH queue
`torch.cuda.Stream()` lands on depends on how many pool streams were taken before -- after `init_process_group("nccl")` (which takes
some for RCCL) the weight-gradient stream of this package came out on the launch stream's own queue and the step lost its whole
two-stream overlap (8.9 instead of 7.3 ms at B = 48, measured through a one-rank RCCL world).  So the streams
the step depends on are PROBED: a single-wave spin kernel (`dx_spin`) occupies the queue of one stream while a second, short one is
timed on the candidate; the candidate is taken only if it finished long before the first.

No reference counterpart (torch DDP / autograd leave the queue assignment to chance)."""
import os

import torch

from daft_exprt import _hip as H
from daft_exprt import config

PROBE_US = 1500


def _spin(us, stream):
    H.check(H.lib().dx_spin(int(us), stream.cuda_stream))


def runs_beside(cand, busy, probe_us=PROBE_US):
    ''' True if a kernel launched on `cand` executes while `busy` is occupied.  Synchronises the device (set-up time only). '''
    dev = busy.device
    torch.cuda.synchronize(dev)
    t0, t_busy, t_cand = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    _spin(20, cand)                      # first use of a stream may bind its queue: outside the measurement
    torch.cuda.synchronize(dev)
    t0.record(busy)
    _spin(probe_us, busy)
    t_busy.record(busy)
    _spin(20, cand)
    t_cand.record(cand)
    torch.cuda.synchronize(dev)
    return t0.elapsed_time(t_cand) < 0.5 * t0.elapsed_time(t_busy)


def pick(beside, device=None, priority=None, tries=12, what='stream'):
    ''' a new torch stream that runs beside every stream in `beside` (a list of torch streams).  Falls back to the last candidate
        (with a warning on stderr) when none of `tries` pool streams qualifies -- e.g. GPU_MAX_HW_QUEUES=1. '''
    device = beside[0].device if device is None else device
    capturing = getattr(torch.cuda, 'is_current_stream_capturing', lambda: False)()     # (a probe synchronises: never inside a capture)
    if not config.STREAM_PROBE or capturing:
        return torch.cuda.Stream(device=device) if priority is None else torch.cuda.Stream(device=device, priority=priority)
    cand = None
    for k in range(tries):
        cand = torch.cuda.Stream(device=device) if priority is None else torch.cuda.Stream(device=device, priority=priority)
        if all(runs_beside(cand, b) for b in beside):
            return cand
    import sys
    print(f'[daft_exprt.streams] no {what} on its own hardware queue among {tries} candidates (GPU_MAX_HW_QUEUES='
          f'{os.environ.get("GPU_MAX_HW_QUEUES", "default")}): it will run in submission order with the launch stream', file=sys.stderr)
    return cand


def collective_runs_beside(busy, group=None, probe_us=PROBE_US):
    ''' True if a collective of torch.distributed's `group` executes while `busy` is occupied (the process group launches on a
        stream of its own that this package cannot choose).  Every rank of the group must call it. '''
    import torch.distributed as dist
    dev = busy.device
    x = torch.zeros(256, device=dev)
    helper = pick([busy], what='probe stream')
    with torch.cuda.stream(helper):
        dist.all_reduce(x, group=group)            # communicator set-up and first launch outside the measurement
        if dist.get_world_size(group) > 1:
            dist.all_reduce(x, group=group)        # a second one brings the ranks within microseconds of each other: a rank that enters the
                                                   # timed collective late would make every other rank's probe look queued behind the spin
    torch.cuda.synchronize(dev)
    t0, t_busy, t_coll = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record(busy)
    _spin(probe_us, busy)
    t_busy.record(busy)
    with torch.cuda.stream(helper):
        dist.all_reduce(x, group=group)            # the group's stream waits for `helper` (idle), not for `busy`
        t_coll.record(helper)
    torch.cuda.synchronize(dev)
    return t0.elapsed_time(t_coll) < 0.5 * t0.elapsed_time(t_busy)
