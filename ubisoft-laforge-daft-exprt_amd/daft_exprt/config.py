"""The package's environment switches, read ONCE at import (one place instead of reads scattered through constructors).

Only switches that select between TESTED paths, or that a measurement protocol of DESIGN.md names, are left; the A/B switches of
rounds 1-4 whose verdicts are recorded in DESIGN section 9 ("not kept") are gone together with the code they guarded.  Tests flip
the module attributes they exercise (`ops.USE_SPLITK`, `ops.USE_WIDE`, `ops.USE_FUSED_HEADS`, `ops.WGRAD_WORKSPACE`,
`model.balanced_tiles`, `model.fuse_ln_backward`, `optimizer.fuse_pack`); those have no environment form.

| variable | default | effect |
|---|---|---|
| DX_HIP_LIB | in-tree csrc/libdaftexprt_hip.so | another build of the library (`DX_BUILD_TAG=... python build_hip.py`) for same-box A/B runs |
| DX_WGRAD_SIDE_STREAM | 1 | 0: weight gradients on the launch stream (no second hardware queue) |
| DX_SKIP_WGRAD | 0 | measurement protocol of DESIGN 5 (what the side-stream work costs the step): 1 skips every weight gradient, 2 only the FF ones (128 <-> 1024), 3 only the 1024 x 1024 pre-net one, 4 only the k = 1 ones |
| DX_FORCE_DIST | 0 | 1: a ONE-rank process group issues every collective of the multi-rank path (tests/test_gpu_ddp.py) |
| DX_STREAM_PROBE | 1 | 0: streams straight from torch's pool, no hardware-queue probes (`streams.py`) |
| DX_SECTIONED_ADAM | auto | per-bucket Adam behind each bucket's all-reduce; auto = only with more than one rank |
| DX_STEP_GRAPH | 0 | 1 / auto: `Trainer.step` replayed from a hipGraph (`train.CapturedStep`); DX_STEP_GRAPH_MAX (8) graphs kept |
| DX_RELU_BITS | 1 | 0: the FF data gradient gates on the stored activation `h` instead of the one-bit-per-element mask its forward conv leaves (`dx_conv1d_relu_bits`); same result bit for bit, 2 KB instead of 128 B of gate bytes per row |
| DX_VIRTUAL_RESIDUAL | 1 | 0: every LayerNorm-fused GEMM stores its fp32 output; 1: in a training FFT block the fp32 copy of the attention sub-layer's output is not stored -- its one reader (the residual add of the FF LayerNorm) re-derives it from the saved LayerNorm input (`dx_conv1d_ln_vres`) |
| DX_POISON | 0 | 1: every buffer `ops` allocates is filled with NaN before the kernel that writes it runs (tests: no kernel may read rows it was not given) |
"""
import os


def _flag(name, default):
    return bool(int(os.environ.get(name, default)))


HIP_LIB = os.environ.get('DX_HIP_LIB') or None
WGRAD_SIDE_STREAM = _flag('DX_WGRAD_SIDE_STREAM', '1')
SKIP_WGRAD = int(os.environ.get('DX_SKIP_WGRAD', '0'))   # 1: all; 2 / 3 / 4: only the FF / the 1024 x 1024 pre-net / the k = 1 ones (where does the cost sit?)
STREAM_PROBE = _flag('DX_STREAM_PROBE', '1')
SECTIONED_ADAM = os.environ.get('DX_SECTIONED_ADAM', 'auto')
STEP_GRAPH = os.environ.get('DX_STEP_GRAPH', '0')
STEP_GRAPH_MAX = int(os.environ.get('DX_STEP_GRAPH_MAX', '8'))
POISON = _flag('DX_POISON', '0')
RELU_BITS = _flag('DX_RELU_BITS', '1')
VIRTUAL_RESIDUAL = _flag('DX_VIRTUAL_RESIDUAL', '1')


def force_dist():
    ''' read at call time: tests set it per sub-process / per test '''
    return os.environ.get('DX_FORCE_DIST', '0') == '1'


def sectioned_adam():
    return os.environ.get('DX_SECTIONED_ADAM', SECTIONED_ADAM)


def step_graph():
    return os.environ.get('DX_STEP_GRAPH', STEP_GRAPH)
