"""Build libdaftexprt_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

One object per source under csrc/ (compiled in parallel, rebuilt only when the source or a
header is newer) written to build/obj[.<tag>]/ (git- and gpurun-ignored), linked into
csrc/libdaftexprt_hip.so -- the library is kept IN-TREE so that it travels to the GPU box with
the repo snapshot; the objects do not.

A/B builds: `DX_BUILD_TAG=ldt40 DX_EXTRA_HIPCC_FLAGS=-DDX_FB_LDT=40 python build_hip.py` writes
build/obj.ldt40/*.o and build/libdaftexprt_hip.ldt40.so (that one does travel: build/*.so is not
gpurun-ignored); run with DX_HIP_LIB=<that .so>.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
TAG = os.environ.get('DX_BUILD_TAG', '')
SUFFIX = f'.{TAG}' if TAG else ''
BUILD = os.path.join(HERE, 'build')
OBJDIR = os.path.join(BUILD, 'obj' + SUFFIX)
LIB = os.path.join(BUILD if TAG else CSRC, f'libdaftexprt_hip{SUFFIX}.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wno-unused-result'] + \
    os.environ.get('DX_EXTRA_HIPCC_FLAGS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith('.h')]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _headers_mtime()):
        return obj, False
    cmd = [HIPCC, *FLAGS, '-x', 'hip', '-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    return obj, True


def build(verbose=True):
    srcs = _sources()
    os.makedirs(OBJDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(changed for _, changed in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[build_hip] {len(srcs)} sources, rebuilt={rebuilt}, lib={LIB}')
    return LIB


if __name__ == '__main__':
    build()
