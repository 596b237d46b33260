"""Data-parallel equivalence on hardware: N-rank `forward_backward` + `GradReducer` == single-process gradients (see
tests/ddp_worker.py).  With >= 2 visible GPUs the ranks use RCCL ("nccl") on separate devices; on a 1-GPU box two ranks share
the device and reduce through gloo (device tensors), which still drives the real model, the side-stream section hooks and the
asynchronous-work / wait ordering."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(backend, world, **extra_env):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), **extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'ddp_worker.py'), backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, r.stderr[-3000:]
    assert '[ddp_worker]' in r.stdout


def test_two_ranks_match_single_process_gradients():
    if torch.cuda.device_count() >= 2:
        _run('nccl', 2)
    else:
        _run('gloo', 2)


def test_one_rank_rccl_world_runs_the_multi_rank_path():
    ''' what a 1-GPU box can prove about the N > 1 path on REAL RCCL: a one-rank "nccl" process group with DX_FORCE_DIST=1 builds
        the communicator, broadcasts, issues the asynchronous per-bucket all-reduces from the backward hooks, orders the per-bucket
        Adam behind them with stream waits, and lands on the parameters of a trainer that does none of that (tests/ddp_worker.py) '''
    _run('nccl', 1, DX_FORCE_DIST='1')


def test_process_group_that_shares_the_launch_queue_is_replaced():
    ''' `GradReducer.pick_group`: a group whose collectives do not run beside the launch / weight-gradient streams (probe forced to fail
        once) is replaced by a NEW process group over the same ranks, and the whole equivalence check then runs on that group '''
    _run('nccl', 1, DX_FORCE_DIST='1', DDP_TEST_REGROUP='1')


def test_bench_line_through_a_one_rank_rccl_world():
    ''' bench.py's N > 1 branches (process group, barriers, max-over-ranks clock, summed frames) on a one-rank RCCL world '''
    import json
    env = dict(os.environ, DX_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '3',
                        '--batch', '8', '--no-cpu-baseline', '--no-probe', '--no-secondary'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 0
    assert 'backend nccl' in r.stderr and 'world 1' in r.stderr, r.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs for an RCCL world')
def test_bench_spawns_its_own_ranks():
    ''' `python bench.py --gpus 2` with no launcher must come back with n_gpus == 2 (the RCCL world size) '''
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '8',
                        '--no-cpu-baseline', '--no-probe'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 16


def test_eight_concurrent_one_rank_worlds_on_one_gpu():
    ''' host-side contention rehearsal for the first 8-GPU run: EIGHT bench processes at once on the one GPU of this
        box, each a one-rank RCCL world of its own (DX_FORCE_DIST=1, distinct rendezvous ports) -- 8 x (communicator set-up, stream probes
        under a contended GPU, `new_group` retries, per-bucket hooks) side by side, two of them driving the real `train()` loop with its
        fork-server DataLoader workers.  Every process must finish with a valid line; the probe verdicts go to the captured output. '''
    import json
    ports = []
    socks = [socket.socket() for _ in range(8)]
    for s in socks:
        s.bind(('127.0.0.1', 0))
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    procs = []
    for k, port in enumerate(ports):
        env = dict(os.environ, DX_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '3', '--batch', '8',
               '--no-cpu-baseline', '--no-probe', '--no-secondary']
        if k >= 6:
            cmd += ['--loop', 'train']
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=1500))
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate())
    for k, (p, (so, se)) in enumerate(zip(procs, outs)):
        print(f'--- process {k} rc {p.returncode}\n' + '\n'.join(l for l in se.splitlines() if 'bench' in l or 'streams' in l)[-1500:])
    for k, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (k, se[-3000:])
        line = json.loads([l for l in so.splitlines() if l.startswith('{')][-1])
        assert line['n_gpus'] == 1 and line['value'] > 0, (k, line)
        if k < 6:
            assert 'backend nccl' in se and 'world 1' in se, (k, se[-2000:])
