#!/usr/bin/env python
"""Benchmark of the Daft-Exprt hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one optimizer step of the full model (forward + 7-term loss + backward + gradient all-reduce + Adam,
dropout ON) on one synthetic batch per GPU of BASELINE.json configs[1]: 11 speakers, batch 48 per GPU, 80-bin mel,
T <= 1000 frames (utterance 0 forced to 1000), bf16 MFMA operands / fp32 accumulate and master weights.
Inputs are resident in HBM when the timed region starts.  value = valid mel frames (padding excluded) processed by
ALL ranks per second.  Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel, live HIP-event timing)
and `cpu_baseline` (the CPU oracle timed on this host, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_MFMA_BF16 = 2.5e15     # dense bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md), FLOP/s
SPEAKERS = [f'spk{i:02d}' for i in range(11)]


def f_fwd(T, L):
    ''' algorithmic forward FLOPs of one utterance with T valid frames and L valid phonemes (SURVEY 8d / BASELINE.md 3) '''
    return T * (21222912 + 4096 * T + 256 * L) + L * (7409664 + 2048 * L) + 723712


def make_hparams(batch, dtype):
    from daft_exprt.hparams import HyperParams
    return HyperParams(verbose=False, training_files='none', validation_files='none', output_directory='/nonexistent_out',
                       language='english', speakers=list(SPEAKERS), batch_size=batch, accumulation_steps=1, compute_dtype=dtype)


def cpu_baseline(hp, batch, n_utt=8, steps=1):
    ''' the CPU oracle (a port of the reference algorithm, oracle/daft_exprt_cpu.py) timed on this host: one full
        train step (fwd + loss + autograd bwd + Adam, dropout on, fp32) on the first `n_utt` utterances of the batch '''
    from oracle import daft_exprt_cpu as O
    P = {k: v.requires_grad_(True) for k, v in O.random_params(hp, seed=0).items()}
    sl = slice(0, n_utt)
    cin = [t[sl].clone() for t in batch[:11]]
    L, T = int(cin[5].max()), int(cin[9].max())
    for i in (0, 1, 2, 3, 4):
        cin[i] = cin[i][:, :L]
    cin[6], cin[7], cin[8] = cin[6][:, :T], cin[7][:, :T], cin[8][:, :, :T]
    cin[1], cin[3], cin[4], cin[6], cin[7], cin[8] = [t.float() for t in (cin[1], cin[3], cin[4], cin[6], cin[7], cin[8])]
    targets = (cin[1], cin[3], cin[4], cin[8], cin[10])
    state = {'step': 0, 'm': {k: torch.zeros_like(v) for k, v in P.items()}, 'v': {k: torch.zeros_like(v) for k, v in P.items()}}
    frames = int(cin[9].sum())
    times = []
    for s in range(steps + 1):   # first pass = warm-up
        t0 = time.time()
        out = O.forward(P, hp, tuple(cin), training=True)
        loss, _ = O.loss(hp, out, targets, 20000)
        grads = torch.autograd.grad(loss, list(P.values()))
        with torch.no_grad():
            O.adam_step(P, dict(zip(P.keys(), grads)), state, 1e-4, hp.betas, hp.epsilon, hp.weight_decay)
        times.append(time.time() - t0)
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {'value': frames / best, 'unit': 'mel-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{steps} full train step(s) (fwd+loss+bwd+Adam, fp32, dropout on) of the CPU oracle on the first {n_utt} '
                      f'utterances of the bench batch ({frames} valid frames, T_max={T}); host os.cpu_count()={os.cpu_count()}',
            's_per_step': best}


def synth_bench(args, hp, dev, rank, world):
    ''' BASELINE configs[3]: batched prosody-transfer synthesis, forward only (prosody encoder on the reference mels ->
        phoneme encoder -> predictor -> integer durations -> Gaussian upsampling -> mel decoder), B sentences per call '''
    import numpy as np
    from daft_exprt.model import DaftExprt
    model = DaftExprt(hp).to(dev).eval()
    with torch.no_grad():   # duration head centred on ~80 ms so that random-init weights give utterances of realistic length
        model._P['prosody_predictor.projection.linear_layer.weight'][0].mul_(0.05)
        model._P['prosody_predictor.projection.linear_layer.bias'].copy_(torch.tensor([0.08, 0., 0.]))
        model.mark_updated()
    hp.stats = {f'spk {i}': {'pitch': {'mean': 5.0, 'std': 0.3}} for i in range(hp.n_speakers)}
    rng = np.random.RandomState(1234 + rank)
    B = args.batch
    L = np.sort(rng.randint(40, 161, size=B))[::-1].copy()
    Tr = rng.randint(250, 1001, size=B)
    Lm, Tm = int(L.max()), int(Tr.max())
    symbols = torch.zeros(B, Lm, dtype=torch.long)
    dur_f = torch.ones(B, Lm)
    e_ref, p_ref, m_ref = torch.zeros(B, Tm), torch.zeros(B, Tm), torch.zeros(B, hp.n_mel_channels, Tm)
    for b in range(B):
        symbols[b, :L[b]] = torch.from_numpy(rng.randint(1, hp.n_symbols, size=L[b]))
        dur_f[b, :L[b]] = min(1., 1000. / (L[b] * 0.08 * 86.13 * 1.15))     # keep every utterance <= ~1000 frames
        e_ref[b, :Tr[b]] = torch.from_numpy(rng.uniform(0, 60, size=Tr[b]).astype(np.float32))
        p_ref[b, :Tr[b]] = torch.from_numpy(np.where(rng.rand(Tr[b]) < 0.3, 0., rng.randn(Tr[b]) * 0.3 + 5.).astype(np.float32))
        m_ref[b, :, :Tr[b]] = torch.from_numpy(np.clip(rng.randn(hp.n_mel_channels, Tr[b]) * 2 - 5, np.log(1e-5), 2.).astype(np.float32))
    inputs = (symbols, dur_f, torch.ones(B, Lm), torch.zeros(B, Lm), torch.from_numpy(L), e_ref, p_ref, m_ref,
              torch.from_numpy(Tr), torch.from_numpy(rng.randint(0, 11, size=B)))
    inputs = tuple(t.to(dev) for t in inputs)
    frames = 0
    for w in range(args.warmup):
        out = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        enc, dec, _ = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
        frames += int(dec[1].sum())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({'metric': 'synth-path mels/sec', 'value': B * args.steps / elapsed, 'unit': 'utterances/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                          'config': {'workload': f'BASELINE configs[3]: batched prosody-transfer synthesis, {B} sentences per call, '
                                                 'L~U{40..160}, reference mels T~U{250..1000}, forward only', 'global_batch': B * world,
                                     'generated_frames_per_s': frames / elapsed, 'mean_generated_frames': frames / args.steps / B,
                                     'audio_seconds_per_s (RTF)': frames / elapsed * hp.hop_length / hp.sampling_rate},
                          'roofline': None, 'cpu_baseline': None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=48)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--pool', type=int, default=4, help='distinct synthetic batches cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-probe', action='store_true')
    ap.add_argument('--workload', default='train', choices=['train', 'synth'],
                    help='train = BASELINE configs[1] (default; configs[4] with --batch 256 --tmin 500); synth = configs[3]')
    ap.add_argument('--tmin', type=int, default=1, help='minimum frames per synthetic utterance (configs[4]: 500)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from daft_exprt import ops
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer

    hp = make_hparams(args.batch, args.dtype)
    torch.manual_seed(hp.seed)
    if args.workload == 'synth':
        return synth_bench(args, hp, dev, rank, world)
    model = DaftExprt(hp).to(dev).train()
    trainer = Trainer(model, hp, world)
    batches, cpu_batches = [], []
    for i in range(args.pool):
        cb = synthetic_batch(hp, args.batch, seed=1234 + rank + 1000 * i, t_min=args.tmin, t_max=1000, force_first_full=True)
        cpu_batches.append(cb)
        inputs, targets, _ = model.parse_batch(dev, cb)
        batches.append((inputs, targets))
    frames = [int(b[0][9].sum()) for b in batches]
    flops = [3. * sum(f_fwd(int(t), int(l)) for t, l in zip(b[0][9].tolist(), b[0][5].tolist())) for b in batches]

    def barrier():
        if world > 1:
            dist.barrier()

    it = 20000   # adversarial weight at its maximum (>= warmup_steps): the GRL path is live
    for w in range(args.warmup):
        trainer.step([batches[w % args.pool]], it + w)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    done_frames, done_flops = 0, 0.
    for k in range(args.steps):
        trainer.step([batches[k % args.pool]], it + args.warmup + k)
        done_frames += frames[k % args.pool]
        done_flops += flops[k % args.pool]
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    stats = torch.tensor([elapsed, float(done_frames), done_flops], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, done_frames, done_flops = float(tmax[0]), float(tot[0]), float(tot[1])

    # ---- dominant-kernel roofline: live HIP-event timing of the conv-as-GEMM kernel over extra (untimed) steps
    roofline = None
    if not args.no_probe:
        ops.PROBE = {}
        nprobe = min(args.steps, 4)
        for k in range(nprobe):
            trainer.step([batches[k % args.pool]], it + k)
        torch.cuda.synchronize()
        fam = {}
        for name, recs in ops.PROBE.items():
            ms = sum(s.elapsed_time(e) for s, e, _, _ in recs)
            fam[name] = (ms, recs)
        ops.PROBE = None
        name = max(fam, key=lambda n: fam[n][0])
        ms, recs = fam[name]
        # algorithmic FLOPs: padded-dense FLOPs of each launch scaled by the valid fraction of its time/phoneme axis
        alg = 0.
        for k in range(nprobe):
            inp = batches[k % args.pool][0]
            Tm, Lm = int(inp[8].shape[2]), int(inp[0].shape[1])
            fT, fL = float(inp[9].sum()) / (inp[9].numel() * Tm), float(inp[5].sum()) / (inp[5].numel() * Lm)
            per_step = len(recs) // nprobe
            for s, e, fl, n_axis in recs[k * per_step: (k + 1) * per_step]:
                alg += fl * (fT if n_axis == Tm else fL)
        n_launch = len(recs)
        achieved = alg / (ms * 1e-3) / 1e12
        roofline = {'kernel': f'{name} (conv / linear as implicit GEMM, all call sites)', 'bound': 'mfma',
                    'achieved': achieved, 'peak': PEAK_MFMA_BF16 / 1e12 if args.dtype == 'bf16' else 157.3, 'unit': 'TFLOP/s',
                    'frac': achieved / (PEAK_MFMA_BF16 / 1e12 if args.dtype == 'bf16' else 157.3), 'traffic': None,
                    'launches_per_step': n_launch // nprobe, 'avg_launch_us': ms * 1e3 / n_launch,
                    'algorithmic_gflop_per_launch': alg / n_launch / 1e9,
                    'share_of_step_time': (ms / nprobe) / (elapsed / args.steps * 1e3),
                    'families_ms_per_step': {k: v[0] / nprobe for k, v in fam.items()},
                    'whole_step': {'achieved': done_flops / elapsed / world / 1e12, 'unit': 'TFLOP/s per GPU (algorithmic 3*F_fwd)',
                                   'frac': done_flops / elapsed / world / (PEAK_MFMA_BF16 if args.dtype == 'bf16' else 157.3e12)}}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(hp, cpu_batches[0])
        out = {'metric': 'training mel-frames/sec', 'value': done_frames / elapsed, 'unit': 'mel-frames/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': f'BASELINE configs[{1 if (args.batch == 48 and args.tmin == 1) else 4}]: full Daft-Exprt train step (fwd + 7-term loss + bwd + '
                                      f'Adam, dropout on, adversarial weight at max), 11 speakers, batch {args.batch} per GPU, 80-bin mel, '
                                      f'{args.tmin}<=T<=1000 (utterance 0 = 1000 frames), {args.dtype} MFMA operands, fp32 accumulate/master',
                          'global_batch': args.batch * world, 'batch_per_gpu': args.batch, 'accumulation_steps': 1,
                          'parallelism': f'dp{world}', 'valid_frames_per_step': done_frames / args.steps,
                          'params': model.n_params},
               'roofline': roofline, 'cpu_baseline': cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
