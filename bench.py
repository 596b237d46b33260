#!/usr/bin/env python
"""Benchmark of the Daft-Exprt hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or bench.py does it itself: with no
WORLD_SIZE in the environment and --gpus N > 1 it re-executes itself under torch.distributed.run (127.0.0.1 rendezvous).
`n_gpus` in the JSON line is always the size of the RCCL world that ran; --gpus that disagrees with WORLD_SIZE is an error.

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


def f_synth(T_ref, T_gen, L):
    ''' the same model split for the synthesis path (forward only): prosody encoder on the T_ref reference frames (pre-net
        7 570 944 + 4 FFT blocks of 1 703 936 + 512 T_ref per frame), mel decoder on the T_gen generated frames (4 blocks + mel
        projection 20 480 + Gaussian upsampling 256 L per frame), phoneme side and per-utterance heads as in f_fwd;
        f_synth(T, T, L) == f_fwd(T, L) '''
    return T_ref * (14386688 + 2048 * T_ref) + T_gen * (6836224 + 2048 * T_gen + 256 * L) + L * (7409664 + 2048 * L) + 723712


def csrc_sha16():
    ''' fingerprint of the kernel sources: recorded PMC figures are only quoted for the build they were measured on '''
    import hashlib
    d = os.path.join(ROOT, 'ubisoft-laforge-daft-exprt_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h', '.cpp')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def recorded_counters(tag):
    ''' newest committed PMC summary profiles/rNN_<tag>.json recorded for THIS kernel build (tools/pmc_counters.py stores the
        csrc fingerprint), or (None, reason) '''
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r[0-9][0-9]_{tag}.json')), reverse=True)
    for path in paths:
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get('csrc_sha16') == csrc_sha16():
            return rec, os.path.relpath(path, ROOT)
        return None, f'{os.path.relpath(path, ROOT)} was recorded for kernel build {rec.get("csrc_sha16")}, this build is {csrc_sha16()}: not quoted'
    return None, 'no PMC summary committed'


def make_hparams(batch, dtype):
    from daft_exprt.hparams import HyperParams
    return HyperParams(verbose=False, training_files='none', validation_files='none', output_directory='/nonexistent_out',
                       language='english', speakers=list(SPEAKERS), batch_size=batch, accumulation_steps=1, compute_dtype=dtype)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_train_steps(hp, batch, n_utt, warmup, steps):
    ''' `warmup` + `steps` full train steps (fwd + 7-term loss + autograd bwd + Adam, dropout on, fp32) of the CPU oracle on the first
        `n_utt` utterances of `batch` (a collate 13-tuple); returns (sorted timed seconds, valid frames, T_max) '''
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
    times = []
    for s in range(warmup + steps):
        t0 = time.time()
        out = O.forward(P, hp, tuple(cin), training=True)
        loss, _ = O.loss(hp, out, targets, 20000)
        grads = torch.autograd.grad(loss, list(P.values()))
        with torch.no_grad():
            O.adam_step(P, dict(zip(P.keys(), grads)), state, 1e-4, hp.betas, hp.epsilon, hp.weight_decay)
        times.append(time.time() - t0)
    return sorted(times[warmup:]), int(cin[9].sum()), T


def cpu_baseline(hp, batch, n_utt=8, warmup=1, steps=3):
    ''' the CPU oracle (a port of the reference algorithm, oracle/daft_exprt_cpu.py) timed on this host, BASELINE.md 4 procedure on a
        BOUNDED sample (the whole B = 48 batch costs ~100 s per step):
          1. thread sweep: torch.set_num_threads in {8, 16, 32, 64, 128} (those the host has), 1 warm-up + 1 timed train step each on the
             first 4 utterances -- the oracle is hundreds of small ATen ops per step, and torch's default of one thread per core
             (128 here) oversubscribes them;
          2. the reported value: `warmup` + `steps` train steps on the first `n_utt` utterances of the bench batch at the best count;
          3. `c1`: BASELINE configs[0] (single speaker, B = 8, T <= 800), same procedure, 1 + 2 steps.
        Runs after the GPU timing, outside every timed region. '''
    default_threads = torch.get_num_threads()
    cores = os.cpu_count() or default_threads
    sweep = {}
    for n in (8, 16, 32, 64, 128):
        if n > cores:
            break
        torch.set_num_threads(n)
        t, fr, _ = _cpu_train_steps(hp, batch, 4, 1, 1)
        sweep[n] = fr / t[0]
    best = max(sweep, key=sweep.get) if sweep else default_threads
    torch.set_num_threads(best)
    timed, frames, T = _cpu_train_steps(hp, batch, n_utt, warmup, steps)
    med = timed[len(timed) // 2]
    # configs[0]: the reference's own CPU-runnable case
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.hparams import HyperParams
    hp1 = HyperParams(verbose=False, training_files='none', validation_files='none', output_directory='/nonexistent_out',
                      language='english', speakers=['LJ'], batch_size=8, accumulation_steps=1, compute_dtype='fp32')
    cb1 = synthetic_batch(hp1, 8, seed=99, t_min=1, t_max=800, force_first_full=True)
    t1, f1, T1 = _cpu_train_steps(hp1, cb1, 8, 1, 2)
    m1 = t1[len(t1) // 2]
    torch.set_num_threads(default_threads)
    return {'value': frames / med, 'unit': 'mel-frames/s', 'cores': best, 'kind': 'port',
            'sample': f'{warmup} warm-up + {steps} timed full train steps (fwd+loss+bwd+Adam, fp32, dropout on) of the CPU oracle '
                      f'on the first {n_utt} utterances of the bench batch ({frames} valid frames, T_max={T}), median, at the best of '
                      f'the swept torch thread counts ({best}); host os.cpu_count()={cores}, CPU "{_cpu_model()}"',
            's_per_step_median': med, 's_per_step_min': timed[0], 's_per_step_max': timed[-1], 'cpu_model': _cpu_model(),
            'thread_sweep_frames_per_s': {str(k): v for k, v in sweep.items()},
            'thread_sweep_sample': '1 warm-up + 1 timed train step on the first 4 utterances per thread count',
            'c1': {'value': f1 / m1, 'unit': 'mel-frames/s', 'cores': best, 's_per_step_median': m1,
                   'sample': f'BASELINE configs[0]: single speaker, B = 8, T <= 800 ({f1} valid frames, T_max={T1}), 1 warm-up + 2 timed '
                             f'train steps of the CPU oracle, median'}}


def measured_traffic(kernel_family, tag='counters'):
    ''' HBM-side bytes per launch of the roofline kernel family from the committed PMC summary (rocprofv3 --pmc
        FETCH_SIZE / --pmc WRITE_SIZE in separate passes, gfx950 x2 fetch correction applied; written by
        tools/pmc_counters.py from the same bench command).  PMC counters cannot be read from inside this process, so
        this is a RECORDED figure: it is quoted only when the summary carries the fingerprint of the kernel sources that
        are running now (otherwise null + the reason). '''
    rec, src = recorded_counters(tag)
    if rec is None:
        return {'bytes_per_launch': None, 'note': src}
    fam = rec.get('families', {}).get(kernel_family)
    if not fam or 'fetch_x2_bytes' not in fam:
        return {'bytes_per_launch': None, 'note': f'{src}: no entry for {kernel_family}'}
    return {'bytes_per_launch': fam['fetch_x2_bytes'] + fam['write_bytes'], 'fetch_x2_bytes': fam['fetch_x2_bytes'],
            'write_bytes': fam['write_bytes'], 'launches': fam['launches'], 'mfma_util': fam.get('mfma_util'),
            'source': f'{src} (rocprofv3 --pmc, recorded run of this bench command on this kernel build {rec["csrc_sha16"]})'}


def hbm_line(ms_per_step, tag='counters'):
    ''' whole-step HBM traffic (sum over every kernel of FETCH_SIZE x 2 + WRITE_SIZE per step, recorded PMC passes) over the
        LIVE step time, against the 8 TB/s HBM3E peak of MI355X_MICROARCH.md '''
    rec, src = recorded_counters(tag)
    if rec is None or 'whole_step' not in rec:
        return {'gbps': None, 'note': src}
    b = rec['whole_step']['fetch_x2_bytes'] + rec['whole_step']['write_bytes']
    return {'gbps': b / (ms_per_step * 1e-3) / 1e9, 'peak_gbps': 8000., 'frac': b / (ms_per_step * 1e-3) / 8e12, 'bytes_per_step': b,
            'fetch_x2_bytes_per_step': rec['whole_step']['fetch_x2_bytes'], 'write_bytes_per_step': rec['whole_step']['write_bytes'],
            'source': f'{src}: bytes per step from the recorded --pmc passes (kernel build {rec["csrc_sha16"]}), time from this run'}


def family_tables(fam, nprobe, valid_frac, peak):
    ''' per-family {ms, launches, achieved TFLOP/s, fraction of peak}; algorithmic FLOPs = padded-dense FLOPs of each launch
        scaled by the valid fraction of its time / phoneme axis (valid_frac(k, n_axis)) '''
    out = {}
    for fname, (fms, frecs) in fam.items():
        falg, per_step = 0., len(frecs) // nprobe
        for k in range(nprobe):
            for s_, e_, fl, n_axis in frecs[k * per_step: (k + 1) * per_step]:
                falg += fl * valid_frac(k, n_axis)
        out[fname] = {'ms_per_step': fms / nprobe, 'launches_per_step': per_step, 'achieved_tflops': falg / (fms * 1e-3) / 1e12,
                      'frac_of_peak': falg / (fms * 1e-3) / peak, 'algorithmic_gflop_per_step': falg / nprobe / 1e9,
                      'stream': 'side (overlapped with the main stream)' if fname == 'conv_wgrad' else 'main'}
    return out


def cpu_baseline_synth(hp, cpu_inputs, n_utt=8, warmup=2, steps=5):
    ''' the CPU oracle's `inference` (oracle/daft_exprt_cpu.py, a port of model.py:866-923) timed on this host on every
        (B / n_utt)-th sentence of the bench batch (bounded sample), `warmup` + `steps` calls, median '''
    from oracle import daft_exprt_cpu as O
    P = O.random_params(hp, seed=0)
    with torch.no_grad():   # same centring of the duration head as the GPU run (random-init weights predict arbitrary durations)
        P['prosody_predictor.projection.linear_layer.weight'][0].mul_(0.05)
        P['prosody_predictor.projection.linear_layer.bias'].copy_(torch.tensor([0.08, 0., 0.]))
    B = cpu_inputs[0].shape[0]
    idx = torch.arange(0, B, max(1, B // n_utt))[:n_utt]
    cin = [t[idx].clone() for t in cpu_inputs]
    L, T = int(cin[4].max()), int(cin[8].max())
    for i in (0, 1, 2, 3):
        cin[i] = cin[i][:, :L]
    cin[5], cin[6], cin[7] = cin[5][:, :T], cin[6][:, :T], cin[7][:, :, :T]
    times, frames = [], 0
    for s in range(warmup + steps):
        t0 = time.time()
        enc, dec, _ = O.inference(P, hp, tuple(t.clone() for t in cin), 'add')
        times.append(time.time() - t0)
        frames = int(dec[1].sum())
    timed = sorted(times[warmup:])
    med = timed[len(timed) // 2]
    return {'value': len(idx) / med, 'unit': 'utterances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{warmup} warm-up + {steps} timed calls of the CPU oracle inference (fp32) on {len(idx)} sentences of the bench batch '
                      f'(every {max(1, B // n_utt)}-th; L_max={L}, T_ref_max={T}, {frames} generated frames), median; '
                      f'host os.cpu_count()={os.cpu_count()}, torch threads={torch.get_num_threads()}, CPU "{_cpu_model()}"',
            'generated_frames_per_s': frames / med, 's_per_call_median': med, 's_per_call_min': timed[0], 's_per_call_max': timed[-1],
            'cpu_model': _cpu_model()}


def synth_bench(args, hp, dev, rank, world, emit=True, cpu_steps=(2, 5)):
    ''' BASELINE configs[3]: batched prosody-transfer synthesis, forward only (prosody encoder on the reference mels ->
        phoneme encoder -> predictor -> integer durations -> Gaussian upsampling -> mel decoder), B sentences per call.
        Accounting follows the reference's own (generate.py:413-435, scripts/synthesize.py:117-135): sentences and generated
        frames per second of wall time of the batched call, RTF = generated audio seconds per second. '''
    from daft_exprt import ops
    from daft_exprt.data_loader import centre_duration_head, synthetic_inference_batch
    from daft_exprt.model import DaftExprt
    model = DaftExprt(hp).to(dev).eval()
    centre_duration_head(model)   # duration head centred on ~80 ms so that random-init weights give utterances of realistic length
    hp.stats = {f'spk {i}': {'pitch': {'mean': 5.0, 'std': 0.3}} for i in range(hp.n_speakers)}
    B = args.batch
    cpu_inputs = synthetic_inference_batch(hp, B, seed=1234 + rank)
    inputs = tuple(t.to(dev) for t in cpu_inputs)
    frames, flops = 0, 0.
    for w in range(args.warmup):
        out = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    lens = []
    for k in range(args.steps):
        enc, dec, _ = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
        lens.append(dec[1])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L_list, Tr_list = cpu_inputs[4].tolist(), cpu_inputs[8].tolist()
    for ol in lens:
        ol = ol.tolist()
        frames += sum(ol)
        flops += sum(f_synth(int(tr), int(tg), int(l)) for tr, tg, l in zip(Tr_list, ol, L_list))
    stats = torch.tensor([elapsed, float(B * args.steps), float(frames), flops], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = stats[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        stats = torch.cat([tmax, tot])
    elapsed, utts, frames, flops = [float(v) for v in stats]
    peak = PEAK_MFMA_BF16 if args.dtype == 'bf16' else 157.3e12
    roofline = None
    if not args.no_probe:
        ops.PROBE = {}
        nprobe = min(args.steps, 3)
        outl = []
        for k in range(nprobe):
            enc, dec, _ = model.inference(tuple(t.clone() for t in inputs), 'add', hp)
            outl.append(dec[1])
        torch.cuda.synchronize()
        fam = {name: (sum(s_.elapsed_time(e_) for s_, e_, _, _ in recs), recs) for name, recs in ops.PROBE.items()}
        ops.PROBE = None
        Lm, Trm = int(inputs[0].shape[1]), int(inputs[7].shape[2])
        fL, fTr = float(inputs[4].sum()) / (B * Lm), float(inputs[8].sum()) / (B * Trm)

        def valid_frac(k, n_axis):   # launches are identified by the length of their position axis
            if n_axis == Trm:
                return fTr
            if n_axis == Lm:
                return fL
            return float(outl[k].sum()) / (B * n_axis)   # decoder: generated frames of this call
        families = family_tables(fam, nprobe, valid_frac, peak)
        name = max(families, key=lambda n: families[n]['ms_per_step'])
        f = families[name]
        roofline = {'kernel': f'{name} (conv / linear as implicit GEMM, all call sites of the synthesis path)', 'bound': 'mfma',
                    'achieved': f['achieved_tflops'], 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': f['frac_of_peak'],
                    'traffic': measured_traffic(name, 'synth_counters') if (B == 256 and args.dtype == 'bf16') else None,
                    'launches_per_step': f['launches_per_step'], 'avg_launch_us': f['ms_per_step'] * 1e3 / max(1, f['launches_per_step']),
                    'algorithmic_gflop_per_launch': f['algorithmic_gflop_per_step'] / max(1, f['launches_per_step']),
                    'share_of_step_time': f['ms_per_step'] / (elapsed / args.steps * 1e3), 'families': families,
                    'whole_step': {'achieved': flops / elapsed / world / 1e12, 'unit': 'TFLOP/s per GPU (algorithmic, sum of f_synth over the generated utterances)',
                                   'frac': flops / elapsed / world / peak},
                    'hbm': hbm_line(elapsed / args.steps * 1e3, 'synth_counters') if (B == 256 and args.dtype == 'bf16') else None}
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_synth(hp, cpu_inputs, warmup=cpu_steps[0], steps=cpu_steps[1])
        line = ({'metric': 'synth-path mels/sec', 'value': utts / elapsed, 'unit': 'utterances/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                          'config': {'workload': f'BASELINE configs[3]: batched prosody-transfer synthesis, {B} sentences per call, '
                                                 'L~U{40..160}, reference mels T~U{250..1000}, forward only', 'global_batch': B * world,
                                     'generated_frames_per_s': frames / elapsed, 'mean_generated_frames': frames / utts,
                                     'audio_seconds_per_s (RTF^-1 of generate.py:422-435)': frames / elapsed * hp.hop_length / hp.sampling_rate},
                          'roofline': roofline, 'cpu_baseline': cpu})
        if not emit:
            return line
        print(json.dumps(line))
    if world > 1 and emit:
        dist.destroy_process_group()
    return None


def train_loop_bench(args, hp):
    ''' the loop users run (`daft_exprt.train.train`, the reference's train.py:236-494): DataLoader workers synthesising and
        collating utterances, `parse_batch` H2D copies, `Trainer.step`, a log line + metrics record per iteration.  Timed from its
        own per-iteration durations (metrics.jsonl), the first `--warmup` iterations dropped. '''
    import tempfile
    from daft_exprt import train as T
    out = tempfile.mkdtemp(prefix='dx_train_loop_')
    hp.output_directory = out
    hp.nb_iterations = args.warmup + args.steps
    hp.iters_per_checkpoint = 10 ** 9
    hp.iters_check_for_model_improvement = 10 ** 9
    hp.checkpoint = ''
    hp.multiprocessing_distributed = False
    hp.synthetic_items = args.batch * (args.warmup + args.steps + 4)
    hp.synthetic_workers = 24
    T.train(0, hp, os.path.join(out, 'train.log'))
    recs = [json.loads(l) for l in open(os.path.join(out, 'metrics.jsonl')) if 'DaftExprt.training/loss' in l]
    recs = recs[args.warmup:]
    secs = sum(r['DaftExprt.optimization/duration'] for r in recs)
    frames = sum(r['valid_frames'] for r in recs)
    print(json.dumps({'metric': 'training mel-frames/sec (real train() loop)', 'value': frames / secs, 'unit': 'mel-frames/s', 'n_gpus': 1,
                      'steps': len(recs), 'warmup': args.warmup, 'ms_per_step': secs / len(recs) * 1e3, 'higher_is_better': True,
                      'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
                      'config': {'workload': f'daft_exprt.train.train(): synthetic utterances through the DataLoader (24 workers) + collate + H2D + '
                                             f'Trainer.step + per-iteration log line, batch {args.batch}, T <= 1000, {args.dtype}',
                                 'valid_frames_per_step': frames / len(recs)},
                      'roofline': None, 'cpu_baseline': None}))


def secondary_train(dev, batch, accum, dtype, steps, warmup, pool=4, group=True, t_min=1, probe=0):
    ''' a second timing of the SAME train step under another schedule / arithmetic, reported as an extra object of the default line:
        `batch` utterances per micro-batch x `accum` micro-batches per optimizer step (the reference's own default is 16 x 3,
        hparams.py:66-67, README.md:180), operands `dtype` (fp32 = the exact-parity mode on v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s).
        Same synthetic utterance statistics as the headline (T <= 1000, utterance 0 of every micro-batch pool entry = 1000 frames).
        group (hparams.group_micro_batches, the default): the micro-batches of a step run as ONE pass over their concatenation, each
        utterance keeping its own micro-batch's padded length as a hard sequence end (data_loader.GroupedBatch: the same gradients
        as the separate passes, tests/test_gpu_grouped.py); the concatenation is input staging (train() does it on the host before
        the H2D copy) and is done once per resident batch, outside the timed region like the H2D copy itself. '''
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.hparams import HyperParams
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer
    hp = HyperParams(verbose=False, training_files='none', validation_files='none', output_directory='/nonexistent_out',
                     language='english', speakers=list(SPEAKERS), batch_size=batch, accumulation_steps=accum, compute_dtype=dtype,
                     group_micro_batches=group)
    torch.manual_seed(hp.seed)
    model = DaftExprt(hp).to(dev).train()
    trainer = Trainer(model, hp, 1)
    groups, host0 = [], []
    for i in range(pool):
        micro = []
        for a in range(accum):
            cb = synthetic_batch(hp, batch, seed=4321 + 1000 * i + 37 * a, t_min=t_min, t_max=1000, force_first_full=(a == 0))
            if i == 0:
                host0.append(cb)
            inputs, targets, _ = model.parse_batch(dev, cb)
            micro.append((inputs, targets))
        groups.append(micro)
    host_group_ms = None
    if group and accum > 1:      # what train() pays on the HOST per optimizer step for the merge (pad + concatenate the collate outputs in the
        from daft_exprt.data_loader import group_host_batches     # loader-consumer path, in front of the H2D copy): not part of the timing below
        ts = []
        for _ in range(8):       # (the first calls grow the host allocator's pools)
            t0 = time.perf_counter()
            group_host_batches(host0)
            ts.append(time.perf_counter() - t0)
        host_group_ms = sorted(ts[3:])[2] * 1e3
    frames = [sum(int(m[0][9].sum()) for m in g) for g in groups]
    flops = [3. * sum(f_fwd(int(t), int(l)) for m in g for t, l in zip(m[0][9].tolist(), m[0][5].tolist())) for g in groups]
    it = 20000
    trainer.captured = None      # eager launches: a replayed hipGraph ties them at these sizes (`--graph` on the headline run)
    for w in range(warmup):
        trainer.step(groups[w % pool], it + w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done_frames, done_flops = 0, 0.
    for k in range(steps):
        trainer.step(groups[k % pool], it + warmup + k)
        done_frames += frames[k % pool]
        done_flops += flops[k % pool]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    peak = PEAK_MFMA_BF16 if dtype == 'bf16' else 157.3e12
    families = None
    if probe and accum == 1:      # per-family HIP-event timing over `probe` extra (untimed) steps, as the headline's roofline object
        from daft_exprt import ops
        ops.PROBE = {}
        for k in range(probe):
            trainer.step(groups[k % pool], it + k)
        torch.cuda.synchronize()
        fam = {name: (sum(s_.elapsed_time(e_) for s_, e_, _, _ in recs), recs) for name, recs in ops.PROBE.items()}
        ops.PROBE = None

        def valid_frac(k, n_axis):
            inp = groups[k % pool][0][0]
            Tm, Lm = int(inp[8].shape[2]), int(inp[0].shape[1])
            return float(inp[9].sum()) / (inp[9].numel() * Tm) if n_axis == Tm else float(inp[5].sum()) / (inp[5].numel() * Lm)
        families = family_tables(fam, probe, valid_frac, peak)
    del trainer, model, groups
    torch.cuda.empty_cache()
    extra = {'families': families} if families is not None else {}
    if host_group_ms is not None:
        extra['host_grouping_ms_per_step'] = host_group_ms    # (the grouped figure excludes it, like the H2D copy; train() overlaps it with the previous step)
    return {**extra, 'value': done_frames / elapsed, 'unit': 'mel-frames/s', 'ms_per_step': elapsed / steps * 1e3, 'steps': steps, 'warmup': warmup,
            'dtype': dtype, 'batch_size': batch, 'accumulation_steps': accum, 'utterances_per_optimizer_step': batch * accum,
            'valid_frames_per_step': done_frames / steps, 'grouped_micro_batches': bool(group and accum > 1),
            'whole_step': {'achieved': done_flops / elapsed / 1e12, 'unit': 'TFLOP/s (algorithmic 3*F_fwd)', 'peak': peak / 1e12,
                           'frac': done_flops / elapsed / peak}}


def spawn_ranks(n):
    ''' re-execute this command line as n ranks (one per GPU) under torch.distributed.run; rank 0 prints the JSON line '''
    import socket
    import subprocess
    if n > torch.cuda.device_count():
        raise SystemExit(f'bench.py: --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument('--cpu-utts', type=int, default=8, help='utterances of the bench batch the cpu_baseline leg steps through (48 = the whole batch, '
                    '~20 s per step: BASELINE.md 4; the default keeps the default run within a few minutes)')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed steps of the cpu_baseline leg (after 1 warm-up)')
    ap.add_argument('--graph', action='store_true', help='replay each resident batch\'s step as ONE captured hipGraph (train.CapturedStep) instead of '
                    'launching its ~300 kernels from the host (measured: a tie at B = 48, see DESIGN 5)')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the extra objects of the default line ("synth" = configs[3] at 256 sentences, "train_16x3" = the reference\'s own '
                         '16 x 3 accumulation schedule, "fp32" = the exact-parity arithmetic); they only run for the default 1-GPU train workload')
    ap.add_argument('--workload', default='train', choices=['train', 'synth'],
                    help='train = BASELINE configs[1] (default; configs[4] with --batch 256 --tmin 500); synth = configs[3]')
    ap.add_argument('--tmin', type=int, default=1, help='minimum frames per synthetic utterance (configs[4]: 500)')
    ap.add_argument('--loop', default='step', choices=['step', 'train'],
                    help='step = Trainer.step back to back on resident batches (the metric); train = the REAL daft_exprt.train.train() loop '
                         '(DataLoader workers, collate, H2D, per-iteration logging) on synthetic utterances, 1 GPU')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks or pass --gpus {world}')
    if world > torch.cuda.device_count():
        raise SystemExit(f'bench.py: {world} ranks requested but only {torch.cuda.device_count()} GPU(s) visible')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # DX_FORCE_DIST=1: also a single rank builds its (one-rank) RCCL world and runs every collective of the N > 1 path
    dist_on = world > 1 or os.environ.get('DX_FORCE_DIST', '0') == '1'
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('NCCL_DEBUG', 'VERSION')
        dist.init_process_group(backend='nccl', rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == world

    from daft_exprt import ops
    from daft_exprt.data_loader import synthetic_batch
    from daft_exprt.model import DaftExprt
    from daft_exprt.train import Trainer

    hp = make_hparams(args.batch, args.dtype)
    torch.manual_seed(hp.seed)
    if args.workload == 'synth':
        return synth_bench(args, hp, dev, rank, world)
    if args.loop == 'train':
        return train_loop_bench(args, hp)
    model = DaftExprt(hp).to(dev).train()
    model.set_rank(rank)
    trainer = Trainer(model, hp, world)
    if rank == 0 or dist_on:      # stderr: the JSON line stays alone on stdout.  EVERY rank of a multi-rank run reports its bucket table,
        # the verdict of the hardware-queue probes and the RCCL build, so that a failed SCALE run can be diagnosed from the record's tail
        try:
            rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:   # noqa: BLE001
            rccl = f'unknown ({type(e).__name__})'
        side = getattr(model, '_side', None)
        print(f'[bench rank {rank}/{world} pid {os.getpid()} {torch.cuda.get_device_name(dev)} cuda:{local_rank}] world '
              f'{dist.get_world_size() if dist_on else 1}, RCCL {rccl}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}, '
              f'GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES", "default")}; {trainer.reducer.describe()}; weight-gradient stream '
              f'{"probed" if side is not None else "not created yet"}; per-bucket Adam {"on" if trainer.sectioned else "off"}',
              file=sys.stderr, flush=True)
    batches, cpu_batches = [], []
    for i in range(args.pool):
        cb = synthetic_batch(hp, args.batch, seed=1234 + rank + 1000 * i, t_min=args.tmin, t_max=1000, force_first_full=True)
        cpu_batches.append(cb)
        inputs, targets, _ = model.parse_batch(dev, cb)
        batches.append((inputs, targets))
    frames = [int(b[0][9].sum()) for b in batches]
    flops = [3. * sum(f_fwd(int(t), int(l)) for t, l in zip(b[0][9].tolist(), b[0][5].tolist())) for b in batches]

    def barrier():
        if dist_on:
            dist.barrier()

    it = 20000   # adversarial weight at its maximum (>= warmup_steps): the GRL path is live
    if args.graph and trainer.captured is not None:          # set-up, like building the model: the hipGraph of each resident batch's step
        trainer.captured.auto = False                          # (one eager step to load the kernels, then the capture, which executes nothing)
        for b in batches:
            trainer.captured.prepare([b], it)
    elif trainer.captured is not None:
        trainer.captured = None
    for w in range(args.warmup):
        trainer.step([batches[w % args.pool]], it + w)
    torch.cuda.synchronize()
    if dist_on:
        trainer.diag = []          # per step: (event behind the last backward kernel, event behind the last all-reduce), see the stderr line below
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
    if dist_on:
        # per-rank diagnostics of the FIRST multi-GPU record: the imbalance term of the scaling prediction (this rank's padded
        # sizes and valid frames: a rank whose T_max is larger runs a longer step and everyone waits for it in the collective) and the
        # communication the backward pass did not hide (event behind the last backward kernel -> event behind the last all-reduce)
        exposed = []
        for e0, e1 in (trainer.diag or []):
            try:
                exposed.append(e0.elapsed_time(e1))
            except Exception:   # noqa: BLE001
                pass
        trainer.diag = None
        exposed.sort()
        shapes = ', '.join(f'T_max {int(b[0][8].shape[2])} L_max {int(b[0][0].shape[1])} frames {f}' for b, f in zip(batches, frames))
        print(f'[bench rank {rank}/{world}] local step {elapsed / args.steps * 1e3:.3f} ms; batches: {shapes}; last backward kernel -> end of the '
              f'last all-reduce: median {exposed[len(exposed) // 2] if exposed else float("nan"):.3f} ms, max {exposed[-1] if exposed else float("nan"):.3f} ms '
              f'over {len(exposed)} steps (negative = the collectives ended under the backward pass)', file=sys.stderr, flush=True)
    stats = torch.tensor([elapsed, float(done_frames), done_flops], dtype=torch.float64, device=dev)
    if dist_on:
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
        # the dominant kernel family of the MAIN stream (the critical path of the step); the weight-gradient family runs on the side
        # stream underneath it and is reported next to it in `families`
        main_fams = {n: v for n, v in fam.items() if n != 'conv_wgrad'} or fam
        name = max(main_fams, key=lambda n: main_fams[n][0])
        ms, recs = fam[name]
        peak = PEAK_MFMA_BF16 if args.dtype == 'bf16' else 157.3e12

        def valid_frac(k, n_axis):   # padded-dense FLOPs of a launch x the valid fraction of its time / phoneme axis
            inp = batches[k % args.pool][0]
            Tm, Lm = int(inp[8].shape[2]), int(inp[0].shape[1])
            return float(inp[9].sum()) / (inp[9].numel() * Tm) if n_axis == Tm else float(inp[5].sum()) / (inp[5].numel() * Lm)
        families = family_tables(fam, nprobe, valid_frac, peak)
        f = families[name]
        n_launch = len(recs)
        c2 = args.batch == 48 and args.tmin == 1 and args.dtype == 'bf16'
        roofline = {'kernel': f'{name} (conv / linear as implicit GEMM on the main stream, all call sites)', 'bound': 'mfma',
                    'achieved': f['achieved_tflops'], 'peak': peak / 1e12, 'unit': 'TFLOP/s', 'frac': f['frac_of_peak'],
                    'traffic': measured_traffic(name) if c2 else None,
                    'launches_per_step': n_launch // nprobe, 'avg_launch_us': ms * 1e3 / n_launch,
                    'algorithmic_gflop_per_launch': f['algorithmic_gflop_per_step'] * nprobe / n_launch,
                    'share_of_step_time': (ms / nprobe) / (elapsed / args.steps * 1e3),
                    'families_ms_per_step': {k: v[0] / nprobe for k, v in fam.items()}, 'families': families,
                    'whole_step': {'achieved': done_flops / elapsed / world / 1e12, 'unit': 'TFLOP/s per GPU (algorithmic 3*F_fwd)',
                                   'frac': done_flops / elapsed / world / peak},
                    'hbm': hbm_line(elapsed / args.steps * 1e3) if c2 else None}
        if roofline['hbm'] and roofline['hbm'].get('bytes_per_step'):
            # SURVEY 8d "algorithmic bytes (secondary)": inputs 328 B + mel out 320 B + alignments 4 L B per valid frame, parameters + Adam
            # 28 B x params per optimizer step; DESIGN 4a adds what a training step must keep for its backward pass
            fr, params = done_frames / args.steps / world, model.n_params
            io = fr * (328 + 320) + sum(float((b[0][9] * b[0][5]).sum()) * 4. for b in batches) / len(batches)
            alg = io + 28. * params
            roofline['hbm']['algorithmic_bytes'] = {'io_plus_adam_bytes_per_step': alg, 'traffic_ratio': roofline['hbm']['bytes_per_step'] / alg,
                                                    'with_saved_activations_bytes_per_step': alg + 2. * 28e3 * fr,
                                                    'traffic_ratio_with_saved_activations': roofline['hbm']['bytes_per_step'] / (alg + 2. * 28e3 * fr),
                                                    'note': 'SURVEY 8d: 328 + 320 + 4 L bytes per valid frame + 28 B x parameters (Adam); second '
                                                            'figure adds ~28 KB per frame of saved activations written and read once (DESIGN 4a)'}

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(hp, cpu_batches[0], n_utt=args.cpu_utts, steps=args.cpu_steps)
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
        c2_default = args.batch == 48 and args.tmin == 1 and args.dtype == 'bf16' and world == 1
        if c2_default and not args.no_secondary:
            # the rest of BASELINE.json's metric and the schedules a user of the reference would run, from the SAME invocation (the driver
            # only ever runs `python bench.py --gpus 1`): each is a complete timing of its own, none touches the headline fields above
            del trainer, model, batches
            torch.cuda.empty_cache()
            out['train_16x3'] = secondary_train(dev, 16, 3, 'bf16', steps=15, warmup=6)
            seq = secondary_train(dev, 16, 3, 'bf16', steps=10, warmup=5, group=False)
            out['train_16x3']['sequential_passes'] = {k: seq[k] for k in ('value', 'ms_per_step', 'whole_step')}
            out['fp32'] = secondary_train(dev, 48, 1, 'fp32', steps=6, warmup=4)
            # BASELINE configs[4] (C5): batch 256, 500 <= T <= 1000, adversarial classifier + gradient reversal live (iteration >= 10000)
            out['c5'] = secondary_train(dev, 256, 1, 'bf16', steps=10, warmup=8, pool=2, t_min=500, probe=2)
            out['c5']['workload'] = ('BASELINE configs[4]: long-utterance stress, batch 256, 500<=T<=1000 (utterance 0 = 1000 frames), 11 speakers, '
                                     'adversarial speaker classifier + gradient reversal at full weight, full train step, bf16')
            sargs = argparse.Namespace(**vars(args))
            sargs.batch, sargs.steps, sargs.warmup, sargs.workload = 256, 10, 3, 'synth'
            out['synth'] = synth_bench(sargs, make_hparams(256, 'bf16'), dev, 0, 1, emit=False, cpu_steps=(1, 3))
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    try:
        main()
    except Exception:   # noqa: BLE001 -- a rank that dies must say which one it was (the launcher interleaves the ranks' stderr)
        import traceback
        print(f'[bench rank {os.environ.get("RANK", "0")}/{os.environ.get("WORLD_SIZE", "1")} pid {os.getpid()}] FAILED:\n' + traceback.format_exc(),
              file=sys.stderr, flush=True)
        raise
