#!/usr/bin/env python
"""Turn rocprofv3 --pmc runs of `python bench.py ...` into the committed per-kernel counter summary.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE                                   -d gpurun_out/pmc_fetch -- python bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE                                   -d gpurun_out/pmc_write -- python bench.py ...
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
                                                                              -d gpurun_out/pmc_mfma  -- python bench.py ...
  python tools/pmc_counters.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma --out profiles/r02_counters.json

(separate passes: FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together, MI355X_MICROARCH.md "rocprofv3 PMC slots").
Per kernel (template instance): launches, average duration, FETCH_SIZE bytes (raw and with the gfx950 x2 correction for wide
coalesced reads, MI355X_MICROARCH.md "HBM"), WRITE_SIZE bytes, and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs)
* 256 CUs * 4 SIMDs) (the gfx94x `MfmaUtil` formula; the counter counts cycles in which a SIMD's matrix pipe is busy).
`families` groups the kernels the way bench.py's roofline does (conv_gemm = conv_gemm_kernel + conv_wreg_kernel + conv_sk_kernel + conv_wide_kernel; conv_wgrad = the weight-gradient kernels) and averages per launch, weighted by launches."""
import argparse
import glob
import json
import os
import re
import sqlite3
from collections import defaultdict

N_SIMD = 256 * 4
N_XCD = 8       # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (a 72 us kernel at 2.4 GHz reads 1.38 M), SQ_* summed over all SIMDs
FAMILIES = {'conv_gemm': ('conv_gemm_kernel', 'conv_wreg_kernel', 'conv_sk_kernel', 'conv_wide_kernel'), 'conv_wgrad': ('conv_wgrad', 'wgrad_reduce'),
            'attention': ('attn_',), 'layernorm': ('ln_fwd_kernel', 'ln_bwd_kernel')}


def short(name):
    name = re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:110]


def read_db(path):
    ''' {kernel: {'n': launches, 'ns': total duration, counter: total}} of one rocprofv3 sqlite database '''
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda k: [t for t in tabs if k in t][0]
    kd, ks, pe, ip = T('kernel_dispatch'), T('kernel_symbol'), T('pmc_event'), T('info_pmc')
    names = {r[0]: r[1] for r in c.execute(f'select id, name from {ip}')}
    disp = {r[0]: (short(r[1]), r[2] - r[3]) for r in
            c.execute(f'select d.event_id, s.kernel_name, d.end, d.start from {kd} d join {ks} s on d.kernel_id = s.id')}
    out = defaultdict(lambda: defaultdict(float))
    seen = set()
    for ev, pid, val in c.execute(f'select event_id, pmc_id, value from {pe}'):
        if ev not in disp:
            continue
        k, d = disp[ev]
        out[k][names[pid]] += val
        if ev not in seen:
            seen.add(ev)
            out[k]['n'] += 1
            out[k]['ns'] += d
    return out


def families_of(kernels):
    ''' per-launch averages of each kernel family, weighted by launches '''
    fams = {}
    for fam, keys in FAMILIES.items():
        members = {k: r for k, r in kernels.items() if any(s in k for s in keys)}
        n = sum(r['launches'] for r in members.values())
        if not n:
            continue
        agg = {'launches': n, 'kernels': sorted(members)}
        for field in ('fetch_x2_bytes', 'write_bytes', 'avg_us'):
            if all(field in r for r in members.values()):
                agg[field] = sum(r[field] * r['launches'] for r in members.values()) / n
        if all('SQ_VALU_MFMA_BUSY_CYCLES' in r and r.get('GRBM_GUI_ACTIVE') for r in members.values()):
            agg['mfma_util'] = sum(r['SQ_VALU_MFMA_BUSY_CYCLES'] * r['launches'] for r in members.values()) / \
                (sum(r['GRBM_GUI_ACTIVE'] * r['launches'] for r in members.values()) / N_XCD * N_SIMD)
        fams[fam] = agg
    return fams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('runs', nargs='+', help='rocprofv3 output directories (or .db files), one per --pmc pass')
    ap.add_argument('--out', required=True)
    ap.add_argument('--command', default='python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-probe')
    ap.add_argument('--steps-total', type=int, default=0, help='steps (warm-up + timed) the profiled command ran: adds whole_step = bytes per step over ALL kernels')
    args = ap.parse_args()
    kernels = defaultdict(dict)
    for run in args.runs:
        dbs = [run] if run.endswith('.db') else sorted(glob.glob(os.path.join(run, '**', '*.db'), recursive=True))
        for db in dbs:
            for k, rec in read_db(db).items():
                n = rec['n']
                for name, v in rec.items():
                    if name in ('n', 'ns'):
                        continue
                    kernels[k][name] = v / n
                kernels[k].setdefault('launches', int(n))
                kernels[k]['avg_us'] = rec['ns'] / n / 1e3      # under the counters of the last pass read (they serialise kernels)
    for k, rec in kernels.items():
        if 'FETCH_SIZE' in rec:       # counter unit: KB
            rec['fetch_raw_bytes'] = rec.pop('FETCH_SIZE') * 1024.
            rec['fetch_x2_bytes'] = rec['fetch_raw_bytes'] * 2.
        if 'WRITE_SIZE' in rec:
            rec['write_bytes'] = rec.pop('WRITE_SIZE') * 1024.
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in rec and rec.get('GRBM_GUI_ACTIVE'):
            rec['mfma_util'] = rec['SQ_VALU_MFMA_BUSY_CYCLES'] / (rec['GRBM_GUI_ACTIVE'] / N_XCD * N_SIMD)
    fams = families_of(kernels)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from bench import csrc_sha16
    whole = None
    if args.steps_total > 0:
        tot = lambda field: sum(r.get(field, 0.) * r['launches'] for r in kernels.values()) / args.steps_total
        whole = {'steps': args.steps_total, 'fetch_x2_bytes': tot('fetch_x2_bytes'), 'write_bytes': tot('write_bytes'),
                 'kernel_launches_per_step': sum(r['launches'] for r in kernels.values()) / args.steps_total}
    out = {'command': args.command, 'csrc_sha16': csrc_sha16(), 'whole_step': whole, 'notes': 'per-launch averages; FETCH_SIZE x2-corrected for gfx950 (MI355X_MICROARCH.md, HBM); '
           'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); valu_util likewise from SQ_ACTIVE_INST_VALU * 4', 'families': fams,
           'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1].get('avg_us', 0.) * kv[1].get('launches', 0)))}
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    if whole:
        print('whole step', {k: round(v) for k, v in whole.items()})
    for fam, a in fams.items():
        print(fam, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items() if k != 'kernels'})


if __name__ == '__main__':
    main()
