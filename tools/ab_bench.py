#!/usr/bin/env python
"""A/B the training bench over values of one environment switch (development aid):
   python tools/ab_bench.py DX_CONV_RING 0 1 [-- extra bench.py args]"""
import json
import os
import subprocess
import sys

args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--')
    args, extra = args[:i], args[i + 1:]
var, vals = args[0], args[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for v in vals:
        env = dict(os.environ)
        env[var] = v
        out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '20', '--warmup', '8'] + extra,
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()
        d = json.loads(out[-1])
        r = d.get('roofline') or {}
        print(f"{var}={v}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}  kernel {r.get('achieved', 0):.0f} TFLOP/s "
              f"avg {r.get('avg_launch_us', 0):.1f} us", flush=True)
