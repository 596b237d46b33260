#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 --pmc sqlite database (development aid).
usage: python tools/pmc_report.py results.db [name-filter]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
T = lambda k: [t for t in tabs if k in t][0]
kd, ks, pe, ip = T('kernel_dispatch'), T('kernel_symbol'), T('pmc_event'), T('info_pmc')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
names = {r[0]: r[1] for r in c.execute(f"select id, name from {ip}")}
disp = {r[0]: (r[1], r[2] - r[3]) for r in c.execute(f"select d.event_id, s.kernel_name, d.end, d.start from {kd} d join {ks} s on d.kernel_id = s.id")}
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
dur = defaultdict(float)
seen = set()
for ev, pid, val in c.execute(f"select event_id, pmc_id, value from {pe}"):
    if ev not in disp:
        continue
    k, d = disp[ev]
    if flt not in k:
        continue
    acc[k][names[pid]] += val
    if ev not in seen:
        seen.add(ev)
        cnt[k] += 1
        dur[k] += d
for k in sorted(acc, key=lambda k: -dur[k])[:12]:
    n = cnt[k]
    print(f"{re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', k)[:70]}  calls {n}  avg {dur[k] / n / 1e3:.1f} us")
    for name, v in sorted(acc[k].items()):
        print(f"    {name:28s} {v / n:16.0f}")
