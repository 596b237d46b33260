#!/usr/bin/env python
"""Per-dispatch timeline of the median-period training step in a rocprofv3 --kernel-trace sqlite database (development aid).
usage: python tools/timeline.py <results.db> [out.txt]   -- M = busiest queue (main stream), s = other queues"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select d.start,d.end,d.queue_id,d.grid_size_x,d.grid_size_y,d.workgroup_size_x,s.kernel_name "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'adam_' in r[6]]
# bench.py cycles a pool of POOL batches of different sizes: among the later steps that ran the SAME batch as the last one, the one with
# the median period (the profiler's own hiccups land on single steps)
POOL = 4
cand = [i for i in range(max(1, len(idx) // 2), len(idx)) if (len(idx) - 1 - i) % POOL == 0]   # the steps that ran the last step's batch
per = {i: rows[idx[i]][1] - rows[idx[i - 1]][1] for i in cand}
pick = sorted(cand, key=lambda i: per[i])[len(cand) // 2]
step = rows[idx[pick - 1] + 1: idx[pick] + 1]
t0 = step[0][0]
qs = sorted(set(r[2] for r in step))
main_q = max(qs, key=lambda q: sum(1 for r in step if r[2] == q))


def short(n):
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return re.sub(r'\(.*', '', n)[:56]


out = [f"{(r[0] - t0) / 1e3:9.1f} {(r[1] - r[0]) / 1e3:7.1f} {'M' if r[2] == main_q else 's'} g={r[3] // r[5]}x{r[4]} {short(r[6])}" for r in step]
m = [r for r in step if r[2] == main_q]
gaps = sum(max(0, m[i + 1][0] - m[i][1]) for i in range(len(m) - 1))
period = per[pick] / 1e6   # Adam end to Adam end: the step as the stream sees it, launch bubbles between steps included
periods = [(rows[idx[i + 1]][1] - rows[idx[i]][1]) / 1e6 for i in range(len(idx) - 1)]
head = (f"# step period {period:.3f} ms (all: {' '.join(f'{x:.2f}' for x in periods)}); step span {(max(r[1] for r in step) - t0) / 1e6:.3f} ms; main-queue kernel time {sum(r[1] - r[0] for r in m) / 1e6:.3f} ms "
        f"({len(m)} dispatches, gaps {gaps / 1e6:.3f} ms); other queues {sum(r[1] - r[0] for r in step if r[2] != main_q) / 1e6:.3f} ms")
text = head + '\n' + '\n'.join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(text + '\n')
print(head)
