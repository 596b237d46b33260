#!/usr/bin/env python
"""Which hardware queue ran which kernels of the median step (rocprofv3 --kernel-trace db): per queue, dispatch count, busy time and
the kernel names in order -- shows how a hipGraph's executor spread the captured streams over its queues.
usage: python tools/qsplit.py <results.db>"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select d.start,d.end,d.queue_id,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'adam_' in r[3]]
pick = len(idx) - 2
step = rows[idx[pick - 1] + 1: idx[pick] + 1]
t0 = step[0][0]
qs = sorted(set(r[2] for r in step))
print(f'step span {(step[-1][1] - t0) / 1e6:.3f} ms, {len(step)} dispatches, queues {qs}')
wg = lambda n: 'wgrad' in n
for q in qs:
    rs = [r for r in step if r[2] == q]
    print(f'queue {q}: {len(rs)} dispatches, busy {sum(r[1] - r[0] for r in rs) / 1e6:.3f} ms, wgrad-family {sum(1 for r in rs if wg(r[3]))}')
# runs of consecutive same-queue dispatches of the data-gradient chain (non-wgrad kernels): how often does the chain change queue?
chain = [r for r in step if not wg(r[3])]
sw = sum(1 for a, b in zip(chain, chain[1:]) if a[2] != b[2])
print(f'main chain ({len(chain)} kernels) changes queue {sw} times; shares a queue with wgrad kernels on queues',
      sorted(set(r[2] for r in chain) & set(r[2] for r in step if wg(r[3]))))
