#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace sqlite database (rocpd schema) into a per-kernel table:
calls, total / average / min / max duration, share of GPU kernel time.
usage: python tools/rocprof_stats.py <results.db> [--skip-first-frac 0.0] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(rocpd_kernel_dispatch)')]
    sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
    name_col = 'display_name' if 'display_name' in sym_cols else 'kernel_name'
    q = f'''select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
            from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
            group by s.{name_col} order by 3 desc'''
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows)
    span = list(cur.execute('select min(start), max(end) from rocpd_kernel_dispatch'))[0]
    print(f'# rocprofv3 --kernel-trace summary ({sys.argv[1].split("/")[-1]})')
    print(f'\nkernel dispatches: {sum(r[1] for r in rows)}; sum of kernel durations: {total / 1e6:.3f} ms; '
          f'first-to-last dispatch span: {(span[1] - span[0]) / 1e6:.3f} ms\n')
    print('| kernel | calls | total ms | avg us | min us | max us | % of kernel time |')
    print('|---|---|---|---|---|---|---|')
    for name, n, tot, mn, mx in rows[:60]:
        print(f'| `{short(name)}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100. * tot / total:.1f} |')


if __name__ == '__main__':
    main()
