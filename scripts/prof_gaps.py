#!/usr/bin/env python
"""rocprofv3 kernel trace (rocpd sqlite) -> GPU idle analysis: union of kernel intervals vs wall span, largest gaps and what precedes them.
usage: scripts/prof_gaps.py <results.db> [skip_fraction]   (skip the first fraction of the trace: warm-up)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    if not rows:
        print('no kernels')
        return
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[1] >= cut]
    span = max(r[2] for r in rows) - rows[0][1]
    busy, cur_s, cur_e = 0, rows[0][1], rows[0][2]
    gaps = []
    last_name = rows[0][0]
    for name, s, e in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, name))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = name
    busy += cur_e - cur_s
    print(f'{len(rows)} kernels over {span / 1e6:.2f} ms: GPU busy {busy / 1e6:.2f} ms = {100.0 * busy / span:.1f} %, idle {100.0 * (span - busy) / span:.1f} % in {len(gaps)} gaps')
    hist = {}
    for g, a, b in gaps:
        k = '<2us' if g < 2000 else '<5us' if g < 5000 else '<10us' if g < 10000 else '<50us' if g < 50000 else '>=50us'
        hist.setdefault(k, [0, 0])
        hist[k][0] += 1
        hist[k][1] += g
    for k in ('<2us', '<5us', '<10us', '<50us', '>=50us'):
        if k in hist:
            print(f'  gaps {k:6s}: {hist[k][0]:5d}  total {hist[k][1] / 1e6:.3f} ms')
    agg = {}
    for g, a, b in gaps:
        key = (a[:60], b[:60])
        agg.setdefault(key, [0, 0])
        agg[key][0] += 1
        agg[key][1] += g
    print('  largest idle contributors (kernel before -> kernel after):')
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f'    {t / 1e3:8.1f} us in {n:4d} gaps  {a}  ->  {b}')


if __name__ == '__main__':
    main()
