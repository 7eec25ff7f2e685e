#!/usr/bin/env python
"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV (calls, total/avg/min/max us, % of GPU kernel time).
usage: scripts/prof_summary.py <results.db> <out.csv> ["header comment"]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    c = sqlite3.connect(db)
    rows = c.execute('select name, count(*), sum(end - start), min(end - start), max(end - start) from kernels group by name').fetchall()
    tot = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    with open(out, 'w', newline='') as fh:
        if note:
            fh.write('"# ' + note.replace('"', "'") + '"\n')
        w = csv.writer(fh)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct'])
        for name, n, t, mn, mx in rows:
            w.writerow([name[:160], n, round(t / 1e3, 1), round(t / 1e3 / n, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * t / tot, 2)])
    print(f'{len(rows)} kernels, {tot / 1e6:.2f} ms total kernel time -> {out}')


if __name__ == '__main__':
    main()
