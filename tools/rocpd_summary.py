"""tools/rocpd_summary.py -- turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the
per-kernel stats table committed under profiles/ (name, calls, total / avg / min / max ns, %)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"']
    for n, cnt, tot, avg, mn, mx in rows:
        lines.append(f'"{n}",{cnt},{tot},{avg:.0f},{mn},{mx},{100.0 * tot / total:.2f}')
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main(*sys.argv[1:3])
