"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table
(the same content as `rocprofv3 --stats` CSV: calls, total/avg/min/max ns, percent)."""
import sqlite3
import sys


def main(db, out=None, skip_first_frac=0.0):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.accum_vgpr_count, s.group_segment_size "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    n0 = int(len(rows) * skip_first_frac)
    rows = rows[n0:]
    agg = {}
    for name, s, e, vg, ag, lds in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0, vg, ag, lds])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    lines = ['# rocprofv3 --kernel-trace summary of %s' % db,
             '# dispatches %d, sum of kernel durations %.3f ms, wall span %.3f ms, GPU busy %.1f%%' %
             (len(rows), tot / 1e6, span / 1e6, 100.0 * tot / span),
             '%-90s %7s %12s %12s %10s %10s %6s %5s %5s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us',
                                                                'max_us', 'pct', 'vgpr', 'agpr', 'lds_B')]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('%-90s %7d %12.1f %12.2f %10.2f %10.2f %6.2f %5s %5s %7s' %
                     (name[:90], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot,
                      a[4], a[5], a[6]))
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
