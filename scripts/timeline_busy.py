"""How busy is the GPU inside the replayed train iterations?  From a rocprofv3 --kernel-trace rocpd database: over the
last ~half of the run, the fraction of wall time covered by at least one kernel, by at least two (stream overlap), and the
average number of kernels in flight.  usage: timeline_busy.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == 'kernels'] or [t for t in tabs if 'kernel_dispatch' in t]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % view[0])]
s_col = [x for x in cols if x in ('start', 'start_timestamp')][0]
e_col = [x for x in cols if x in ('end', 'end_timestamp')][0]
rows = sorted(c.execute('select %s, %s from %s' % (s_col, e_col, view[0])).fetchall())
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * 0.55
hi = t0 + (t1 - t0) * 0.95
ev = []
for s, e in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1))
        ev.append((e, -1))
ev.sort()
depth, last, busy1, busy2, area = 0, lo, 0, 0, 0
for t, d in ev:
    dt = t - last
    if depth >= 1:
        busy1 += dt
    if depth >= 2:
        busy2 += dt
    area += depth * dt
    depth += d
    last = t
span = hi - lo
print('window %.1f ms: >=1 kernel running %.1f %%, >=2 running %.1f %%, mean kernels in flight %.2f' %
      (span / 1e6, 100.0 * busy1 / span, 100.0 * busy2 / span, area / span))
