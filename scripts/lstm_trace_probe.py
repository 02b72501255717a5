"""Durations of the recurrent-step launches of a kernel-trace database by grid size, and of one step's neighbours in start order
(what runs beside / between the steps of the caption branch).  usage: lstm_trace_probe.py <db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]
ks = [x for x in t if 'kernel_symbol' in x][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % kd)]
gx = 'grid_size_x' if 'grid_size_x' in cols else 'grid_x'
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.{gx}, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id "
                 f"order by d.start").fetchall()
agg = {}
for name, s, e, g, q in rows:
    if 'lstm_step' in name:
        a = agg.setdefault((name[:40], g), [])
        a.append((e - s) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print(k, 'n', len(v), 'min %.1f med %.1f p90 %.1f max %.1f' % (v[0], v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
# one late step: 40 dispatches around the middle of the trace
mid = len(rows) * 3 // 4
t0 = rows[mid][1]
for name, s, e, g, q in rows[mid:mid + 70]:
    print('%9.1f %8.1f q%-3s g%-7s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, g, name[:60]))
