"""Which kernels run ALONE, and how many workgroups do they bring?  From a rocprofv3 kernel trace (rocpd database) of the
replayed train iterations: over the 55-95 % window, the time during which exactly one kernel is in flight, charged to that kernel
and bucketed by its workgroup count (a launch with fewer workgroups than the chip has slots cannot fill it on its own).
usage: alone_time.py <results.db>"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
rows = c.execute('select d.start, d.end, s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z, '
                 'd.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z from %s d join %s s on d.kernel_id = s.id' % (kd, ks)).fetchall()
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * 0.55, t0 + (t1 - t0) * 0.95
ev = []
for i, (s, e, n, g, w) in enumerate(rows):
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1, i))
        ev.append((e, -1, i))
ev.sort()
live = set()
last = lo
alone = defaultdict(float)
alone_wg = defaultdict(float)
tot = {1: 0.0, 2: 0.0, 0: 0.0, 3: 0.0}
for t, d, i in ev:
    dt = t - last
    k = min(len(live), 3)
    tot[k] += dt
    if len(live) == 1:
        j = next(iter(live))
        name = rows[j][2].split('(')[0][:64]
        wgs = rows[j][3] // max(rows[j][4], 1)
        alone[name] += dt
        b = '<256' if wgs < 256 else '<512' if wgs < 512 else '<768' if wgs < 768 else '<1536' if wgs < 1536 else '>=1536'
        alone_wg[b] += dt
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
    last = t
span = hi - lo
print('window %.1f ms: 0 kernels %.1f %%, exactly 1: %.1f %%, 2: %.1f %%, >=3: %.1f %%' %
      (span / 1e6, 100 * tot[0] / span, 100 * tot[1] / span, 100 * tot[2] / span, 100 * tot[3] / span))
print('time alone by workgroup count of the lone kernel (%% of window): ' +
      '  '.join('%s: %.1f' % (k, 100 * v / span) for k, v in sorted(alone_wg.items(), key=lambda kv: -kv[1])))
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:18]:
    print('   alone %-66s %5.1f %%' % (k, 100 * v / span))
