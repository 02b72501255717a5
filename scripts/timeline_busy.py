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

# where the idle time sits: for every interval with no kernel running, charge it to the kernel that ends it
try:
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    named = sorted(c.execute('select d.start, d.end, s.kernel_name from %s d join %s s on d.kernel_id = s.id' % (kd, ks)).fetchall())
    named = [r for r in named if r[0] >= lo and r[1] <= hi]
    from collections import defaultdict
    gap_by, cnt_by, hist = defaultdict(float), defaultdict(int), defaultdict(int)
    cur_end = named[0][1]
    for s, e, n in named[1:]:
        if s > cur_end:
            g = s - cur_end
            key = n.split('(')[0][:70]
            gap_by[key] += g
            cnt_by[key] += 1
            hist[min(int(g / 1000), 50)] += 1
        cur_end = max(cur_end, e)
    tot = sum(gap_by.values())
    print('idle (no kernel running) %.2f ms of the window = %.1f %%; intervals %d, median-ish histogram (us: count): %s'
          % (tot / 1e6, 100.0 * tot / span, sum(cnt_by.values()),
             ' '.join('%d:%d' % (k, v) for k, v in sorted(hist.items())[:12])))
    for k, v in sorted(gap_by.items(), key=lambda kv: -kv[1])[:14]:
        print('   idle before %-72s %8.1f us  x%-5d avg %5.2f us' % (k, v / 1e3, cnt_by[k], v / 1e3 / cnt_by[k]))
except Exception as ex:     # older databases without the symbol table
    print('gap attribution skipped:', ex)
