"""One replay of the generator forward + backward graph as text (offset, duration, queue, workgroups, kernel) from a
rocprofv3 --kernel-trace database of scripts/gen_fb_trace.py, plus the idle time between consecutive dispatches per queue.
usage: gfb_timeline.py <results.db> [replays back from the end, default 5]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
rows = c.execute('select d.start, d.end, s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z / '
                 '(d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z), d.queue_id '
                 'from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
# a replay starts with the layout change of the sketches (the only nchw_to_nhwc launch of the graph)
starts = [r[0] for r in rows if 'nchw_to_nhwc' in r[2]]
t0, t1 = starts[-back - 1], starts[-back]


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('conv_ut_kernel', 'ut').replace('conv_wgrad128_kernel', 'wg128').replace('conv_wgrad_kernel', 'wg')
    return n[:60]


sel = [r for r in rows if t0 <= r[0] < t1]
qs = sorted(set(r[4] for r in sel))
print('# replay of %.3f ms, %d dispatches, queues %s' % ((t1 - t0) / 1e6, len(sel), qs))
busy_until = 0
idle = 0.0
small = 0.0
for s, e, n, wgs, q in sel:
    gap = max(0, s - busy_until) if busy_until else 0
    idle += gap
    if (e - s) < 60000:
        small += (e - s)
    busy_until = max(busy_until, e)
    print('%9.1f %8.1f  q%d %6d wg  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), wgs, gap / 1e3, short(n)))
print('# no kernel running: %.1f us; dispatches shorter than 60 us: %.1f us in total' % (idle / 1e3, small / 1e3))
