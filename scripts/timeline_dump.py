"""One replayed train iteration as text: every dispatch of the iteration that ends with the last-but-one generator Adam launch
(offset in us from the iteration's start, duration, queue, workgroups, LDS bytes, kernel), from a rocprofv3 --kernel-trace
database.  usage: timeline_dump.py <results.db> [iterations back from the end, default 3]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
rows = c.execute('select d.start, d.end, s.kernel_name, d.grid_size_x * d.grid_size_y * d.grid_size_z / '
                 '(d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z), d.queue_id, d.group_segment_size '
                 'from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()
adam = [r for r in rows if 'adam_tf' in r[2]]
# two Adam launches per iteration (D then G): the iteration = (end of G-adam k-1, end of G-adam k]
g_ends = [r[1] for r in adam][1::2]
t1 = g_ends[-back]
t0 = g_ends[-back - 1]


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('conv_ut_kernel', 'ut').replace('conv_wgrad128_kernel', 'wg128').replace('conv_wgrad_kernel', 'wg')
    return n[:52]


qs = sorted(set(r[4] for r in rows if t0 < r[1] <= t1 + 1))
print('# iteration of %.3f ms; queues %s' % ((t1 - t0) / 1e6, qs))
for s, e, n, wgs, q, lds in rows:
    if e <= t0 or s > t1:
        continue
    print('%9.1f %8.1f  q%d %6d wg %6d B  %s' % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), wgs, lds, short(n)))
