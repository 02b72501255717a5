"""Last N dispatches of a rocprofv3 kernel-trace database in start order: offset us, duration us, queue, kernel (debug aid).
usage: trace_tail.py <results.db> [N]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
rows = c.execute('select d.start, d.end, s.kernel_name, d.queue_id from %s d join %s s on d.kernel_id = s.id order by d.start' % (kd, ks)).fetchall()[-n:]
t0 = rows[0][0]
qs = sorted(set(r[3] for r in rows))
for s, e, nm, q in rows:
    nm = re.sub(r'\(.*$', '', re.sub(r'^void ', '', nm))[:70]
    print('%9.1f %7.1f q%d %s' % ((s - t0) / 1e3, (e - s) / 1e3, qs.index(q), nm))
