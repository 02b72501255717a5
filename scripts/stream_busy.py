"""Per-queue (HIP stream) busy time inside the replayed train iterations, from a rocprofv3 --kernel-trace database: which
stream is the critical path, how much of the others it hides.  usage: stream_busy.py <results.db>"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % kd)]
qcol = [x for x in cols if x in ('queue_id', 'stream_id')]
print('columns:', cols)
rows = c.execute('select d.start, d.end, d.%s, s.kernel_name from %s d join %s s on d.kernel_id = s.id order by d.start' % (qcol[0], kd, ks)).fetchall()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * 0.45, t0 + (t1 - t0) * 0.60
busy, cnt, gemm = defaultdict(float), defaultdict(int), defaultdict(float)
for s, e, q, n in rows:
    if s >= lo and e <= hi:
        busy[q] += e - s
        cnt[q] += 1
        if 'conv_' in n or 'lstm_step' in n or 'narrow' in n:
            gemm[q] += e - s
span = hi - lo
for q in sorted(busy, key=lambda k: -busy[k]):
    print('queue %s: busy %.1f %% of the window (%d kernels), GEMM-like share %.1f %%' % (q, 100.0 * busy[q] / span, cnt[q], 100.0 * gemm[q] / max(busy[q], 1)))
