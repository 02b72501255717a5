"""Where do a kernel's wave cycles go?  Sums rocprofv3 --pmc counters per kernel from one or more rocpd databases.
usage: pmc_breakdown.py <db> [<db> ...]   (each db = one --pmc pass of the same command)"""
import sqlite3
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection'):
        short = name.split('(')[0].replace('void ', '')[:60]
        acc[short][counter] += float(value)
        cnt[short][counter] += 1
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    n = max(cnt[k].values())
    print('== %s  (%d dispatches)' % (k, n))
    wc = d.get('SQ_WAVE_CYCLES', 0)
    for name, v in sorted(d.items()):
        extra = ''
        if wc and name.startswith('SQ_') and name not in ('SQ_WAVE_CYCLES',):
            extra = '  %6.3f of SQ_WAVE_CYCLES' % (v / wc)
        print('   %-30s %16.0f  per dispatch %14.0f%s' % (name, v, v / n, extra))
    if 'GRBM_GUI_ACTIVE' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        simd = d['GRBM_GUI_ACTIVE'] / 8.0 * 1024
        print('   mfma_busy %.3f   (64 cyc x #MFMA / (1024 SIMDs x kernel cycles))' % (d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd))
