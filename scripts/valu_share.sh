# Which kernels spend the step's vector-ALU issue cycles?  One PMC pass over 3 eager train iterations, every kernel.
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/valu_share
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/valu_share -o p -- python $R/bench.py --steps 2 --warmup 1 --preheat-seconds 0 --no-graphs --no-kernel-events --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
from collections import defaultdict
db = glob.glob('/tmp/valu_share/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection'):
    k = name.split('(')[0].replace('void ', '')[:58]
    acc[k][counter] += float(value)
    if counter == 'SQ_INSTS_VALU': n[k] += 1
tot_valu = sum(d.get('SQ_ACTIVE_INST_VALU', 0) for d in acc.values())
tot_mfma = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for d in acc.values())
print('total SQ_ACTIVE_INST_VALU %.3g  SQ_VALU_MFMA_BUSY_CYCLES %.3g' % (tot_valu, tot_mfma))
print('%-58s %6s %9s %9s %9s' % ('kernel', 'disp', 'valu_act%', 'mfma%', 'gui_act%'))
tg = sum(d.get('GRBM_GUI_ACTIVE', 0) for d in acc.values())
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_ACTIVE_INST_VALU', 0))[:28]:
    print('%-58s %6d %9.2f %9.2f %9.2f' % (k, n[k], 100 * d.get('SQ_ACTIVE_INST_VALU', 0) / tot_valu,
                                          100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(tot_mfma, 1), 100 * d.get('GRBM_GUI_ACTIVE', 0) / tg))
PY
