# FETCH_SIZE / WRITE_SIZE of the encoder_3 filter gradient in a loop: scripts/pmc_traffic_wg3.sh [env...]
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t_$pass
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_t_$pass -o p -- python $R/scripts/conv_microbench.py ${LAYER:-wg3} 10 32 > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob('/tmp/pmc_t_$pass/**/*.db',recursive=True)[0]
c=sqlite3.connect(db)
from collections import defaultdict
acc=defaultdict(list)
for n,cn,v in c.execute('select kernel_name, counter_name, value from counters_collection'):
    acc[n.split('(')[0][:50]].append(v)
for k,v in acc.items():
    if 'wgrad' in k or 'conv_ut' in k: print('$pass', k, 'per launch %.1f MB'%(sum(v)/len(v)*1024/1e6*(2 if '$pass'=='FETCH_SIZE' else 1)))
PY
done
