# every secondary workload of bench.py, one JSON line each -> gpurun_out/r02_secondary_workloads.json
mkdir -p gpurun_out
: > gpurun_out/r02_secondary_workloads.json
for w in fg_infer fg_resid fg_mru bg768 bg768_train; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r02_secondary_workloads.json
done
for bt in Residual MRU; do
  timeout 900 python bench.py --block-type $bt --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r02_secondary_workloads.json
done
python - <<'PY'
import json
for l in open('gpurun_out/r02_secondary_workloads.json'):
    l = l.strip()
    if not l:
        continue
    d = json.loads(l)
    print(d['config'].get('workload'), round(d['value'], 1), d['unit'], round(d['ms_per_step'], 2), 'ms/step')
PY
