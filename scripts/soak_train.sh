# short training runs of every block type through the CLI (synthetic data): losses must stay finite
cd /root/repo
for bt in Pix2Pix Residual MRU; do
  rm -rf /tmp/soak_$bt && mkdir -p /tmp/soak_$bt && cd /tmp/soak_$bt
  timeout 900 python /root/repo/obj_colorization_main.py --mode train -bt $bt -si 1 -bs 8 -mi 150 -smf 1000 -swf 10 -clt 50 > log.txt 2>&1
  python - <<PY
import json, glob
f = glob.glob('outputs/*/log/scalars.jsonl')[0]
rows = [json.loads(l) for l in open(f)]
print('$bt', len(rows), 'first', {k: round(v, 3) for k, v in rows[0].items() if 'total' in k}, 'last', {k: round(v, 3) for k, v in rows[-1].items() if 'total' in k})
PY
  tail -2 log.txt | cut -c1-200
  cd /root/repo
done
