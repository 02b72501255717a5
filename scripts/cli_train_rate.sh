# what a user of the reference's command line gets: seconds per iteration of `obj_colorization_main.py --mode train` (synthetic
# queue unless data/tfrecord/train exists) against bench.py's step on the same box, with the losses read one launch late (default)
# and where they are fetched (SSC_CLI_LAZY_LOSS=0).  usage: cli_train_rate.sh [block type] [batch]
BT=${1:-Pix2Pix}; BS=${2:-32}
R=$(pwd)
for rep in 1 2; do for lazy in 1 0; do for gr in 1 0; do
  rm -rf /tmp/cli_rate && mkdir -p /tmp/cli_rate && cd /tmp/cli_rate
  SSC_TRAIN_GRAPHS=$gr SSC_CLI_LAZY_LOSS=$lazy timeout 900 python $R/obj_colorization_main.py --mode train -bt $BT -si 0 -bs $BS -mi 400 -smf 100000 -swf 100 -clt 100 > log.txt 2>&1
  echo "[SSC_CLI_LAZY_LOSS=$lazy SSC_TRAIN_GRAPHS=$gr] $(grep 'Average time' log.txt | tail -2 | sed 's/.*Average time: //' | tr '\n' ' ')"
  cd $R
done; done; done
python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --no-gen-fb --block-type $BT --batch $BS --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py step', round(b['ms_per_step'],3), 'ms')"
