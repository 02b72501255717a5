#!/bin/bash
# the round's final evidence set, one gpurun call: GPU tests + parity margins, hash-matched PMC passes + replayed-step kernel stats,
# the default bench line (with the PMC summary attached), kernel stats of the secondary train steps, a short soak through the CLI
T=r06
python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests.log 2>&1; tail -2 gpurun_out/${T}_gpu_tests.log
cp gpurun_out/parity.jsonl gpurun_out/${T}_parity.jsonl 2>/dev/null
bash scripts/run_profile_pmc.sh $T
cp gpurun_out/${T}_pmc.json profiles/${T}_pmc.json
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; tail -c 400 gpurun_out/${T}_bench_default.json; echo
bash scripts/run_mru_train_profile.sh; cp gpurun_out/mru_train_kernel_stats.txt gpurun_out/${T}_mru_train_kernel_stats.txt
bash scripts/run_resid_train_profile.sh; cp gpurun_out/resid_train_kernel_stats.txt gpurun_out/${T}_residual_train_kernel_stats.txt
bash scripts/run_bg_train_profile.sh; cp gpurun_out/bg_train_kernel_stats.txt gpurun_out/${T}_bg_train_kernel_stats.txt
rm -rf gpurun_out/prof_mru_train
bash scripts/soak_train.sh > gpurun_out/${T}_soak.txt 2>&1; cat gpurun_out/${T}_soak.txt
