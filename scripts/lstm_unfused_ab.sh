# the recurrent step as ONE launch for every cell (SSC_LSTM_UNFUSED_ROWS=1000000) vs the conv kernel's GEMM + the gate kernel from
# 1024 rows on (the default): the Pix2Pix train step, then the secondary workloads whose cells have that many rows
export SSC_DEV_SWITCHES=1
bash scripts/ab_env3.sh SSC_LSTM_UNFUSED_ROWS=1024 SSC_LSTM_UNFUSED_ROWS=1000000
for w in train_mru train_residual bg768 bg768_train fg_infer; do
  bash scripts/ab_secondary.sh $w SSC_LSTM_UNFUSED_ROWS=1024 SSC_LSTM_UNFUSED_ROWS=1000000
done
