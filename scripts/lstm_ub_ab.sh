# recurrent step: 64 x 32 tiles (SSC_LSTM_UB=1) vs 64 x 64 (2) at the caption branch's shapes (rows x C)
export SSC_DEV_SWITCHES=1
for cfg in "512 576 1152 2304" "1024 576 2304 4608"; do set -- $cfg; C=$1; shift
  for UB in 1 2; do echo "== C=$C SSC_LSTM_UB=$UB"; LSTM_C=$C SSC_LSTM_UB=$UB python scripts/lstm_step_probe.py "$@" 2>&1 | grep -E "bf16x6 full"; done
done
