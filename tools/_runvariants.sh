for m in 0 1 3 7 15 8 4; do
  echo "mask=$m $(LAS_DBG_LSTM=$m timeout -k 10 100 python tools/bench_lstm.py 2>&1 | grep -E '^bf16 T=1200')"
done
