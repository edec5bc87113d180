for m in 0 15 31 47 79 111 127; do
  echo "mask=$m $(LAS_DBG_LSTM=$m timeout -k 10 100 python tools/bench_lstm.py 2>&1 | grep -E '^bf16 T=1200')"
done
