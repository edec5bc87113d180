#!/bin/bash
# round-end measurement set (one gpurun call): GPU tests, bench C2 (+cpu baseline), rocprof stats, PMC passes, C5, 2-rank rehearsal
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -q -x -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
timeout -k 10 400 python bench.py > gpurun_out/final_bench_c2.json 2> gpurun_out/final_bench_c2.err || exit 2
cut -c1-400 gpurun_out/final_bench_c2.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof_c2 -o c2 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_prof_c2.log 2>&1 || exit 3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/final_pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final_pmc_$c.log 2>&1 || exit 4
done
python tools/pmc_summary.py gpurun_out/final_pmc_FETCH_SIZE gpurun_out/final_pmc_WRITE_SIZE gpurun_out/final_pmc_c2.json
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/final_bench_c5.json 2> gpurun_out/final_bench_c5.err || exit 5
cut -c1-300 gpurun_out/final_bench_c5.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof_c5 -o c5 --output-format csv -- python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_prof_c5.log 2>&1 || exit 6
LAS_LSTM_NO_XL=1 LAS_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/final_rank2.err > gpurun_out/final_rank2.json || exit 7
cut -c1-200 gpurun_out/final_rank2.json
rm -rf gpurun_out/final_pmc_FETCH_SIZE/*/*kernel_trace* gpurun_out/final_pmc_WRITE_SIZE/*/*kernel_trace*
