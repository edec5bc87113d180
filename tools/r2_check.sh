#!/bin/bash
# mid-round check (one gpurun call): all GPU tests, then the c3 / c5 / c2 bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -q -x -m gpu > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -2 gpurun_out/r2_tests.log
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_c3.json 2> gpurun_out/r2_bench_c3.err || { tail -20 gpurun_out/r2_bench_c3.err; exit 2; }
cut -c1-600 gpurun_out/r2_bench_c3.json
timeout -k 10 400 python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bench_c5.json 2> gpurun_out/r2_bench_c5.err || { tail -20 gpurun_out/r2_bench_c5.err; exit 3; }
cut -c1-600 gpurun_out/r2_bench_c5.json
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r2_bench_c2.json 2> gpurun_out/r2_bench_c2.err || { tail -20 gpurun_out/r2_bench_c2.err; exit 4; }
cut -c1-400 gpurun_out/r2_bench_c2.json
