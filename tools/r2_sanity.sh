#!/bin/bash
# sanity after a rebuild (one gpurun call): all GPU tests, then the c3 bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -q -x -m gpu > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -2 gpurun_out/r2_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_c3.json 2> gpurun_out/r2_bench_c3.err || { tail -20 gpurun_out/r2_bench_c3.err; exit 2; }
cut -c1-900 gpurun_out/r2_bench_c3.json
