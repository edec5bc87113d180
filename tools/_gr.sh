#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -q -x -m gpu > gpurun_out/gr_tests.log 2>&1 || { tail -40 gpurun_out/gr_tests.log; exit 1; }
tail -1 gpurun_out/gr_tests.log
LAS_POISON=1 timeout -k 10 700 python -m pytest tests -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/gr_bench_c3.json 2> gpurun_out/gr_bench_c3.err || { tail gpurun_out/gr_bench_c3.err; exit 3; }
cut -c1-250 gpurun_out/gr_bench_c3.json
LAS_NO_CTC_OVERLAP=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/gr_bench_c3_old.json 2> gpurun_out/gr_bench_c3.err || { tail gpurun_out/gr_bench_c3.err; exit 3; }
cut -c1-250 gpurun_out/gr_bench_c3_old.json
