#!/bin/bash
# round-2 measurement set, part 1 (one gpurun call): GPU tests, PMC passes (-> profiles/r02_pmc_traffic_c3.json on the box, so that
# the bench line that follows carries this build's traffic), bench c3 (+cpu baseline), rocprof stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -q -x -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -1 gpurun_out/final_tests.log
rm -rf gpurun_out/final_pmc_FETCH_SIZE gpurun_out/final_pmc_WRITE_SIZE gpurun_out/final_prof_c3
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/final_pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/final_pmc_$c.log 2>&1 || exit 4
done
python tools/pmc_summary.py gpurun_out/final_pmc_FETCH_SIZE gpurun_out/final_pmc_WRITE_SIZE gpurun_out/final_pmc_c3.json || exit 5
cp gpurun_out/final_pmc_c3.json profiles/r02_pmc_traffic_c3.json
rm -rf gpurun_out/final_pmc_FETCH_SIZE/*/*kernel_trace* gpurun_out/final_pmc_WRITE_SIZE/*/*kernel_trace*
timeout -k 10 400 python bench.py > gpurun_out/final_bench_c3.json 2> gpurun_out/final_bench_c3.err || { tail gpurun_out/final_bench_c3.err; exit 2; }
cut -c1-400 gpurun_out/final_bench_c3.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof_c3 -o c3 --output-format csv -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/final_prof_c3.log 2>&1 || exit 3
