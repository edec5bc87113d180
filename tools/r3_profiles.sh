#!/bin/bash
# Round-3 measurement set: every profiles/r03_* file comes from this script (run through gpurun in the parts given as $1,
# each part within one call's time limit):
#   part a: GPU tests; bench lines (c3 with the CPU baseline, c2, c4, c5, libri_vgg, c1 with its CPU baseline);
#   part b: rocprofv3 --kernel-trace --stats for c1 .. c5;
#   part c: PMC passes for c3 and c5 -- FETCH_SIZE, WRITE_SIZE (HBM bytes, separate passes as MI355X_MICROARCH.md prescribes)
#           and SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE (MFMA utilisation) -- summarised per kernel by
#           tools/pmc_summary.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
part=${1:-a}
if [ "$part" = a ]; then
  timeout -k 10 900 python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
  tail -1 $O/tests.log
  timeout -k 10 500 python bench.py --cpu-steps 3 > $O/bench_c3.json 2> $O/bench_c3.err || { tail $O/bench_c3.err; exit 2; }
  cut -c1-300 $O/bench_c3.json
  for w in c2 c4 c5 libri_vgg; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail $O/bench_$w.err; exit 3; }
    cut -c1-200 $O/bench_$w.json
  done
  timeout -k 10 300 python bench.py --workload c1 --cpu-sample-b 8 --cpu-steps 5 > $O/bench_c1.json 2> $O/bench_c1.err || { tail $O/bench_c1.err; exit 4; }
  cut -c1-200 $O/bench_c1.json
elif [ "$part" = b ]; then
  for w in c3 c5 c1 c2 c4; do
    rm -rf $O/prof_$w
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o $w --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 3 --no-cpu-baseline > $O/prof_$w.log 2>&1 || { tail $O/prof_$w.log; exit 5; }
    f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1)
    cp "$f" $O/kernel_stats_$w.csv
    find $O/prof_$w -name "*kernel_trace.csv" -delete
    head -4 $O/kernel_stats_$w.csv | cut -c1-160
  done
elif [ "$part" = c ]; then
  for w in c3 c5; do
    for c in FETCH_SIZE WRITE_SIZE MFMA; do
      rm -rf $O/pmc_${w}_$c
      if [ $c = MFMA ]; then ctr="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; else ctr=$c; fi
      timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${w}_$c -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_${w}_$c.log 2>&1 || { tail $O/pmc_${w}_$c.log; exit 6; }
    done
    python tools/pmc_summary.py $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE $O/pmc_traffic_$w.json $O/pmc_${w}_MFMA || exit 7
    find $O -name "*kernel_trace.csv" -delete
  done
fi
