#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  LAS_DBG_ATT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/attdbg_$v -o p --output-format csv -- python3 bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/attdbg_$v.log 2>&1 || exit 1
  grep -h "att_bwd_step" gpurun_out/attdbg_$v/p_kernel_stats.csv | cut -d, -f1-4 | sed "s/^/dbg=$v /" | cut -c1-160
done
