#!/bin/bash
# kernel timeline of a few c3 steps (start/end/stream per dispatch) -> gpurun_out/trace_<tag>_kernel_trace.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-c3}; TAG=${2:-x}
rm -rf gpurun_out/trace_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/trace_$TAG -o t --output-format csv -- python3 bench.py --workload $W --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/trace_$TAG.log 2>&1 || { tail -20 gpurun_out/trace_$TAG.log; exit 2; }
find gpurun_out/trace_$TAG -name '*kernel_trace.csv' -exec ls -la {} \;
