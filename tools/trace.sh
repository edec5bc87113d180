#!/bin/bash
# one step's dispatch timeline of a bench workload: tools/trace.sh <workload> [step]   (run on the GPU box through gpurun)
w=${1:-c3}; step=${2:-5}
O=$GRAFT_REPO_ROOT/gpurun_out/trace_$w
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o $w --output-format csv -- python3 bench.py --workload $w --steps 8 --warmup 4 --no-cpu-baseline > $O/log 2>&1 || { tail $O/log; exit 5; }
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f --step $step --min-us ${3:-12} > $O/timeline.txt
rm -f $f
head -150 $O/timeline.txt
