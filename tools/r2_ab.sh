#!/bin/bash
# A/B after a change (one gpurun call): GPU tests, c3 bench (10 steps), kernel timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-x}; TESTS=${2:-tests}
timeout -k 10 800 python -m pytest $TESTS -q -x -m gpu > gpurun_out/ab_tests_$TAG.log 2>&1 || { tail -40 gpurun_out/ab_tests_$TAG.log; exit 1; }
tail -2 gpurun_out/ab_tests_$TAG.log
bash tools/r2_bench.sh c3 $TAG || exit 2
bash tools/r2_trace.sh c3 $TAG || exit 3
