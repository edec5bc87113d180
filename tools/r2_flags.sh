#!/bin/bash
# A/B of env switches on the c3 bench (20 steps each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline > gpurun_out/fl_$tag.json 2> gpurun_out/fl_$tag.err || { tail -5 gpurun_out/fl_$tag.err; return 1; }
  python -c "import json;d=json.load(open('gpurun_out/fl_$tag.json'));print('$tag', round(d['ms_per_step'],3), round(d['config']['ms_per_step_incl_h2d'],3))"; }
run base X=1 && run q8 GPU_MAX_HW_QUEUES=8 && run q16 GPU_MAX_HW_QUEUES=16 && run q8_oneside GPU_MAX_HW_QUEUES=8 LAS_ONE_SIDE_STREAM=1 && run base2 X=1 && run q8b GPU_MAX_HW_QUEUES=8
