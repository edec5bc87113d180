#!/bin/bash
# A/B of the stream switches on the c3 bench (20 steps each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline > gpurun_out/fl_$tag.json 2> gpurun_out/fl_$tag.err || { tail -5 gpurun_out/fl_$tag.err; return 1; }
  python -c "import json;d=json.load(open('gpurun_out/fl_$tag.json'));print('$tag', round(d['ms_per_step'],3), round(d['config']['ms_per_step_incl_h2d'],3))"; }
run all X=1 && run noctc LAS_NO_CTC_BRANCH=1 && run oneside LAS_ONE_SIDE_STREAM=1 && run nolen LAS_NO_LEN_STREAM=1 && run nosplit LAS_NO_DEC_SPLIT=1 && run none LAS_NO_CTC_BRANCH=1 LAS_ONE_SIDE_STREAM=1 LAS_NO_LEN_STREAM=1 LAS_NO_DEC_SPLIT=1 && run all2 X=1
