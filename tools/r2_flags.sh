#!/bin/bash
# A/B of env switches on a bench workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-c3}; ST=${2:-20}; WU=${3:-10}
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --workload $W --steps $ST --warmup $WU --no-cpu-baseline > gpurun_out/fl_$tag.json 2> gpurun_out/fl_$tag.err || { tail -5 gpurun_out/fl_$tag.err; return 1; }
  python -c "import json;d=json.load(open('gpurun_out/fl_$tag.json'));print('$W $tag', round(d['ms_per_step'],3), round(d['config']['ms_per_step_incl_h2d'],3), [(b['kernel'][:12], round(b['total_ms']/3,2)) for b in d['roofline']['breakdown'] if 'decoder' in b['kernel']])"; }
run base X=1 && run f18 LAS_DEC_PK_CFG=1,8 && run noxl LAS_DEC_NO_XL=1 && run base2 X=1
