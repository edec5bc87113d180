#!/bin/bash
# quick A/B: bench c3 (no cpu baseline), prints ms/step and the breakdown
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-c3}; TAG=${2:-x}
timeout -k 10 300 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2b_${W}_$TAG.json 2> gpurun_out/r2b_${W}_$TAG.err || { tail -20 gpurun_out/r2b_${W}_$TAG.err; exit 2; }
python - <<PY
import json
d=json.load(open('gpurun_out/r2b_${W}_$TAG.json'))
print('$W $TAG ms/step', round(d['ms_per_step'],2), 'incl h2d', round(d['config']['ms_per_step_incl_h2d'],2))
for b in d['roofline']['breakdown']:
    print('   %-45s n=%4d tot/3=%8.3f ms avg=%7.3f ms frac=%.4f %s' % (b['kernel'][:45], b['launches'], b['total_ms']/3, b['avg_launch_ms'], b['frac'], round(b.get('us_per_dependent_step',0),2)))
PY
