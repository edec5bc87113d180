#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -q -x -m gpu > gpurun_out/gr_tests.log 2>&1 || { tail -40 gpurun_out/gr_tests.log; exit 1; }
tail -1 gpurun_out/gr_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/gr_bench_c3.json 2> gpurun_out/gr_bench_c3.err || { tail gpurun_out/gr_bench_c3.err; exit 3; }
python - <<'PY'
import json
j=json.load(open('gpurun_out/gr_bench_c3.json'))
print(j['ms_per_step'], j['value'])
for b in j['roofline']['breakdown']: print(b['kernel'], round(b['avg_launch_ms'],3), b.get('us_per_dependent_step'))
PY
