cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in 5 15 30; do
timeout -k 10 200 python bench.py --warmup $W --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('warmup', d['warmup'], 'ms', round(d['ms_per_step'],2), 'h2d-incl', round(d['config']['ms_per_step_incl_h2d'],2))"
done
