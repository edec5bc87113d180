#!/bin/bash
# kernel stats of one bench workload: tools/kstat.sh <workload> [name filter regex]   (run on the GPU box through gpurun)
w=${1:-c3}; pat=${2:-.}
O=$GRAFT_REPO_ROOT/gpurun_out/kstat_$w
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $w --output-format csv -- python3 bench.py --workload $w --steps 8 --warmup 4 --no-cpu-baseline > $O/log 2>&1 || { tail $O/log; exit 5; }
python3 - "$O" "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = [int(r['Calls']) for r in rows if 'adadelta' in r['Name'] or 'adam_kernel' in r['Name']][0]
for r in rows:
    if re.search(sys.argv[2], r['Name']):
        m = re.search(r'(\w+)<([^>]*)>', r['Name'])
        n = f'{m.group(1)}<{m.group(2)[:30]}>' if m else r['Name'][:50]
        print(f"{n:56s} {int(r['Calls']) / steps:7.1f}/step  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step")
PY
rm -f $O/*/*kernel_trace.csv $O/*kernel_trace.csv
