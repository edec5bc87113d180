cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1 2 4 7 8; do
  LAS_DBG_ATTF=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/varf$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/varf$m.log 2>&1
  f=$(ls gpurun_out/varf$m/*/*kernel_stats.csv | head -1)
  python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'att_energy_fwd' in r['Name'] or 'skinny_direct' in r['Name'] or 'att_softmax' in r['Name']: print('mask=$m', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us')
"
done
