cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 16 32 64 79; do
  LAS_DBG_ATT=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var$m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/var$m.log 2>&1
  f=$(ls gpurun_out/var$m/*/*kernel_stats.csv | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'att_bwd_energy' in r['Name'] or 'att_bwd_da' in r['Name'] or 'att_energy_fwd' in r['Name']: print('mask=$m', r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, 'us')
"
done
