cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  ls gpurun_out/pmc_$c/*/ | head -5
done
echo "--- 2-rank rehearsal (gloo, both ranks on cuda:0)"
LAS_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>gpurun_out/rank2.err | tee gpurun_out/rank2.json | cut -c1-330
tail -3 gpurun_out/rank2.err
