#!/bin/bash
# round-2 measurement set, part 2: C5 (6x1024) bench + rocprof stats, c2 / c4 / c1 / libri_vgg lines, 2-rank gloo rehearsal on one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final_bench_c5.json 2> gpurun_out/final_bench_c5.err || { tail gpurun_out/final_bench_c5.err; exit 5; }
cut -c1-300 gpurun_out/final_bench_c5.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof_c5 -o c5 --output-format csv -- python3 bench.py --workload c5 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/final_prof_c5.log 2>&1 || exit 6
for w in c2 c4 libri_vgg; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/final_bench_$w.json 2> gpurun_out/final_bench_$w.err || exit 7
  cut -c1-200 gpurun_out/final_bench_$w.json
done
timeout -k 10 300 python bench.py --workload c1 --cpu-sample-b 8 > gpurun_out/final_bench_c1.json 2> gpurun_out/final_bench_c1.err || exit 8
cut -c1-200 gpurun_out/final_bench_c1.json
LAS_LSTM_NO_XL=1 LAS_DEC_NO_PK=1 LAS_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/final_rank2.err | grep "^{\"metric\"" > gpurun_out/final_rank2.json || { tail gpurun_out/final_rank2.err; exit 9; }
cut -c1-200 gpurun_out/final_rank2.json
