#!/usr/bin/env python
"""Headline benchmark: real mel-frames/s of the full LAS+CTC train step (forward -> joint CTC + CE loss -> backward ->
gradient all-reduce -> clip -> optimiser) on synthetic 80-dim fbank batches.

  python bench.py --gpus N --steps K --warmup W      (N>1: one rank per GPU; under torch.distributed.run the ranks are
  the launcher's, otherwise bench.py starts them itself as a child `python -m torch.distributed.run ... bench.py`)

Workloads: c3 (default; BASELINE.json configs[2], the config the metric "LAS+CTC" is quoted on:
LibriSpeech-100h hybrid CTC+attention, ctc_weight 0.5, bf16, 1 GPU), c2 (configs[1]: the same model attention-only), c4
(configs[3]: V=5000, L_max=60), c5 (configs[4]: 6x1024 pBLSTM, V=5000), c1 (configs[0]: timit_example.yaml shapes, fp32
MFMA), libri_vgg / libri_vgg_max (the reference's shipped config/libri_example.yaml with its VGG front-end).
`value` is measured with the batches resident in HBM when the timed region starts; the same loop fed from pinned host
memory (the H2D copy of SURVEY.md §8d inside the timed region) is reported beside it as config.value_incl_h2d.
Prints ONE JSON line (rank 0) with `roofline` (the dominant single kernel, live HIP-event timing) and `cpu_baseline`
(the CPU oracle timed on this host on a bounded sample of the same workload)."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    'c1': dict(D=39, V=63, B=8, T_max=300, L_max=40, prec='f32', opt=('Adam', 1e-4), ctc=0.0, enc=('256_256_256', '2_2_1'),
               att=('dot', 256), dec=256, name='TIMIT phoneme LAS (timit_example.yaml shapes), attention-only, fp32'),
    'c2': dict(D=80, V=31, B=24, T_max=1200, L_max=200, prec='bf16', opt=('Adadelta', 1.0), ctc=0.0,
               enc=('320_320_320_320_320', '2_2_1_1_1'), att=('loc', 300), dec=320,
               name='LibriSpeech-100h char-level LAS, attention-only, bf16, 5x320 pBLSTM (2_2_1_1_1 concat), loc-attn'),
    'c3': dict(D=80, V=31, B=24, T_max=1200, L_max=200, prec='bf16', opt=('Adadelta', 1.0), ctc=0.5,
               enc=('320_320_320_320_320', '2_2_1_1_1'), att=('loc', 300), dec=320,
               name='LibriSpeech-100h hybrid CTC+attention (ctc_weight=0.5), bf16'),
    'c4': dict(D=80, V=5000, B=24, T_max=1200, L_max=60, prec='bf16', opt=('Adadelta', 1.0), ctc=0.5,
               enc=('320_320_320_320_320', '2_2_1_1_1'), att=('loc', 300), dec=320,
               name='LibriSpeech-360h subword (V=5000) LAS+CTC, bf16'),
    # SURVEY.md 8d C5 = BASELINE configs[4]: 6-layer pBLSTM x 1024 (2_2_1_1_1_1 concat), loc-attn A=300, dec 1x1024, V=5000
    'c5': dict(D=80, V=5000, B=24, T_max=1200, L_max=60, prec='bf16', opt=('Adadelta', 1.0), ctc=0.5,
               enc=('1024_1024_1024_1024_1024_1024', '2_2_1_1_1_1'), att=('loc', 300), dec=1024,
               name='LibriSpeech-960h subword (V=5000), 6-layer pBLSTM x 1024 + loc-attn, joint CTC-attn 0.5, bf16'),
    # the reference's shipped config/libri_example.yaml, verbatim: VGG front-end + 5x320 BiLSTM (no pyramid), loc-attn, CTC 0.5
    'libri_vgg': dict(D=80, V=5000, B=24, T_max=1200, L_max=60, prec='bf16', opt=('Adadelta', 1.0), ctc=0.5,
               enc=('320_320_320_320_320', '1_1_1_1_1'), enc_type='VGGBiRNN', style='drop', att=('loc', 300), dec=320,
               name='config/libri_example.yaml: VGGBiRNN (VGG front-end + 5x320 BiLSTM), loc-attn, CTC 0.5, V=5000, bf16'),
    # stress shape, not a headline: the same config at its `max_timestep: 3000` / `max_label_len: 400` limits
    'libri_vgg_max': dict(D=80, V=5000, B=24, T_max=3000, L_max=400, prec='bf16', opt=('Adadelta', 1.0), ctc=0.5,
               enc=('320_320_320_320_320', '1_1_1_1_1'), enc_type='VGGBiRNN', style='drop', att=('loc', 300), dec=320,
               name='config/libri_example.yaml at max_timestep 3000 / max_label_len 400 (stress shape)'),
}


def time_reduction(w):
    synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
    return synth.total_downsample(w['enc'][1]) * (4 if 'VGG' in w.get('enc_type', '') else 1)


def model_cfg(w):
    nl = len(w['enc'][0].split('_'))
    return dict(optimizer=dict(type=w['opt'][0], learning_rate=w['opt'][1], joint_ctc=w['ctc']),
                encoder=dict(enc_type=w.get('enc_type', 'BiRNN'), sample_rate=w['enc'][1], sample_style=w.get('style', 'concat'), dim=w['enc'][0],
                             dropout='_'.join(['0'] * nl), rnn_cell='LSTM'),
                attention=dict(att_mode=w['att'][0], dim=w['att'][1], proj=True, num_head=1),
                decoder=dict(dim=w['dec'], layer=1, dropout=0, rnn_cell='LSTMCell'))


def cpu_baseline(w, cfg, sample_B, steps=1):
    """The oracle (torch-CPU restatement of the reference step, packed-LSTM fast path) on a bounded sample."""
    from oracle import las_ref as R
    synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
    tr = time_reduction(w)
    x, y, lens = synth.make_batch(0, w['B'], w['T_max'], w['D'], w['V'], w['L_max'], tr, ctc=w['ctc'] > 0)
    x, y, lens = x[:sample_B], y[:sample_B], lens[:sample_B]
    torch.manual_seed(0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))           # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    # random-init weights of the architecture (reference init scheme), on the CPU
    asr = importlib.import_module('end-to-end-asr-pytorch_amd.asr')
    shapes = asr.param_shapes(x, w['V'], cfg)
    W = {k: (torch.randn(s) / max(1.0, (s[1] if len(s) > 1 else 1) ** 0.5)).numpy() if len(s) > 1 else torch.zeros(s).numpy()
         for k, s in shapes.items()}
    ref = R.RefTrainStep(W, cfg, fast=True)
    ref.step(x.numpy(), y.numpy())           # untimed: thread pool / allocator warm-up on the same sample
    t0 = time.time()
    for _ in range(steps):
        ref.step(x.numpy(), y.numpy())
    dt = time.time() - t0
    frames = sum(lens) * steps
    return dict(value=frames / dt, unit='mel-frames/s', cores=cores, kind='port',
                sample=f'{steps} full train step(s) of the oracle (after one untimed step) on the first {sample_B} utterances '
                       f'of the step-0 synthetic batch (T_max={w["T_max"]}, {frames} real frames in all, {dt:.1f} s)')


def attach_pmc_traffic(roof, workload):
    """`traffic` = HBM bytes per launch of a single-kernel row from this round's rocprofv3 --pmc passes of THIS command
    (FETCH_SIZE and WRITE_SIZE in separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; a third
    pass with SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE gives `mfma_util`, the fraction of the chip's matrix-pipe cycles
    the launch used), summarised by tools/pmc_summary.py into profiles/r03_pmc_traffic_<workload>.json (tools/r3_profiles.sh).
    The profiler cannot run inside the bench, so a row whose kernel is not in that file keeps traffic = null."""
    path = os.path.join(ROOT, 'profiles', f'r03_pmc_traffic_{workload}.json')
    if not os.path.exists(path):
        return
    pmc = json.load(open(path))
    meta = pmc.pop('_meta', {})
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from pmc_summary import kernel_sources_sha
    stale = meta.get('kernel_sources_sha') != kernel_sources_sha()     # counters were taken on other kernel sources
    # multi-kernel brackets whose HBM traffic is that of ONE launch each of the named kernels (the persistent loops and the
    # post-loop sums of the same C-ABI call; the zero fills and the per-step fallback kernels are not counted)
    group_kernels = {'decoder_bwd (L steps BPTT)': ('dec_pk_bwd_kernel', 'att_dpsi_kernel', 'att_loc_post_mma', 'att_loc_post'),
                     'decoder_fwd (L attend+spell steps)': ('dec_pk_fwd_kernel',)}
    for row in [roof] + roof.get('breakdown', []):
        if not row.get('single_kernel'):
            names = group_kernels.get(row['kernel'])
            per = [[v for n, v in pmc.items() if n.split('<')[0] == k and 'hbm_MB_per_launch_corrected' in v] for k in (names or ())]
            if names and per[0]:
                row['traffic'] = sum(sum(h['hbm_MB_per_launch_corrected'] * h['launches'] for h in hs) / sum(h['launches'] for h in hs)
                                     for hs in per if hs) * 1e6
                row['traffic_unit'] = 'B/call: one launch each of ' + ' + '.join(k for k, hs in zip(names, per) if hs) + ' (PMC, ' + os.path.basename(path) + ')'
                row['traffic_stale'] = stale
            continue
        hits = [v for n, v in pmc.items() if n.split('<')[0] == row['kernel'] and 'hbm_MB_per_launch_corrected' in v]
        if hits:
            n = sum(h['launches'] for h in hits)
            row['traffic'] = sum(h['hbm_MB_per_launch_corrected'] * h['launches'] for h in hits) / n * 1e6
            mu = [h['mfma_util'] for h in hits if h.get('mfma_util') is not None]
            if mu:
                row['mfma_util_pmc'] = sum(mu) / len(mu)
            row['traffic_unit'] = 'B/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, separate passes, ' + os.path.basename(path) + ')'
            row['traffic_stale'] = stale


def launch_ranks_if_needed(a):
    """`python bench.py --gpus N` with no torchrun environment starts the N ranks itself: a CHILD process
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` (one rank per GPU over RCCL),
    whose rank 0 prints the JSON line on the inherited stdout; this process only waits and exits with the child's code.
    Nothing here touches the GPU (the parent never initialises HIP: `torch.cuda.device_count()` does not on this stack).
    Refuses, with a non-zero exit, a rank count the node cannot give one GPU each (RCCL rejects two ranks on one device);
    LAS_DIST_BACKEND=gloo lifts that for a rehearsal of the host logic on fewer GPUs."""
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is not None:
        if int(env_world) != a.gpus:
            print(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}', file=sys.stderr)
            sys.exit(2)
        return
    if a.gpus <= 1:
        return
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < a.gpus and os.environ.get('LAS_DIST_BACKEND', 'nccl') == 'nccl':
        print(f'bench.py: --gpus {a.gpus} asked for, {have} visible: one rank per GPU is required over RCCL '
              '(LAS_DIST_BACKEND=gloo rehearses the host logic on fewer GPUs)', file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note('starting %d ranks: %s' % (a.gpus, ' '.join(cmd)))
    sys.exit(subprocess.call(cmd))


def note(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # steady state = mean over >= 50 steps after >= 10 warm-up steps (the first ~10 steps after start-up run
    # 5-7 % slower: allocator and clock settling; measured 27.2 ms at --warmup 5 against 25.5 ms at 15 and 30)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    # CPU baseline sample: 8 utterances of the step-0 batch, one untimed + 3 timed full train steps (~5 s each on 16 cores at c3)
    ap.add_argument('--cpu-sample-b', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=3)
    a = ap.parse_args()
    launch_ranks_if_needed(a)
    w = WORKLOADS[a.workload]
    cfg = model_cfg(w)

    ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
    solver = importlib.import_module('end-to-end-asr-pytorch_amd.solver')
    synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    world, rank, local = ldist.init()
    if world != a.gpus:                    # never report n_gpus = 1 for a run that was asked for N ranks
        print(f'bench.py: --gpus {a.gpus} but {world} rank(s) are running', file=sys.stderr)
        sys.exit(2)
    tmp = tempfile.mkdtemp(prefix='las_bench_')
    tr = time_reduction(w)
    config = dict(asr_model=cfg, clm=dict(enable=False),
                  solver=dict(dataset='synthetic', data_path='', n_jobs=0, max_timestep=0, max_label_len=0,
                              train_set=['train'], batch_size=w['B'], apex=False, total_steps=10 ** 9, tf_start=1.0,
                              tf_end=1.0, dev_set=['dev'], dev_batch_size=w['B'], dev_step=10 ** 9, test_set=['test'],
                              decode_beam_size=1,
                              synthetic=dict(T_max=w['T_max'], D=w['D'], V=w['V'], L_max=w['L_max'], time_reduction=tr,
                                             n_batches=1)))
    paras = argparse.Namespace(gpu=True, name='bench', config='bench.yaml', seed=0, ckpdir=os.path.join(tmp, 'ckpt'),
                               logdir=os.path.join(tmp, 'log'), load=None, verbose=False, njobs=1)
    torch.manual_seed(0)
    ops.set_precision(w['prec'])
    t = solver.Trainer(config, paras)
    t.load_data()
    t.set_model()
    dev = t.device
    # pre-stage distinct batches (per-step seed = 1234 + global batch index; each rank its own shard): in HBM for the
    # timed region behind `value`, and in pinned host memory for the H2D-inclusive loop
    # (no more distinct batches than warm-up steps: every staged shape has been through the allocator before the timed region)
    n_stage = max(1, min(8, a.warmup, a.steps + a.warmup))
    staged, pinned = [], []
    for i in range(n_stage):
        x, y, lens = synth.make_batch(i * world + rank, w['B'], w['T_max'], w['D'], w['V'], w['L_max'], tr, ctc=w['ctc'] > 0)
        staged.append((x.to(dev), y.to(dev), sum(lens), (lens, int((y != 0).sum(-1).max()))))
        pinned.append((x.pin_memory(), y.pin_memory()))
    t.asr_opt.zero_grad()

    def run(k0, k, known_lengths=False, from_host=False):
        frames = 0
        for i in range(k0, k0 + k):
            x, y, f, hl = staged[i % n_stage]
            ready = True                       # the staged batches are complete in HBM
            if from_host:                      # the reference's per-step H2D (solver.py:132-133), from pinned memory,
                with torch.cuda.stream(ops.copy_stream()):        # on the copy stream as in Trainer.exec
                    x, y = (v.to(dev, non_blocking=True) for v in pinned[i % n_stage])
                    ready = torch.cuda.Event()
                    ready.record(ops.copy_stream())
                for v in (x, y):
                    v.record_stream(torch.cuda.current_stream())
            t.train_step(x, y, 1.0, host_lens=hl if known_lengths else None, inputs_ready=ready)
            frames += f
        return frames

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    note(f'model built ({t.asr_model.n_params} params), {n_stage} batches staged; warmup {a.warmup}')
    # the step's dependency chain runs on a high-priority stream, as in Trainer.exec: where a side stream's kernel and the
    # chain's next kernel are both ready, the chain's goes first
    ms = ops.main_stream()
    ms.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(ms)
    run(0, a.warmup)
    barrier()
    note(f'timing {a.steps} steps')
    t0 = time.perf_counter()
    frames = run(a.warmup, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    stat = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stat[0:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        fsum = stat[1:2].clone()
        torch.distributed.all_reduce(fsum, op=torch.distributed.ReduceOp.SUM)
        dt, frames = float(tmax), float(fsum)
    assert int(t.asr_model.status.item()) == 0, 'persistent LSTM hand-off timed out'
    skipped = bool(t.asr_opt.norm3[2].item())
    # the same loop with the batch coming from pinned host memory inside the timed region
    barrier()
    t1 = time.perf_counter()
    frames_h = run(a.warmup, a.steps, from_host=True)
    barrier()
    dt_h = time.perf_counter() - t1
    stat_h = torch.tensor([dt_h, float(frames_h)], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(stat_h[0:1], op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(stat_h[1:2], op=torch.distributed.ReduceOp.SUM)
    dt_h, frames_h = float(stat_h[0]), float(stat_h[1])

    note(f'{dt * 1e3 / a.steps:.1f} ms/step; kernel timing pass')
    # ---- roofline of the dominant kernel: live HIP-event timing of its launches over 3 more steps
    # The brackets are HIP events on the launch stream, so a bracket also counts any time the GPU waits for the host
    # to enqueue the next kernel.  Each timed step therefore starts with a ~25 ms device-side sleep: the host (≈6 ms
    # of enqueue work per step) gets a full step ahead and the brackets see back-to-back kernels, as in the timed
    # loop above and in the rocprofv3 trace.
    run(a.warmup + a.steps, 1, known_lengths=True)
    prof = ops.enable_kernel_timing()
    for i in range(3):
        if hasattr(torch.cuda, '_sleep'):
            torch.cuda._sleep(50_000_000)
        run(a.warmup + a.steps + 1 + i, 1, known_lengths=True)
    torch.cuda.synchronize()
    roof = ops.kernel_timing_summary(prof)
    ops.disable_kernel_timing()
    attach_pmc_traffic(roof, a.workload)

    if rank == 0:
        out = {
            'metric': 'mel-frames/sec per train step (LAS+CTC, 80-dim fbank)' if w['ctc'] > 0 else
                      'mel-frames/sec per train step (LAS attention-only, 80-dim fbank)',
            'value': frames / dt, 'unit': 'mel-frames/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': dt * 1e3 / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if w['prec'] == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': f"{a.workload}: {w['name']}", 'per_gpu_batch': w['B'], 'global_batch': w['B'] * world,
                       'T_max': w['T_max'], 'L_max': w['L_max'], 'D': w['D'], 'V': w['V'], 'optimizer': w['opt'][0],
                       'joint_ctc': w['ctc'], 'params': int(t.asr_model.n_params), 'parallelism': f'dp{world}',
                       'nan_skipped_last_step': skipped, 'value_incl_h2d': frames_h / dt_h,
                       'ms_per_step_incl_h2d': dt_h * 1e3 / a.steps},
            'roofline': roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            note('cpu baseline (oracle on host cores)')
            out['cpu_baseline'] = cpu_baseline(w, cfg, a.cpu_sample_b, a.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
