"""How long does the HOST need to enqueue one train step (no syncs) vs the GPU to execute it?"""
import importlib, sys, time, argparse, os, tempfile, torch
sys.path.insert(0, '.')
import bench
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
solver = importlib.import_module('end-to-end-asr-pytorch_amd.solver')
synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
w = bench.WORKLOADS[os.environ.get('LAS_W', 'c3')]; cfg = bench.model_cfg(w); tr = synth.total_downsample(w['enc'][1]); tmp = tempfile.mkdtemp()
config = dict(asr_model=cfg, clm=dict(enable=False), solver=dict(dataset='synthetic', data_path='', n_jobs=0, max_timestep=0, max_label_len=0,
              train_set=['train'], batch_size=w['B'], apex=False, total_steps=10**9, tf_start=1.0, tf_end=1.0, dev_set=['dev'], dev_batch_size=w['B'],
              dev_step=10**9, test_set=['test'], decode_beam_size=1, synthetic=dict(T_max=w['T_max'], D=w['D'], V=w['V'], L_max=w['L_max'], time_reduction=tr, n_batches=1)))
paras = argparse.Namespace(gpu=True, name='b', config='b.yaml', seed=0, ckpdir=tmp + '/c', logdir=tmp + '/l', load=None, verbose=False, njobs=1)
ops.set_precision(w['prec'])
t = solver.Trainer(config, paras); t.load_data(); t.set_model()
x, y, lens = synth.make_batch(0, w['B'], w['T_max'], w['D'], w['V'], w['L_max'], tr, ctc=w['ctc'] > 0)
x, y = x.cuda(), y.cuda(); hl = (lens, int((y != 0).sum(-1).max()))
t.asr_opt.zero_grad()
for _ in range(3): t.train_step(x, y, 1.0, host_lens=hl)
torch.cuda.synchronize()
N = 6
t0 = time.perf_counter()
for _ in range(N): t.train_step(x, y, 1.0, host_lens=hl)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue {1e3*(t1-t0)/N:.1f} ms/step ; wall incl. GPU drain {1e3*(t2-t0)/N:.1f} ms/step')
if os.environ.get('LAS_HOST_PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(N): t.train_step(x, y, 1.0, host_lens=hl)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.sort_stats('cumulative').print_stats(40)
# ---- pure host cost: enqueue one step while the GPU is parked in a device-side sleep (nothing can block on a full queue)
if hasattr(torch.cuda, '_sleep'):
    import cProfile, pstats
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        torch.cuda._sleep(400_000_000)        # ~0.2 s
        t0 = time.perf_counter()
        t.train_step(x, y, 1.0, host_lens=hl)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print('pure host enqueue per step (GPU parked): ' + ', '.join(f'{1e3*v:.1f} ms' for v in ts))
    torch.cuda._sleep(400_000_000)
    pr = cProfile.Profile(); pr.enable()
    t.train_step(x, y, 1.0, host_lens=hl)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
