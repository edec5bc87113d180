"""Micro-benchmark: persistent BiLSTM recurrence, us per timestep (fwd and bwd) at config-C2 layer shapes."""
import importlib, sys, time
import torch
sys.path.insert(0, '.')
importlib.import_module('end-to-end-asr-pytorch_amd')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
dev = 'cuda:0'
for prec in ['bf16', 'f32']:
    ops.set_precision(prec)
    for (T, B, I, H, sr) in [(1200, 24, 80, 320, 2), (600, 24, 1280, 320, 2), (300, 24, 1280, 320, 1), (300, 8, 39, 256, 2)]:
        x = torch.randn(T, B, I, device=dev, requires_grad=True)
        lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        w_ih = (torch.randn(8 * H, I, device=dev) / I ** 0.5).requires_grad_(True)
        w_hh = (torch.randn(2, 4 * H, H, device=dev) / H ** 0.5).requires_grad_(True)
        b_ih = torch.zeros(8 * H, device=dev, requires_grad=True)
        b_hh = torch.zeros(8 * H, device=dev, requires_grad=True)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        for it in range(3):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            y = ops.lstm_layer(x, lens, w_ih, w_hh, b_ih, b_hh, sr, True, status)
            e[1].record()
            y.backward(torch.ones_like(y))
            e[2].record()
            torch.cuda.synchronize()
        print(f'{prec} T={T} B={B} I={I} H={H}: fwd {e[0].elapsed_time(e[1])*1e3/T:.2f} us/step  '
              f'bwd {e[1].elapsed_time(e[2])*1e3/T:.2f} us/step  (incl. GEMMs) status={status.item()}', flush=True)
