"""Sweep of the persistent LSTM kernels' time per dependent timestep over H (workgroups per direction = H/16) and B
(batch slices of <= 12 rows): what does the hand-off cost depend on?  Kernel-only (HIP events around the C call)."""
import importlib, sys
import torch
sys.path.insert(0, '.')
importlib.import_module('end-to-end-asr-pytorch_amd')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
dev = 'cuda:0'
ops.set_precision('bf16')
import os
T = int(os.environ.get("T", 300))
SR = int(os.environ.get("SR", 1))
SHAPES = [(24, 320, 2), (12, 320, 2), (12, 320, 1), (24, 160, 2), (12, 160, 2), (12, 64, 2), (12, 32, 1), (12, 16, 1), (48, 320, 2), (24, 512, 2), (24, 1024, 2), (12, 1024, 2), (24, 768, 2)]
if os.environ.get('ONLY_H'):
    SHAPES = [s_ for s_ in SHAPES if s_[1] == int(os.environ['ONLY_H'])]
for (B, H, ND) in SHAPES:
    I = 64
    x = torch.randn(T, B, I, device=dev, requires_grad=True)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    w_ih = (torch.randn(ND * 4 * H, I, device=dev) / I ** 0.5).requires_grad_(True)
    w_hh = (torch.randn(ND, 4 * H, H, device=dev) / H ** 0.5).requires_grad_(True)
    b_ih = torch.zeros(ND * 4 * H, device=dev, requires_grad=True)
    b_hh = torch.zeros(ND * 4 * H, device=dev, requires_grad=True)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(2):
        y = ops.lstm_layer(x, lens, w_ih, w_hh, b_ih, b_hh, SR, True, status)
        y.backward(torch.ones_like(y))
    ops.join_side_stream(); torch.cuda.synchronize()
    rec = ops.enable_kernel_timing()
    for it in range(3):
        y = ops.lstm_layer(x, lens, w_ih, w_hh, b_ih, b_hh, SR, True, status)
        y.backward(torch.ones_like(y))
    ops.join_side_stream(); torch.cuda.synchronize()
    fw = [e0.elapsed_time(e1) for n, e0, e1, *_ in rec if n.startswith('lstm_fwd')]
    bw = [e0.elapsed_time(e1) for n, e0, e1, *_ in rec if n.startswith('lstm_bwd')]
    ops.disable_kernel_timing()
    print(f'B={B:3d} H={H:4d} ND={ND} WGs/dir={H // 16:3d} slices={(B + 11) // 12}: fwd {min(fw) * 1e3 / T:.2f} us/step  bwd {min(bw) * 1e3 / T:.2f} us/step  status={status.item()}', flush=True)
