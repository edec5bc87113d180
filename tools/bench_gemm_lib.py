#!/usr/bin/env python
"""What the library GEMM (torch.mm on bf16 operands = hipBLASLt / rocBLAS) does on the shapes of tools/bench_gemm.py: the yardstick for
gemm_big.hip.  bf16 output, and fp32 output where torch.mm(out_dtype=) exists.  GPU only."""
import os
import torch
B = 24
SH = []
if os.environ.get('SHAPES') == 'c5':
    for tag, T, I in (('l0', 1200, 80), ('l1', 600, 4096), ('l2', 300, 4096), ('l3', 300, 2048)):
        SH += [(f'{tag} xproj A*B^T', False, True, T * B, 8192, I)] + ([(f'{tag} dX A*B', False, False, T * B, I, 8192)] if tag != 'l0' else []) + \
              [(f'{tag} dW_ih A^T*B', True, False, 8192, I, T * B), (f'{tag} dW_hh A^T*B', True, False, 4096, 1024, T * B)]
else:
    for tag, T, I in (('l0', 1200, 80), ('l1', 600, 1280), ('l2', 300, 1280), ('l3', 300, 640)):
        SH += [(f'{tag} xproj A*B^T', False, True, T * B, 2560, I)] + ([(f'{tag} dX A*B', False, False, T * B, I, 2560)] if tag != 'l0' else []) + \
              [(f'{tag} dW_ih A^T*B', True, False, 2560, I, T * B), (f'{tag} dW_hh A^T*B', True, False, 1280, 320, T * B)]
dev = 'cuda:0'
for tag, ta, tb, M, N, K in SH:
    A = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
    Bm = torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16)
    a, b = (A.t() if ta else A), (Bm.t() if tb else Bm)
    res = []
    for kind in ('bf16', 'f32'):
        try:
            f = (lambda: torch.mm(a, b)) if kind == 'bf16' else (lambda: torch.mm(a, b, out_dtype=torch.float32))
            for _ in range(3):
                f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            res.append(f'{kind} out {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s')
        except Exception as ex:
            res.append(f'{kind} out: {type(ex).__name__}')
    print(f'{tag:18s} M={M:6d} N={N:5d} K={K:6d}  ' + ' | '.join(res), flush=True)
