#!/usr/bin/env python
"""Times las_gemm (through the C ABI) on the GEMM shapes of the c2 training step -- the x*W_ih^T projections of the five
BiLSTM layers, their dX and dW products -- and prints TFLOP/s per shape.  GPU only."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')

B = 24
SHAPES = []   # (tag, transA, transB, M, N, K)
for tag, T, I in (('l0', 1200, 80), ('l1', 600, 1280), ('l2', 300, 1280), ('l3', 300, 640)):
    SHAPES.append((f'{tag} xproj  A*B^T', False, True, T * B, 2560, I))
    if tag != 'l0':
        SHAPES.append((f'{tag} dX     A*B  ', False, False, T * B, I, 2560))
    SHAPES.append((f'{tag} dW_ih  A^T*B', True, False, 2560, I, T * B))
    SHAPES.append((f'{tag} dW_hh  A^T*B', True, False, 1280, 320, T * B))


if os.environ.get('SHAPES') == 'c5':      # the 6x1024 BiLSTM of BASELINE configs[4]: 4H*ND = 8192 gate columns, 4096- / 2048-wide inputs
    SHAPES = []
    for tag, T, I in (('l0', 1200, 80), ('l1', 600, 4096), ('l2', 300, 4096), ('l3', 300, 2048)):
        SHAPES.append((f'{tag} xproj  A*B^T', False, True, T * B, 8192, I))
        if tag != 'l0':
            SHAPES.append((f'{tag} dX     A*B  ', False, False, T * B, I, 8192))
        SHAPES.append((f'{tag} dW_ih  A^T*B', True, False, 8192, I, T * B))
        SHAPES.append((f'{tag} dW_hh  A^T*B', True, False, 4096, 1024, T * B))


def main():
    dev = torch.device('cuda:0')
    ops.set_precision('bf16') if hasattr(ops, 'set_precision') else None
    reps = int(os.environ.get('REPS', 10))
    for tag, ta, tb, M, N, K in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        Bm = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        tw = dict(A16=A.to(torch.bfloat16), B16=Bm.to(torch.bfloat16)) if os.environ.get('TWINS', '1') == '1' else {}
        for _ in range(2):
            ops.gemm(A, Bm, C, transA=ta, transB=tb, **tw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.gemm(A, Bm, C, transA=ta, transB=tb, **tw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double() if M * N * K < 2e10 else None
        err = '' if ref is None else f'  rel.err {float((C.double() - ref).abs().max() / ref.abs().max()):.1e}'
        print(f'{tag:20s} M={M:6d} N={N:5d} K={K:6d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s{err}', flush=True)


if __name__ == '__main__':
    main()
