"""Per-shape GEMM timing at the config-C2 shapes (fwd NT, dgrad NN, wgrad TN)."""
import importlib, sys, torch
sys.path.insert(0, '.')
importlib.import_module('end-to-end-asr-pytorch_amd')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
dev = 'cuda:0'
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
shapes = [  # (name, M, N, K)
    ('L0 xproj', 28800, 2560, 80), ('L1 xproj', 14400, 2560, 1280), ('L2 xproj', 7200, 2560, 1280), ('L3 xproj', 7200, 2560, 640),
    ('L0 proj', 14400, 1280, 1280), ('L1 proj', 7200, 1280, 1280), ('L2 proj', 7200, 640, 640), ('psi', 7200, 300, 640),
    ('char', 3600, 31, 320), ('sq4096', 4096, 4096, 4096)]
tot = {}
for name, M, N, K in shapes:
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); dY = torch.randn(M, N, device=dev)
    for tag, fn, fl in [('fwd  NT', lambda: ops.gemm(X, W, transB=True), 2 * M * N * K),
                        ('dgrad NN', lambda: ops.gemm(dY, W), 2 * M * N * K),
                        ('wgrad TN', lambda: ops.gemm(dY, X, transA=True), 2 * M * N * K)]:
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tot[tag] = tot.get(tag, 0) + ms
        print(f'{name:10s} {tag} M={M:6d} N={N:5d} K={K:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s', flush=True)
print(tot)
