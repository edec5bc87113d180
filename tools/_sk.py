import importlib, sys, torch, os, ctypes
sys.path.insert(0, '.')
importlib.import_module('end-to-end-asr-pytorch_amd')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops'); lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
L_ = lib.lib(); P, I = lib.P, lib.I
dev = 'cuda:0'
def cell(B, C, E, tag):
    x = torch.randn(B, C + E, device=dev); h = torch.randn(B, C, device=dev); c = torch.randn(B, C, device=dev)
    w_ih = torch.randn(4 * C, C + E, device=dev) / 30; w_hh = torch.randn(4 * C, C, device=dev) / 18; b1 = torch.zeros(4 * C, device=dev)
    ho = torch.empty(B, C, device=dev); co = torch.empty(B, C, device=dev); go = torch.empty(B, 4 * C, device=dev)
    print('##', tag, flush=True)
    for it in range(2):
        lib.check(L_.las_lstm_cell_fwd(I(0), P(x.data_ptr()), ctypes.c_int64(C + E), I(C + E), P(h.data_ptr()), P(c.data_ptr()), P(w_ih.data_ptr()), P(w_hh.data_ptr()), P(b1.data_ptr()), P(b1.data_ptr()), I(B), I(C),
                                       P(ho.data_ptr()), P(co.data_ptr()), P(go.data_ptr()), lib.cur_stream()), 'cell')
        torch.cuda.synchronize()
def lin(B, N, K, tag):
    x = torch.randn(B, K, device=dev); w = torch.randn(N, K, device=dev) / 18; o = torch.empty(B, N, device=dev)
    print('##', tag, flush=True)
    for it in range(2):
        lib.check(L_.las_skinny_linear(I(0), P(x.data_ptr()), ctypes.c_int64(K), P(w.data_ptr()), ctypes.c_int64(K), I(B), I(N), I(K), None, I(1), I(0), P(o.data_ptr()), ctypes.c_int64(N), lib.cur_stream()), 'lin')
        torch.cuda.synchronize()
cell(24, 320, 640, 'cell B24 K=1280')
cell(12, 320, 640, 'cell B12 K=1280')
cell(24, 320, 0 + 320, 'cell B24 K=960')
lin(24, 300, 320, 'q B24 N300 K320')
lin(24, 960, 1280, 'dx B24 N960 K1280')
