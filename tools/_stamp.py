import importlib, sys, torch
sys.path.insert(0, '.')
importlib.import_module('end-to-end-asr-pytorch_amd')
ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops'); lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
dev = 'cuda:0'; ops.set_precision('bf16')
T, B, H, ND, I = 300, 24, 320, 2, 64
L_ = lib.lib()
x = torch.randn(T, B, I, device=dev); lens = torch.full((B,), T, dtype=torch.int32, device=dev)
w_ih = torch.randn(ND * 4 * H, I, device=dev) / 8; w_hh = torch.randn(ND, 4 * H, H, device=dev) / 18
b = torch.zeros(ND * 4 * H, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
xproj = ops.gemm(x.view(T * B, I), w_ih, transB=True)
hf = torch.empty(T, B, ND * H, device=dev); hx = torch.zeros(ND * T * B * H * 2, dtype=torch.uint8, device=dev)
gates = torch.empty(T, B, ND * 4 * H, device=dev); cs = torch.empty(T, B, ND * H, device=dev)
sync = torch.empty(4096, dtype=torch.uint8, device=dev)
P, Ii = lib.P, lib.I
for it in range(3):
    lib.check(L_.las_lstm_rec_fwd(Ii(0), P(xproj.data_ptr()), P(b.data_ptr()), P(b.data_ptr()), P(w_hh.data_ptr()), P(lens.data_ptr()), Ii(T), Ii(B), Ii(H), Ii(ND), Ii(1), Ii(1),
              P(hf.data_ptr()), P(hf.data_ptr()), P(hx.data_ptr()), P(gates.data_ptr()), P(cs.data_ptr()), P(sync.data_ptr()), P(status.data_ptr()), lib.cur_stream()), 'fwd')
torch.cuda.synchronize()
w = sync.view(torch.int64).cpu()
names = ['xproj issue+wait(counter)', 'pull+barrier', 'mfma', 'acc->lds+barrier', 'pointwise+stores', 'drain vmcnt(0)', 'barrier', 'atomic']
n = T - 9
for who, off in (('thread0', 32), ('sync thread', 48)):
    print(who, {nm: round(int(w[off + i]) / n, 1) for i, nm in enumerate(names)}, 'counter ticks per step')
# ---- backward
dy = torch.randn(T, B, ND * H, device=dev)
dgx = torch.empty(ND * T * B * 4 * H * 2, dtype=torch.uint8, device=dev); dgf = torch.empty(T * B, ND * 4 * H, device=dev)
for it in range(3):
    lib.check(L_.las_lstm_rec_bwd(Ii(0), P(dy.data_ptr()), P(gates.data_ptr()), P(cs.data_ptr()), P(w_hh.data_ptr()), P(lens.data_ptr()), Ii(T), Ii(B), Ii(H), Ii(ND), Ii(1), Ii(1),
              P(dgx.data_ptr()), P(dgf.data_ptr()), P(sync.data_ptr()), P(status.data_ptr()), lib.cur_stream()), 'bwd')
torch.cuda.synchronize()
w = sync.view(torch.int64).cpu()
names = ['input loads issue + wait(counter)', 'direct loads + mfma + acc->lds', 'barrier', 'pointwise+stores', 'drain vmcnt(0)', 'barrier2']
for who, off in (('bwd thread0', 32), ('bwd sync thread', 48)):
    print(who, {nm: round(int(w[off + i]) / n, 1) for i, nm in enumerate(names)}, 'ticks per step')
