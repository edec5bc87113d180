"""GPU: las_gemm vs an fp64 numpy matmul.  f32 mode: exact-f32 MFMA, rel err <= 2e-6*sqrt(K)-ish;
bf16 mode: operands rounded to bf16 (rel 2^-9 each), fp32 accumulate -> compare against the same
rounding applied on the host (tight) and against the unrounded product (loose, 2e-2 of row norm)."""
import importlib
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    importlib.import_module('end-to-end-asr-pytorch_amd')
    return importlib.import_module('end-to-end-asr-pytorch_amd.ops')


def bf16_round(a):
    t = torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float32)
    return t.numpy().astype(np.float64)


SHAPES = [(128, 128, 32), (24, 1280, 320), (2400, 2048, 39), (300, 31, 640), (77, 130, 45), (1, 1, 1),
          (513, 257, 100), (4800, 63, 256)]


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', SHAPES)
def test_gemm(ops, prec, ta, tb, M, N, K):
    rng = np.random.RandomState(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rng.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    B = rng.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    C0 = rng.randn(M, N).astype(np.float32)
    dev = torch.device('cuda:0')
    ops.set_precision(prec)
    try:
        C = torch.tensor(C0, device=dev)
        ops.gemm(torch.tensor(A, device=dev), torch.tensor(B, device=dev), C, transA=bool(ta), transB=bool(tb),
                 alpha=0.5, beta=1.0, bias=torch.tensor(bias, device=dev), act=1)
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    got = C.cpu().numpy().astype(np.float64)
    rA, rB = (A, B) if prec == 'f32' else (bf16_round(A), bf16_round(B))
    opA = rA.T if ta else rA
    opB = rB.T if tb else rB
    want = np.tanh(0.5 * (opA.astype(np.float64) @ opB.astype(np.float64)) + C0 + bias)
    np.testing.assert_allclose(got, want, atol=3e-5 * max(1.0, np.sqrt(K) / 4), rtol=1e-5)


def test_gemm_batched_and_colsum(ops):
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(3)
    b, M, N, K = 5, 70, 40, 33
    A = rng.randn(b, M, K).astype(np.float32)
    B = rng.randn(b, K, N).astype(np.float32)
    ops.set_precision('f32')
    try:
        C = ops.gemm(torch.tensor(A, device=dev), torch.tensor(B, device=dev), batch=b, sA=M * K, sB=K * N)
    finally:
        ops.set_precision('bf16')
    np.testing.assert_allclose(C.cpu().numpy(), np.einsum('bmk,bkn->bmn', A.astype(np.float64), B), atol=1e-4)
    X = rng.randn(1000, 77).astype(np.float32)
    out = torch.ones(77, device=dev)
    ops.colsum(torch.tensor(X, device=dev), out, beta=2.0)
    np.testing.assert_allclose(out.cpu().numpy(), 2.0 + X.astype(np.float64).sum(0), atol=2e-3)
    # two outputs in one pass (las_colsum2: bias_ih / bias_hh), beta = 1 (no scale launch), 0 and another value; a strided source
    Xd = torch.tensor(X, device=dev)
    for beta in (1.0, 0.0, 0.5):
        o1, o2 = torch.full((77,), 3.0, device=dev), torch.full((77,), -1.0, device=dev)
        ops.colsum(Xd, o1, beta=beta, out2=o2)
        ref = X.astype(np.float64).sum(0)
        np.testing.assert_allclose(o1.cpu().numpy(), beta * 3.0 + ref, atol=2e-3)
        np.testing.assert_allclose(o2.cpu().numpy(), beta * -1.0 + ref, atol=2e-3)
    o1 = torch.zeros(30, device=dev)
    ops.colsum(Xd[:, 10:40], o1, beta=1.0)                     # ld = 77, N = 30
    np.testing.assert_allclose(o1.cpu().numpy(), X[:, 10:40].astype(np.float64).sum(0), atol=2e-3)


@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(256, 384, 128), (1000, 328, 520), (136, 2560, 7208)])
def test_gemm_bf16_sources_equal_converted_fp32_sources(ta, tb, M, N, K):
    """las_gemm_ex with bf16 SOURCE operands (the activation twins / weight shadow) against las_gemm on the same values held
    in fp32: identical operand bits after staging, same accumulation order -> equal to 1e-6 of the largest entry; every
    layout, edge tiles (M, N not multiples of 128), a split-K shape, and the bf16 copy of the result."""
    import importlib
    ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
    torch.manual_seed(M + N)
    dev = 'cuda:0'
    A16 = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B16 = torch.randn((N, K) if tb else (K, N), device=dev).to(torch.bfloat16)
    A, B = A16.float(), B16.float()
    bias = torch.randn(N, device=dev)
    ops.set_precision('bf16')
    ref = ops.gemm(A, B, transA=ta, transB=tb, bias=bias, act=1)
    for a16, b16 in [(True, True), (True, False), (False, True)]:
        C16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        got = ops.gemm(A, B, transA=ta, transB=tb, bias=bias, act=1, A16=A16 if a16 else None, B16=B16 if b16 else None, C16=C16)
        torch.cuda.synchronize()
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), (a16, b16)
        assert torch.equal(C16, got.to(torch.bfloat16))
    # accumulate form (beta = 1, split-K eligible: no bias / activation)
    C0 = torch.randn(M, N, device=dev)
    r2 = ops.gemm(A, B, C0.clone(), transA=ta, transB=tb, beta=1.0)
    g2 = ops.gemm(A, B, C0.clone(), transA=ta, transB=tb, beta=1.0, A16=A16, B16=B16)
    torch.cuda.synchronize()
    assert float((g2 - r2).abs().max()) <= 2e-5 * float(r2.abs().max())     # (split-K partials are added with float atomics)


# ------------------------------------------------------------------------------------------------------------------
# gemm_big.hip: the large-tile LDS-DMA kernel las_gemm_ex dispatches to when both operands are bf16 twins and the shape
# fills the chip.  Shapes chosen so that the dispatcher takes each configuration: 256x256 tiles (16 x 16 = 256 tiles),
# 256x128 tiles with ragged edges in M and N and a K tail (K % 64 != 0), and the split-K form (few output tiles, long K:
# the weight-gradient shape).  Reference: fp64 product of the SAME bf16 values -> only the fp32 accumulation order differs.
@pytest.mark.parametrize('ta,tb', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K,epi', [(4096, 4096, 512, True), (3608, 4104, 456, True), (512, 640, 16384, False),
                                       (7200, 2560, 1280, True), (264, 136, 264, True)])
def test_gemm_big_tile(ops, ta, tb, M, N, K, epi):
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(M + 3 * N + 7 * K + 2 * ta + tb)
    A16 = torch.randn((K, M) if ta else (M, K), generator=g).to(torch.bfloat16)
    B16 = torch.randn((N, K) if tb else (K, N), generator=g).to(torch.bfloat16)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    opA = (A16.t() if ta else A16).double()
    opB = (B16.t() if tb else B16).double()
    prod = opA @ opB
    ops.set_precision('bf16')
    Ad, Bd = A16.to(dev), B16.to(dev)
    Af, Bf = Ad.float(), Bd.float()
    if epi:      # alpha, beta, bias, tanh and the bf16 copy of the result
        C = C0.to(dev)
        C16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        got = ops.gemm(Af, Bf, C, transA=ta, transB=tb, alpha=0.125, beta=1.0, bias=bias.to(dev), act=1, A16=Ad, B16=Bd, C16=C16)
        torch.cuda.synchronize()
        want = torch.tanh(0.125 * prod + C0.double() + bias.double())
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 2e-5 * max(1.0, K ** 0.5 / 8), err
        assert torch.equal(C16, got.to(torch.bfloat16))
    else:        # accumulate form (the weight gradients): beta = 1, no bias / activation -> split-K eligible
        C = C0.to(dev)
        got = ops.gemm(Af, Bf, C, transA=ta, transB=tb, beta=1.0, A16=Ad, B16=Bd)
        torch.cuda.synchronize()
        want = prod + C0.double()
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 3e-6 * float(want.abs().max()) * max(1.0, K ** 0.5 / 32), (err, float(want.abs().max()))
