"""GPU parity of the VGG front-end (las_vgg_fwd / las_vgg_bwd through VGGFn) against golden vectors from the
imported reference (tests/golden/g1_vgg_*.npz: MFCC 2x13 and fbank 1x40 inputs, T%4 != 0, ragged lengths) and,
at a larger size, against the CPU oracle's restatement on seeded inputs.
f32 mode (exact f32 MFMA): 5e-5 + 1e-3 of the largest entry of each tensor.  bf16 mode (bf16 MFMA operands, fp32
accumulate): the forward output within 3e-2 of its largest entry against the fp32 goldens; the gradients are checked
(2e-3 of the largest entry) against the oracle run with the same operand rounding, because with different rounding
the ReLU masks / max-pool winners of near-ties differ and each such flip moves a whole gradient term (measured:
0.4 % output error but 10-40 % max-norm gradient error on the 48-row golden case, all from ~10 flipped winners)."""
import importlib
import os
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def mods():
    importlib.import_module('end-to-end-asr-pytorch_amd')
    return (importlib.import_module('end-to-end-asr-pytorch_amd.ops'),
            importlib.import_module('end-to-end-asr-pytorch_amd.vgg'))


def near(got, ref, prec, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1e-6, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    lim = {'f32': 5e-5 + 1e-3 * scale, 'bf16': 3e-2 * scale, 'bf16_emul': 2e-3 * scale}[prec]
    assert err <= lim, (what, err, lim)


def run_case(mods, prec, x_np, W_np, gy_np, time_major=False):
    ops, vgg = mods
    x = torch.tensor(x_np, device=DEV, requires_grad=True)
    W = {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in W_np.items()}
    ops.set_precision(prec)
    try:
        y = vgg.vgg_extractor(x, W, time_major=time_major, prefix='')
        gy = torch.tensor(gy_np, device=DEV)
        if time_major:
            gy = gy.transpose(0, 1).contiguous()
        (y * gy).sum().backward()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    if time_major:
        y = y.transpose(0, 1)
    return y, x.grad, {k: v.grad for k, v in W.items()}


@pytest.mark.parametrize('time_major', [False, True])
@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('name', ['mfcc26', 'fbank40'])
def test_vgg_golden(mods, name, prec, time_major):
    d = np.load(os.path.join(GOLDEN, f'g1_vgg_{name}.npz'))
    W = {k[2:]: d[k] for k in d.files if k.startswith('w.')}
    y, gx, gw = run_case(mods, prec, d['x'], W, d['gy'], time_major)
    near(y, d['y'], prec, 'y')
    if prec == 'bf16':                      # gradients of the bf16 mode: see test_vgg_vs_oracle
        return
    near(gx, d['gx'], prec, 'gx')
    for k, g in gw.items():
        near(g, d['grad.' + k], prec, k)


@pytest.mark.parametrize('B,T,D', [(3, 50, 80), (2, 37, 39), (1, 4, 13)])
def test_vgg_vs_oracle(mods, B, T, D):
    """Seeded random inputs at sizes the CPU oracle finishes in seconds; both precisions."""
    from oracle import las_ref as R
    g = torch.Generator().manual_seed(100 + T)
    cin = D // 13 if D % 13 == 0 else D // 40
    shapes = [(64, cin), (64, 64), (128, 64), (128, 128)]
    W = {}
    for i, (co, ci) in enumerate(shapes, 1):
        W[f'conv{i}.weight'] = (torch.randn(co, ci, 3, 3, generator=g) / (9 * ci) ** 0.5 * 1.4).requires_grad_(True)
        W[f'conv{i}.bias'] = (0.1 * torch.randn(co, generator=g)).requires_grad_(True)
    x = torch.randn(B, T, D, generator=g).requires_grad_(True)
    lens = [T] * B
    gy = None
    for prec in ('f32', 'bf16'):
        x.grad = None
        for v in W.values():
            v.grad = None
        y, ol = R.vgg_extractor(x, lens, {'V.' + k: v for k, v in W.items()}, prefix='V', bf16_operands=prec == 'bf16')
        assert ol == [T // 4] * B
        if gy is None:
            gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        yy, gx, gw = run_case(mods, prec, x.detach().numpy(), {k: v.detach().numpy() for k, v in W.items()}, gy.numpy())
        tol = 'f32' if prec == 'f32' else 'bf16_emul'
        near(yy, y.detach().numpy(), tol, 'y')
        near(gx, x.grad.numpy(), tol, 'gx')
        for k in W:
            near(gw[k], W[k].grad.numpy(), tol, k)


def test_vgg_rejects_bad_dim(mods):
    ops, vgg = mods
    lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
    with pytest.raises(lib.LasError):
        vgg.get_dims(2, 16, 41)
    with pytest.raises(ValueError):
        vgg.check_dim(41)
