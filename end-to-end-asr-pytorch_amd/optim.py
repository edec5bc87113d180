"""Fused clip + optimiser over the model's flat parameter vector (reference solver.py:101-106,178-182)."""
import torch

from . import _lib
from ._lib import I, F, Z, ptr, check, cur_stream
from .ops import LL

GRAD_CLIP = 5.0          # reference solver.py:20


class FlatOptimizer:
    """torch.optim.Adam / Adadelta semantics (lr, eps=1e-8, torch defaults otherwise) on model.flat_params,
    preceded by clip_grad_norm_(GRAD_CLIP) with the reference's NaN guard decided on the device.
    `apex: True` + Adam in the YAML (solver.py:101) maps to this same fused Adam."""

    def __init__(self, model, opt_type, lr, eps=1e-8, world_size=1):
        if opt_type not in ('Adam', 'Adadelta'):
            raise NotImplementedError(f'optimizer {opt_type}: only Adam and Adadelta are built')
        self.model, self.type, self.lr, self.eps = model, opt_type, float(lr), float(eps)
        self.p, self.g = model.flat_params, model.flat_grads
        dev = self.p.device
        self.s1 = torch.zeros_like(self.p)            # Adam m | Adadelta square_avg
        self.s2 = torch.zeros_like(self.p)            # Adam v | Adadelta acc_delta
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.norm3 = torch.zeros(3, dtype=torch.float32, device=dev)   # grad norm, clip coef, skip flag
        self.ws = torch.empty(_lib.lib().las_grad_norm_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.world_size = world_size
        self.p16 = getattr(model, 'flat_params16', None)      # bf16 shadow of the weights, rewritten by the update kernels

    def zero_grad(self):
        self.g.zero_()

    def step(self, zero_grad=True):
        """Clip + update; leaves (norm, coef, skip) in self.norm3 on the device."""
        from . import ops
        ops.join_side_stream()             # weight gradients may still be accumulating on the side stream
        L_ = _lib.lib()
        n = self.p.numel()
        check(L_.las_grad_norm(ptr(self.g), LL(n), F(1.0 / self.world_size), F(GRAD_CLIP), ptr(self.ws), ptr(self.norm3),
                               ptr(self.step_dev), cur_stream()), 'las_grad_norm')
        if self.type == 'Adam':
            check(L_.las_adam_step(ptr(self.p), ptr(self.g), ptr(self.s1), ptr(self.s2), LL(n), F(self.lr), F(0.9), F(0.999),
                                   F(self.eps), ptr(self.norm3), ptr(self.step_dev), I(int(zero_grad)), self._p16(), cur_stream()),
                  'las_adam_step')
        else:
            check(L_.las_adadelta_step(ptr(self.p), ptr(self.g), ptr(self.s1), ptr(self.s2), LL(n), F(self.lr), F(0.9),
                                       F(self.eps), ptr(self.norm3), I(int(zero_grad)), self._p16(), cur_stream()), 'las_adadelta_step')

    def _p16(self):
        from ._lib import P
        return P(self.p16.data_ptr()) if self.p16 is not None and self.p16.is_cuda else None

    def state_dict(self):
        return dict(type=self.type, lr=self.lr, s1=self.s1, s2=self.s2, step=self.step_dev)

    def load_state_dict(self, sd):
        self.s1.copy_(sd['s1']); self.s2.copy_(sd['s2']); self.step_dev.copy_(sd['step'])
