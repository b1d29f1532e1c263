"""Real spherical-harmonics direction encoding (degree <= 8) on the MI355X: module-level counterpart of the reference's
encoder/shencoder/sphere_harmonics.py:11-83 (same class name, constructor, attributes and call convention; the work is done by
ac_sh_encode_forward / ac_sh_encode_backward, csrc/shencoder.hip)."""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

from .backend import _backend

_MAX_DEGREE = 8


def _new(like, *shape):
    return torch.empty(shape, dtype=like.dtype, device=like.device)


class SphericalHarmonicsFn(torch.autograd.Function):
    """y[b, :] = the degree**2 real SH basis functions at direction x[b, :]; optionally keeps dy/dx for the backward"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)         # sphere_harmonics.py:13 of the reference: half under autocast
    def forward(ctx, directions, degree, want_input_grad):
        x = directions.contiguous()
        n, d = x.shape
        n_out = degree * degree
        y = _new(x, n, n_out)
        jac = _new(x, n, d * n_out) if want_input_grad else _new(x, 1)
        _backend.sh_encode_forward(x, y, n, d, degree, want_input_grad, jac)
        ctx.want_input_grad = bool(want_input_grad)
        if ctx.want_input_grad:
            ctx.save_for_backward(x, jac)
            ctx.shape = (n, d, degree)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    @custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        if not ctx.want_input_grad:
            return None, None, None
        x, jac = ctx.saved_tensors
        n, d, degree = ctx.shape
        dx = torch.zeros_like(x)
        _backend.sh_encode_backward(dy.to(x.dtype).contiguous(), x, n, d, degree, jac, dx)
        return dx, None, None


def sh_encode(inputs, degree, calc_grad_inputs=False):
    return SphericalHarmonicsFn.apply(inputs, degree, calc_grad_inputs)


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < degree <= _MAX_DEGREE, "SH encoder only supports degree in [1, 8]"
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree * degree

    def extra_repr(self):
        return f"input_dim={self.input_dim} degree={self.degree}"

    def __repr__(self):
        return f"SHEncoder: {self.extra_repr()}"

    def forward(self, inputs, size=1):
        lead = inputs.shape[:-1]
        flat = (inputs / size).reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(*lead, self.output_dim)
