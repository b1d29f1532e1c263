"""Multiresolution hash-grid encoding on the MI355X: module-level counterpart of the reference's
encoder/hashencoder/hashgrid.py:11-142 (same class name, constructor arguments, parameter `embeddings`, buffer `offsets`,
attributes output_dim / n_params / max_params, and forward(inputs in [-size, size], size) -> [..., L*C]).

Two autograd functions sit on the native library: GridEncodeFn (one point per sample; ac_hash_encode_forward/backward, the
reference's operator) and GridStencilFn (the 7-point finite-difference stencil of the render core in one launch each way;
ac_hash_stencil_forward/backward)."""
import numpy as np
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

from .backend import _backend


def level_layout(input_dim, num_levels, per_level_scale, base_resolution, max_entries):
    """entries per level = min(max_entries, (ceil(base * scale**l) + 1) ** input_dim); returns the prefix sums [L+1] (int32)"""
    # scalar `per_level_scale ** l` level by level: the same expression (and operand types) the reference evaluates, so that a
    # resolution sitting exactly on an integer (level 15 of the default model: 2048) rounds the same way
    sizes = [min(max_entries, (int(np.ceil(base_resolution * per_level_scale ** l)) + 1) ** input_dim) for l in range(num_levels)]
    return np.concatenate([[0], np.cumsum(np.asarray(sizes, np.int64))]).astype(np.int32)


class GridEncodeFn(torch.autograd.Function):
    """float32, float16 and float64 tensors alike (one dtype per call: hashencoder.cu:352 dispatches on inputs.scalar_type()); under autocast the
    operands arrive as half, exactly like the reference's `@custom_fwd(cast_inputs=torch.half)` (hashgrid.py:13)"""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, x01, table, offsets, per_level_scale, base_resolution, want_input_grad=False):
        x01, table, offsets = x01.contiguous(), table.contiguous(), offsets.contiguous()
        n, dim = x01.shape
        levels, feats = offsets.shape[0] - 1, table.shape[1]
        log2_scale = np.log2(per_level_scale)
        level_major = torch.empty(levels, n, feats, device=x01.device, dtype=x01.dtype)          # the kernel's [L,B,C]
        jac = torch.empty((n, levels * dim * feats) if want_input_grad else (1,), device=x01.device, dtype=x01.dtype)
        _backend.hash_encode_forward(x01, table, offsets, level_major, n, dim, feats, levels, log2_scale, base_resolution, want_input_grad, jac)
        ctx.save_for_backward(x01, table, offsets, jac)
        ctx.cfg = (n, dim, feats, levels, log2_scale, base_resolution, bool(want_input_grad))
        return level_major.permute(1, 0, 2).reshape(n, levels * feats)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x01, table, offsets, jac = ctx.saved_tensors
        n, dim, feats, levels, log2_scale, base_resolution, want_input_grad = ctx.cfg
        dy = dy.to(x01.dtype).view(n, levels, feats).permute(1, 0, 2).contiguous()
        d_table = torch.zeros_like(table)
        d_x = torch.zeros_like(x01) if want_input_grad else torch.zeros(1, device=x01.device, dtype=x01.dtype)
        _backend.hash_encode_backward(dy, x01, table, offsets, d_table, n, dim, feats, levels, log2_scale, base_resolution, want_input_grad, jac, d_x)
        return (d_x if want_input_grad else None), d_table, None, None, None, None


def hash_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
    return GridEncodeFn.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs)


class GridStencilFn(torch.autograd.Function):
    """features of x and of clamp(x +- eps e_k, -bound, bound), k = x,y,z: [7, B, L*C] in ONE launch each way (the reference makes
    7 HashEncoder calls per SDF query of the render core, models/instant_nsr.py:627-642,687-704).  x: [B,3] world space, no grad."""

    @staticmethod
    def forward(ctx, x, table, offsets, per_level_scale, base_resolution, eps, bound):
        x, table = x.contiguous(), table.contiguous()
        n, levels, feats = x.shape[0], offsets.shape[0] - 1, table.shape[1]
        log2_scale = np.log2(per_level_scale)
        out = torch.empty(7, levels, n, feats, device=x.device, dtype=x.dtype)
        _backend.hash_stencil_forward(x, table, offsets, out, n, feats, levels, log2_scale, base_resolution, eps, bound)
        ctx.save_for_backward(x, table, offsets)
        ctx.cfg = (n, feats, levels, log2_scale, base_resolution, eps, bound)
        return out.permute(0, 2, 1, 3).reshape(7, n, levels * feats)

    @staticmethod
    def backward(ctx, dy):
        x, table, offsets = ctx.saved_tensors
        n, feats, levels, log2_scale, base_resolution, eps, bound = ctx.cfg
        dy = dy.view(7, n, levels, feats).permute(0, 2, 1, 3).contiguous()
        d_table = torch.zeros_like(table)
        _backend.hash_stencil_backward(dy, x, offsets, d_table, n, feats, levels, log2_scale, base_resolution, eps, bound)
        return None, d_table, None, None, None, None, None


hash_encode_stencil = GridStencilFn.apply


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None):
        super().__init__()
        if desired_resolution is not None:            # geometric progression from base_resolution to desired_resolution
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        if level_dim % 2 != 0:
            print('[WARN] detected HashGrid level_dim % 2 != 0, which will cause very slow backward is also enabled fp16! (maybe fix later)')
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution, self.log2_hashmap_size = per_level_scale, base_resolution, log2_hashmap_size
        self.output_dim = num_levels * level_dim
        self.max_params = 2 ** log2_hashmap_size
        layout = level_layout(input_dim, num_levels, per_level_scale, base_resolution, self.max_params)
        self.register_buffer('offsets', torch.from_numpy(layout))
        self.n_params = self.offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(layout[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.uniform_(self.embeddings, -1e-4, 1e-4)

    def __repr__(self):
        return (f"HashEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base_resolution={self.base_resolution} per_level_scale={self.per_level_scale} params={tuple(self.embeddings.shape)}")

    def forward(self, inputs, size=1):
        lead = inputs.shape[:-1]
        x01 = ((inputs + size) / (2 * size)).view(-1, self.input_dim)           # [-size, size] -> [0, 1]
        out = hash_encode(x01, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, x01.requires_grad)
        return out.view(*lead, self.output_dim)

    def forward_stencil(self, x, size, eps):
        """x [B,3] in [-size,size] -> [7, B, L*C]: the encodings of x, x+eps e_x, x-eps e_x, ... (offsets clamped to the bound)"""
        if self.input_dim != 3 or self.level_dim != 2:
            raise RuntimeError("forward_stencil: input_dim 3 and level_dim 2 only")
        return hash_encode_stencil(x.reshape(-1, 3), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, float(eps),
                                   float(size))
