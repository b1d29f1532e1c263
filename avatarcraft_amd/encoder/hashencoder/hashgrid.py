"""HashEncoder module + autograd function: drop-in for encoder/hashencoder/hashgrid.py:11-142.

Same constructor, parameters (`embeddings` [sum T_l, C] ~ U(-1e-4,1e-4), buffer `offsets` [L+1] int32),
attributes (output_dim, n_params, max_params) and forward(inputs in [-size,size], size) -> [..., L*C].
The level-major [L,B,C] kernel output and the permute back to [B, L*C] follow the reference."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .backend import _backend


class _hash_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
        inputs = inputs.contiguous()
        embeddings = embeddings.contiguous()
        offsets = offsets.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        H = base_resolution
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=inputs.dtype)
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=inputs.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=inputs.dtype)
        _backend.hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx)
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H]
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        calc_grad_inputs = ctx.calc_grad_inputs
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        if calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs)
        else:
            grad_inputs = torch.zeros(1, device=inputs.device, dtype=inputs.dtype)
        _backend.hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs,
                                      dy_dx, grad_inputs)
        if calc_grad_inputs:
            return grad_inputs, grad_embeddings, None, None, None, None
        return None, grad_embeddings, None, None, None, None


hash_encode = _hash_encode.apply


class _hash_encode_stencil(Function):
    """features of x and of clamp(x +- eps e_k, -bound, bound), k = x,y,z: [7, B, L*C] in ONE launch each way (the reference makes
    7 HashEncoder calls per SDF query of the render core, models/instant_nsr.py:627-642,687-704).  x: [B,3] world space, no grad."""

    @staticmethod
    def forward(ctx, x, embeddings, offsets, per_level_scale, base_resolution, eps, bound):
        x = x.contiguous()
        embeddings = embeddings.contiguous()
        B = x.shape[0]
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = np.log2(per_level_scale)
        out = torch.empty(7, L, B, C, device=x.device, dtype=x.dtype)
        _backend.hash_stencil_forward(x, embeddings, offsets, out, B, C, L, S, base_resolution, eps, bound)
        ctx.save_for_backward(x, embeddings, offsets)
        ctx.cfg = (B, C, L, S, base_resolution, eps, bound)
        return out.permute(0, 2, 1, 3).reshape(7, B, L * C)

    @staticmethod
    def backward(ctx, grad):
        x, embeddings, offsets = ctx.saved_tensors
        B, C, L, S, H, eps, bound = ctx.cfg
        grad = grad.view(7, B, L, C).permute(0, 2, 1, 3).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        _backend.hash_stencil_backward(grad, x, offsets, grad_embeddings, B, C, L, S, H, eps, bound)
        return None, grad_embeddings, None, None, None, None, None


hash_encode_stencil = _hash_encode_stencil.apply


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        if level_dim % 2 != 0:
            print('[WARN] detected HashGrid level_dim % 2 != 0, which will cause very slow backward is also enabled fp16! (maybe fix later)')
        self.max_params = 2 ** log2_hashmap_size
        offsets, offset = [], 0
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            offsets.append(offset)
            offset += min(self.max_params, (resolution + 1) ** input_dim)
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = self.offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"HashEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base_resolution={self.base_resolution} per_level_scale={self.per_level_scale} params={tuple(self.embeddings.shape)}")

    def forward(self, inputs, size=1):
        inputs = (inputs + size) / (2 * size)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = hash_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad)
        return outputs.view(prefix_shape + [self.output_dim])

    def forward_stencil(self, x, size, eps):
        """x [B,3] in [-size,size] -> [7, B, L*C]: the encodings of x, x+eps e_x, x-eps e_x, ... (offsets clamped to the bound)"""
        if self.input_dim != 3 or self.level_dim != 2:
            raise RuntimeError("forward_stencil: input_dim 3 and level_dim 2 only")
        return hash_encode_stencil(x.reshape(-1, 3), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, float(eps),
                                   float(size))
