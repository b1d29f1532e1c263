"""Autograd-level raymarching operators: drop-in for raymarching/raymarching.py:21-188
(march_rays_train, composite_rays_train, march_rays, composite_rays, compact_rays)."""
import torch
from torch.autograd import Function

from .backend import _backend

__all__ = ["march_rays_train", "composite_rays_train", "march_rays", "composite_rays", "compact_rays"]


class _march_rays_train(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_grid, mean_density, iter_density, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False):
        rays_o = rays_o.float().contiguous().view(-1, 3)
        rays_d = rays_d.float().contiguous().view(-1, 3)
        N = rays_o.shape[0]
        H = density_grid.shape[0]
        M = N * 1024
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        deltas = torch.zeros(M, dtype=rays_o.dtype, device=rays_o.device)
        rays = torch.empty(N, 3, dtype=torch.int32, device=rays_o.device)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=rays_o.device)
        _backend.march_rays_train(rays_o, rays_d, density_grid, mean_density, iter_density, bound, N, H, M, xyzs, dirs, deltas,
                                  rays, step_counter, perturb)
        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, bound):
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        deltas = deltas.float().contiguous()
        rays = rays.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, bound, M, N, weights_sum, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.dims = [M, N, bound]
        return weights_sum, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_image):
        grad_weights_sum = grad_weights_sum.contiguous()
        grad_image = grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        M, N, bound = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, bound,
                                               M, N, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


class _march_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_grid, mean_density, near, far, align=-1,
                perturb=False):
        rays_o = rays_o.float().contiguous().view(-1, 3)
        rays_d = rays_d.float().contiguous().view(-1, 3)
        H = density_grid.shape[0]
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        deltas = torch.zeros(M, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, H, density_grid, mean_density, near, far,
                            xyzs, dirs, deltas, perturb)
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map):
        _backend.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map)
        return tuple()


composite_rays = _composite_rays.apply


class _compact_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
        _backend.compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter)
        return tuple()


compact_rays = _compact_rays.apply
