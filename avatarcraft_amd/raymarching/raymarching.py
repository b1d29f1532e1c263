"""Occupancy-grid ray marching and packed compositing on the MI355X: the five operators of the reference's
raymarching/raymarching.py:21-188 under the same names and call signatures.

Only composite_rays_train is differentiable, so only it is an autograd Function; the marchers, the inference compositor and the
compaction are plain functions that allocate their outputs and enqueue one native call (csrc/raymarching.hip).  Unlike the
reference's atomicAdd slot reservation the native marchers lay samples out in ray order (deterministic)."""
import torch

from .backend import _backend

__all__ = ["march_rays_train", "composite_rays_train", "march_rays", "composite_rays", "compact_rays"]


def _rays(t):
    return t.float().contiguous().view(-1, 3)


def _round_up(n, align):
    return n + (align - n % align) if align > 0 else n


@torch.no_grad()
def march_rays_train(rays_o, rays_d, bound, density_grid, mean_density, iter_density, step_counter=None, mean_count=-1, perturb=False,
                     align=-1, force_all_rays=False):
    """-> xyzs [M,3], dirs [M,3], deltas [M], rays [N,3] (index, offset, count)"""
    o, d = _rays(rays_o), _rays(rays_d)
    n_rays, grid_res, dev = o.shape[0], density_grid.shape[0], o.device
    budgeted = (not force_all_rays) and mean_count > 0
    capacity = _round_up(mean_count, align) if budgeted else n_rays * 1024
    zeros = lambda *s: torch.zeros(s, dtype=o.dtype, device=dev)
    xyzs, dirs, deltas = zeros(capacity, 3), zeros(capacity, 3), zeros(capacity)
    rays = torch.empty(n_rays, 3, dtype=torch.int32, device=dev)
    counter = step_counter if step_counter is not None else torch.zeros(2, dtype=torch.int32, device=dev)
    _backend.march_rays_train(o, d, density_grid, mean_density, iter_density, bound, n_rays, grid_res, capacity, xyzs, dirs, deltas, rays,
                              counter, perturb)
    if not budgeted:                                   # trim to what was produced (the only host sync of the op)
        used = _round_up(int(counter[0].item()), align)
        xyzs, dirs, deltas = xyzs[:used], dirs[:used], deltas[:used]
    return xyzs, dirs, deltas, rays


class CompositeTrainFn(torch.autograd.Function):
    """weights_sum [N], image [N,3] from packed per-sample (sigma, rgb, delta) and the rays table"""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, bound):
        s, c, dl, r = sigmas.float().contiguous(), rgbs.float().contiguous(), deltas.float().contiguous(), rays.contiguous()
        n_samples, n_rays = s.shape[0], r.shape[0]
        wsum = torch.empty(n_rays, dtype=s.dtype, device=s.device)
        img = torch.empty(n_rays, 3, dtype=s.dtype, device=s.device)
        _backend.composite_rays_train_forward(s, c, dl, r, bound, n_samples, n_rays, wsum, img)
        ctx.save_for_backward(s, c, dl, r, wsum, img)
        ctx.meta = (n_samples, n_rays, bound)
        return wsum, img

    @staticmethod
    def backward(ctx, d_wsum, d_img):
        s, c, dl, r, wsum, img = ctx.saved_tensors
        n_samples, n_rays, bound = ctx.meta
        d_s, d_c = torch.zeros_like(s), torch.zeros_like(c)
        _backend.composite_rays_train_backward(d_wsum.contiguous(), d_img.contiguous(), s, c, dl, r, wsum, img, bound, n_samples, n_rays, d_s, d_c)
        return d_s, d_c, None, None, None


def composite_rays_train(sigmas, rgbs, deltas, rays, bound):
    return CompositeTrainFn.apply(sigmas, rgbs, deltas, rays, bound)


@torch.no_grad()
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_grid, mean_density, near, far, align=-1, perturb=False):
    """inference: up to n_step samples for each of the n_alive rays -> xyzs [M,3], dirs [M,3], deltas [M,2]"""
    o, d = _rays(rays_o), _rays(rays_d)
    capacity = _round_up(n_alive * n_step, align)
    zeros = lambda *s: torch.zeros(s, dtype=o.dtype, device=o.device)
    xyzs, dirs, deltas = zeros(capacity, 3), zeros(capacity, 3), zeros(capacity, 2)
    _backend.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, density_grid.shape[0], density_grid, mean_density, near, far, xyzs,
                        dirs, deltas, perturb)
    return xyzs, dirs, deltas


@torch.no_grad()
def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map):
    """inference: accumulate in place into weights / depth / image / normal_map, advance rays_t, mark finished rays with -1"""
    _backend.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map)
    return tuple()


@torch.no_grad()
def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
    """order-preserving compaction of the still-alive rays"""
    _backend.compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter)
    return tuple()
