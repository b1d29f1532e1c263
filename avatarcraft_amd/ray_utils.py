"""SMPL-guided sample warp and mesh-guided near/far: counterpart of utils/ray_utils.py:62-90 and :270-308.

Same call signatures as the reference.  The reference runs both on the CPU (libigl + numpy fp64, and a torch version that
materialises three [N,V,3] temporaries) with GPU<->CPU copies around them (models/instant_nsr.py:166-172,198-203); here
both are HIP kernels (csrc/warp.hip) and accept either torch CUDA tensors (no copies) or numpy arrays (uploaded, and the
results are returned as numpy, like the reference)."""
import numpy as np
import torch

from . import _lib as L

DEFAULT_GEO_THRESH = 0.05


def _dev(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype).contiguous()


def to_homogeneous(pts):
    if isinstance(pts, torch.Tensor):
        return torch.cat([pts, torch.ones_like(pts[..., 0:1])], axis=-1)
    return np.concatenate([pts, np.ones_like(pts[..., 0:1])], axis=-1)


_ACCEL_CACHE = None


def invalidate_caches():
    """drop the cached search structure of the last mesh (keyed by data_ptr / tensor version / stream: in-place writes that bypass the version
    counter -- `.data.copy_`, raw-pointer kernels -- need this call before the next warp)"""
    global _ACCEL_CACHE
    _ACCEL_CACHE = None


def warp_samples_to_canonical(pts, verts, faces, T, threshold=0.2, device=None, return_torch=None, accel=True):
    """pts [num_rays, num_samples, 3]; verts [V,3]; faces [F,>=3] (first three columns); T [V',4,4] (fp64).
    Returns (can_pts [R,S,3] fp64, can_dirs [R,S,3], closest [R,S,3], mask [R*S] bool) -- numpy if pts is numpy."""
    assert len(pts.shape) == 3, 'pts should have shape [num_rays, num_samples, 3]'
    assert pts.shape[-1] == 3
    as_torch = isinstance(pts, torch.Tensor) if return_torch is None else return_torch
    device = device or (pts.device if isinstance(pts, torch.Tensor) and pts.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    R, S, _ = pts.shape
    p = _dev(pts, torch.float32, device).reshape(-1, 3)
    v = _dev(verts, torch.float32, device).reshape(-1, 3)
    f = _dev(faces, torch.int32, device)[:, :3].contiguous()
    Tm = _dev(T, torch.float64, device).reshape(-1, 4, 4)
    P = p.shape[0]
    can = torch.empty((P, 3), dtype=torch.float64, device=device)
    clo = torch.empty((P, 3), dtype=torch.float64, device=device)
    mask = torch.empty(P, dtype=torch.uint8, device=device)
    st = L.current_stream(device)
    nbytes = int(L.lib().ac_warp_accel_bytes(f.shape[0])) if accel else 0
    if nbytes:                                    # exact culling (same results bit for bit); brute force for meshes it does not cover
        # the per-mesh structure (face tiles + cell grids, ~1 ms to build) is kept for the next call on the same device tensors: the reference
        # calls this function twice per ray batch with one mesh per frame
        key = (v.data_ptr(), v._version, tuple(v.shape), f.data_ptr(), f._version, tuple(f.shape), str(device), st)
        global _ACCEL_CACHE
        if _ACCEL_CACHE is not None and _ACCEL_CACHE[0] == key and isinstance(verts, torch.Tensor) and verts.is_cuda:
            acc = _ACCEL_CACHE[1]
        else:
            acc = torch.empty(nbytes, dtype=torch.uint8, device=device)
            L.check(L.lib().ac_warp_accel_build(v.data_ptr(), f.data_ptr(), v.shape[0], f.shape[0], acc.data_ptr(), nbytes, st), "warp_accel_build")
            _ACCEL_CACHE = (key, acc, v, f) if isinstance(verts, torch.Tensor) and verts.is_cuda else None
        L.check(L.lib().ac_warp_samples_accel(p.data_ptr(), v.data_ptr(), f.data_ptr(), Tm.data_ptr(), P, v.shape[0], f.shape[0], float(threshold),
                                              acc.data_ptr(), can.data_ptr(), None, clo.data_ptr(), None, None, mask.data_ptr(), st),
                "warp_samples_to_canonical")
    else:
        L.check(L.lib().ac_warp_samples(p.data_ptr(), v.data_ptr(), f.data_ptr(), Tm.data_ptr(), P, v.shape[0], f.shape[0], float(threshold),
                                        can.data_ptr(), None, clo.data_ptr(), None, None, mask.data_ptr(), st), "warp_samples_to_canonical")
    can = can.reshape(R, S, 3)
    closest = clo.reshape(R, S, 3)
    dirs = can[:, 1:] - can[:, :-1]
    dirs = torch.cat([dirs, dirs[:, -1:]], dim=1)
    dirs = dirs / torch.linalg.norm(dirs, dim=2, keepdim=True)
    mask = mask.bool()
    if as_torch:
        return can, dirs, closest, mask
    return can.cpu().numpy(), dirs.cpu().numpy(), closest.cpu().numpy(), mask.cpu().numpy()


def geometry_guided_near_far(orig, dir, vert, geo_threshold=DEFAULT_GEO_THRESH):
    """per ray: min / max over the vertex spheres of radius geo_threshold (inf / -inf when the ray misses all of them)"""
    as_torch = isinstance(orig, torch.Tensor)
    device = orig.device if as_torch and orig.is_cuda else torch.device("cuda", torch.cuda.current_device())
    o = _dev(orig, torch.float32, device).reshape(-1, 3)
    d = _dev(dir, torch.float32, device).reshape(-1, 3)
    v = _dev(vert, torch.float32, device).reshape(-1, 3)
    near = torch.empty(o.shape[0], dtype=torch.float32, device=device)
    far = torch.empty_like(near)
    L.check(L.lib().ac_mesh_near_far(o.data_ptr(), d.data_ptr(), v.data_ptr(), o.shape[0], v.shape[0], float(geo_threshold), near.data_ptr(),
                                     far.data_ptr(), L.current_stream(device)), "geometry_guided_near_far")
    if as_torch:
        return near, far
    return near.cpu().numpy(), far.cpu().numpy()


geometry_guided_near_far_torch = geometry_guided_near_far
geometry_guided_near_far_np = geometry_guided_near_far


def batch_add_translation(Ts, transl):
    """[R|t] + [0|transl]  (utils/ray_utils.py:341-352)"""
    assert Ts.shape[0] == transl.shape[0]
    out = Ts.copy()
    out[:, :3, 3] += transl
    return out
