"""Python front end of the fused Instant-NSR renderer (ac_render_rays & friends).

Host-side plumbing only: builds the ac_field / ac_render_opts / ac_render_out structs from torch
CUDA tensors and enqueues the HIP kernels on torch's current stream.  The numerical work of
NeRFRenderer.run (reference models/instant_nsr.py:133-299) is entirely inside
libavatarcraft_hip.so; there is no eager fallback.
"""
import ctypes as C
import math

import os
import torch

from . import _lib as L

_F32 = torch.float32


def _chk(t, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != _F32:
        raise RuntimeError(f"{name} must be a float32 tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class Field:
    """Device-resident parameters of the default NeRFNetwork in the layout ac_field wants:
    table [n_entries,2], offsets (17 host ints), S=log2(per_level_scale), H, and the EFFECTIVE
    (weight-normed) MLP matrices W1[64,35] b1[64] W2[16,64] b2[16] Wc1[64,21] Wc2[64,64] Wc3[3,64]."""

    def __init__(self, table, offsets, per_level_scale, base_resolution, W1, b1, W2, b2, Wc1, Wc2, Wc3, Wc1_sh=None):
        """Wc1_sh [64,16] (optional): NeRFNetwork(use_viewdirs=True) -- the columns of the effective color_net.0 weight that multiply the 16 spherical
        harmonics of the ray direction (columns 3..18 of its [64,37] matrix; Wc1 then holds the other 21: x, normal, geo_feat).  See ac_field.Wc1_sh."""
        offsets = [int(v) for v in offsets]
        if len(offsets) != 17:
            raise RuntimeError("the fused renderer supports the 16-level hash grid only")
        self.t = dict(table=_chk(table, "table", (offsets[-1], 2)), W1=_chk(W1, "W1", (64, 35)), b1=_chk(b1, "b1", (64,)),
                      W2=_chk(W2, "W2", (16, 64)), b2=_chk(b2, "b2", (16,)), Wc1=_chk(Wc1, "Wc1", (64, 21)),
                      Wc2=_chk(Wc2, "Wc2", (64, 64)), Wc3=_chk(Wc3, "Wc3", (3, 64)))
        import numpy as np
        self.S = float(np.float32(np.log2(per_level_scale)))
        self.H = int(base_resolution)
        f = L.ac_field()
        f.table = self.t["table"].data_ptr()
        for i, v in enumerate(offsets):
            f.offsets[i] = v
        f.S = self.S
        f.H = self.H
        for k in ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3"):
            setattr(f, k, self.t[k].data_ptr())
        if Wc1_sh is not None:
            self.t["Wc1_sh"] = _chk(Wc1_sh, "Wc1_sh", (64, 16))
            f.Wc1_sh = self.t["Wc1_sh"].data_ptr()
        self.has_viewdirs = Wc1_sh is not None
        self.c = f
        self.device = table.device
        self.prepared = None

    def prepare(self):
        """ac_field_prepare: lay the weights out once in the order the renderer keeps them in LDS (its 512 workgroups per launch then copy
        the image linearly).  Call again whenever a parameter tensor of this Field is modified in place."""
        if self.prepared is None:
            self.prepared = torch.empty(L.FIELD_PREPARED_BYTES, dtype=torch.uint8, device=self.device)
        self.c.prepared = None
        L.check(L.lib().ac_field_prepare(C.byref(self.c), self.prepared.data_ptr(), L.current_stream(self.device)), "field_prepare")
        self.c.prepared = self.prepared.data_ptr()
        return self


_LIN_CACHE = {}


def linspace_tables(num_steps, device):
    """lin_z = torch.linspace(0,1,num_steps), lin_u = torch.linspace(0.5/16, 1-0.5/16, 16), made on the
    CPU exactly as the reference makes them (instant_nsr.py:155, :34) and cached on the device."""
    key = (int(num_steps), str(device))
    if key not in _LIN_CACHE:
        lin_z = torch.linspace(0.0, 1.0, num_steps, dtype=_F32)
        lin_u = torch.linspace(0. + 0.5 / 16, 1. - 0.5 / 16, steps=16, dtype=_F32)
        _LIN_CACHE[key] = (lin_z.to(device), lin_u.to(device))
    return _LIN_CACHE[key]


PRECISIONS = {"exact": 0, "fast": 1}


class RenderResult(dict):
    """the tensors a render launch produced (a dict), plus `.opts`: the launch's ac_render_opts and the tensors its pointers refer to"""
    opts = None


def _inv_s_arg(inv_s):
    """(float for ac_render_opts.inv_s, device tensor or None for .inv_s_dev): a CUDA tensor (forward_variance()) is handed over as a
    pointer -- the trainable variance never takes a host round trip"""
    if isinstance(inv_s, torch.Tensor):
        if inv_s.is_cuda:
            t = inv_s.detach().reshape(-1)[:1].to(_F32).contiguous()
            return 0.0, t
        return float(inv_s.detach().reshape(-1)[0]), None
    return float(inv_s), None


def render_rays(field, rays_o, rays_d, num_steps=64, upsample_steps=64, bound=1.6, inv_s=1.0, bg=None, noise=None,
                cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, extras=False, debug_indices=False, out=None, events=None, warp=None,
                train_extras=False, near_far=None, precision="exact", skip_masked=False, opacity_only=False):
    """One launch of the fused renderer for N rays.  Returns a dict of CUDA tensors:
    image[N,3] weights_sum[N] depth[N] normal_map[N,3] eik[N,2] gradient_error[] (+ z_vals, weights,
    alpha, color, sdf, gradient when extras; + ss_inds, sort_index when debug_indices; + sdf_out16 [N,T,16], pts [N,T,3] and
    eik_res = (gradient_error, eikonal denominator) when train_extras: what the render-core backward needs).
    inv_s: a float, or forward_variance() as a CUDA tensor (read on the device).
    near_far = (near [N], far [N]): per-ray sampling range that overrides the cube's where finite (the mesh-guided range of a canonical render).
    precision: "exact" (every product an fp32 fma, bit-identical to the CPU oracle) or "fast" (layer 1 of the six finite-difference
    evaluations as a split-bf16 correction of the centre's; sample positions, indices and sdf unchanged bit for bit; ac_render_opts.precision).
    warp = WarpMesh(...) renders in posed space (run(render_can=False)): + can_mid[N,T,3], mask[N,T] views of the scratch.
    opacity_only: the colour network is not evaluated (image = background over a black body); weights_sum, depth, normal_map, gradient_error unchanged.
    skip_masked (posed space only): tiles of 16 samples that the warp masks out entirely are not evaluated (ac_render_opts.skip_masked): image,
    weights_sum, depth, normal_map, weights and alpha unchanged bit for bit; sdf / color / gradient of skipped samples 0, gradient_error over the
    evaluated samples."""
    rays_o = _chk(rays_o.reshape(-1, 3), "rays_o")
    rays_d = _chk(rays_d.reshape(-1, 3), "rays_d")
    N = rays_o.shape[0]
    dev = rays_o.device
    T = num_steps + upsample_steps
    nup = upsample_steps // 16
    lin_z, lin_u = linspace_tables(num_steps, dev)
    res = out if out is not None else RenderResult()

    def buf(name, shape, dtype=_F32):
        t = res.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=dev)
            res[name] = t
        return t
    o = L.ac_render_out()
    o.image = buf("image", (N, 3)).data_ptr()
    o.weights_sum = buf("weights_sum", (N,)).data_ptr()
    o.depth = buf("depth", (N,)).data_ptr()
    o.normal_map = buf("normal_map", (N, 3)).data_ptr()
    o.eik = buf("eik", (N, 2)).data_ptr()
    er = buf("eik_res", (2,))              # (gradient_error, its denominator): reduced by the render launch itself (ac_render_out.eik_reduced)
    o.eik_reduced = er.data_ptr()
    if N == 0:
        er.zero_()
    if extras:
        o.z_vals = buf("z_vals", (N, T)).data_ptr()
        o.weights = buf("weights", (N, T)).data_ptr()
        o.alpha = buf("alpha", (N, T)).data_ptr()
        o.color = buf("color", (N, T, 3)).data_ptr()
        o.sdf = buf("sdf", (N, T)).data_ptr()
        o.gradient = buf("gradient", (N, T, 3)).data_ptr()
    if train_extras:
        o.sdf_out16 = buf("sdf_out16", (N, T, 16)).data_ptr()
        o.pts = buf("pts", (N, T, 3)).data_ptr()
        if SAVE_STENCIL_FEATURES:
            o.feat7 = buf("feat7", (N * T // 16, 14, 64, 4)).data_ptr()
    if debug_indices:
        o.ss_inds = buf("ss_inds", (N, max(nup, 1), 16), torch.int32).data_ptr()
        o.sort_index = buf("sort_index", (N, max(nup, 1), 128), torch.int32).data_ptr()
    if bg is not None:
        bg = _chk(bg.reshape(-1, 3), "bg_color", (N, 3))
    if noise is not None:
        noise = _chk(noise.reshape(N, num_steps), "noise")
    import numpy as np
    inv_s_f, inv_s_t = _inv_s_arg(inv_s)
    nm = fm = None
    if near_far is not None:
        nm, fm = _chk(near_far[0].reshape(-1), "near", (N,)), _chk(near_far[1].reshape(-1), "far", (N,))
    op = L.ac_render_opts(N, int(num_steps), int(upsample_steps), float(bound), inv_s_f, float(cos_anneal_ratio),
                          float(np.float32(0.005 * (1.0 - normal_epsilon_ratio))), int(noise is not None), L.ptr(inv_s_t), L.ptr(nm), L.ptr(fm),
                          PRECISIONS[precision], int(bool(skip_masked) and warp is not None), int(bool(opacity_only)))
    if isinstance(res, RenderResult):
        res.opts = (op, inv_s_t, nm, fm)
    st = L.current_stream(dev)
    if events is not None:          # (start, end) torch.cuda.Event pair around the render kernel only (bench.py roofline)
        events[0].record()
    if warp is None:
        L.check(L.lib().ac_render_rays(C.byref(field.c), C.byref(op), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(bg), L.ptr(noise),
                                       lin_z.data_ptr(), lin_u.data_ptr(), C.byref(o), st), "render_rays")
    else:
        offs = (C.c_size_t * 6)()
        nbytes = L.lib().ac_render_rays_warped_scratch(N, T, offs)
        scratch = buf("_warp_scratch", (max(int(nbytes), 1),), torch.uint8)
        L.check(L.lib().ac_render_rays_warped(C.byref(field.c), C.byref(op), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(bg), L.ptr(noise),
                                              lin_z.data_ptr(), lin_u.data_ptr(), C.byref(warp.c), scratch.data_ptr(), int(nbytes), C.byref(o), st),
                "render_rays_warped")
        res["can_mid"] = scratch[offs[3]:offs[3] + N * T * 12].view(_F32).view(N, T, 3)
        res["mask"] = scratch[offs[4]:offs[4] + N * T].view(N, T)
        if skip_masked and warp.accel is not None and upsample_steps > 0:      # rays the cell grids proved masked out (never sampled): the scratch's last segment
            tail = (N + 255) & ~255
            res["ray_dead"] = scratch[int(nbytes) - tail:int(nbytes) - tail + N]
        if warp.use_mesh_guide:          # the mesh-guided range the launch sampled in (inf where the ray misses the body): what a backward pass needs
            res["near_m"] = scratch[offs[0]:offs[0] + N * 4].view(_F32)
            res["far_m"] = scratch[offs[1]:offs[1] + N * 4].view(_F32)
    if events is not None:
        events[1].record()
    res["gradient_error"] = er[0]
    return res


def handoff_timeouts(device=None):
    """segment hand-offs of the render launches on the CURRENT stream of `device` that timed out (ac_render_handoff_timeouts): 0 on a healthy run"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = C.c_uint32(0)
    with torch.cuda.device(dev):
        L.check(L.lib().ac_render_handoff_timeouts(L.current_stream(dev), C.addressof(n)), "render_handoff_timeouts")
    return int(n.value)


def render_rays_pair(field, rays_o, rays_d, noise2, num_steps=64, upsample_steps=64, bound=1.6, inv_s=1.0, bg2=None, cos_anneal_ratio=1.0,
                     normal_epsilon_ratio=0.0, precision="exact", out=None, events=None, keep_weights=False):
    """ac_render_rays_pair: the same N rays rendered twice in ONE launch -- copy a with noise2[0] / bg2[0] (per-ray outputs only), copy b with
    noise2[1] / bg2[1] (+ everything the render-core backward needs: the training forward).  Returns (a, b): two RenderResult dicts whose tensors are
    the two halves of shared [2N, ...] buffers; b.opts is the N-ray ac_render_opts the backward takes.  Bit-identical to
    render_rays(..., noise=noise2[0]) and render_rays(..., noise=noise2[1], extras=True, train_extras=True)."""
    rays_o = _chk(rays_o.reshape(-1, 3), "rays_o")
    rays_d = _chk(rays_d.reshape(-1, 3), "rays_d")
    N, dev, T = rays_o.shape[0], rays_o.device, num_steps + upsample_steps
    lin_z, lin_u = linspace_tables(num_steps, dev)
    res = out if out is not None else RenderResult()

    def buf(name, shape, dtype=_F32):
        t = res.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=dev)
            res[name] = t
        return t
    o = L.ac_render_out()
    per_ray = {"image": (2 * N, 3), "weights_sum": (2 * N,), "depth": (2 * N,), "normal_map": (2 * N, 3), "eik": (2 * N, 2)}
    for k, shp in per_ray.items():
        setattr(o, k, buf("pair_" + k, shp).data_ptr())
    er2 = buf("pair_eik_res", (2, 2))      # per copy: (gradient_error, its denominator), reduced by the launch itself
    o.eik_reduced = er2.data_ptr()
    if N == 0:
        er2.zero_()
    per_sample = {"z_vals": (N, T), "color": (N, T, 3), "sdf": (N, T), "gradient": (N, T, 3), "sdf_out16": (N, T, 16), "pts": (N, T, 3)}      # what the backward reads
    if keep_weights:
        per_sample.update({"weights": (N, T), "alpha": (N, T)})
    if SAVE_STENCIL_FEATURES:
        per_sample["feat7"] = (N * T // 16, 14, 64, 4)
    for k, shp in per_sample.items():
        setattr(o, k, buf(k, shp).data_ptr())
    noise2 = _chk(noise2.reshape(2 * N, num_steps), "noise2")
    if bg2 is not None:
        bg2 = _chk(bg2.reshape(-1, 3), "bg2", (2 * N, 3))
    import numpy as np
    inv_s_f, inv_s_t = _inv_s_arg(inv_s)
    op = L.ac_render_opts(N, int(num_steps), int(upsample_steps), float(bound), inv_s_f, float(cos_anneal_ratio),
                          float(np.float32(0.005 * (1.0 - normal_epsilon_ratio))), 1, L.ptr(inv_s_t), None, None, PRECISIONS[precision], 0)
    st = L.current_stream(dev)
    if events is not None:
        events[0].record()
    L.check(L.lib().ac_render_rays_pair(C.byref(field.c), C.byref(op), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(bg2), noise2.data_ptr(),
                                        lin_z.data_ptr(), lin_u.data_ptr(), C.byref(o), st), "render_rays_pair")
    if events is not None:
        events[1].record()
    ra, rb = RenderResult(), RenderResult()
    for k in per_ray:
        t = res["pair_" + k]
        ra[k], rb[k] = t[:N], t[N:]
    for k in per_sample:
        rb[k] = res[k]
    ra["eik_res"], rb["eik_res"] = er2[0], er2[1]
    ra["gradient_error"], rb["gradient_error"] = er2[0, 0], er2[1, 0]
    rb.opts = ra.opts = (op, inv_s_t, None, None)
    rb._keep = ra._keep = (noise2, bg2, res)
    return ra, rb


def sample_rays(field, rays_o, rays_d, num_steps=64, upsample_steps=64, bound=1.6, noise=None, near_far=None):
    """the no-grad sampling stage of run() only -> z_vals [N, num_steps + upsample_steps] (identical to render_rays' z_vals)"""
    rays_o = _chk(rays_o.reshape(-1, 3), "rays_o")
    rays_d = _chk(rays_d.reshape(-1, 3), "rays_d")
    N, dev = rays_o.shape[0], rays_o.device
    lin_z, lin_u = linspace_tables(num_steps, dev)
    if noise is not None:
        noise = _chk(noise.reshape(N, num_steps), "noise")
    z = torch.empty((N, num_steps + upsample_steps), dtype=_F32, device=dev)
    nm = fm = None
    if near_far is not None:
        nm, fm = _chk(near_far[0].reshape(-1), "near", (N,)), _chk(near_far[1].reshape(-1), "far", (N,))
    op = L.ac_render_opts(N, int(num_steps), int(upsample_steps), float(bound), 1.0, 1.0, 0.005, int(noise is not None), None, L.ptr(nm), L.ptr(fm), 0, 0)
    L.check(L.lib().ac_sample_rays(C.byref(field.c), C.byref(op), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(noise), lin_z.data_ptr(),
                                   lin_u.data_ptr(), z.data_ptr(), L.current_stream(dev)), "sample_rays")
    return z


class WarpMesh:
    """the posed SMPL mesh of one frame + its per-vertex rest->scene transforms, on the device (ac_warp_mesh)"""

    def __init__(self, verts, faces, Ts, device, threshold=0.05, geo_threshold=0.05, use_mesh_guide=True, accel=True):
        import numpy as np

        def dev(a, dtype):
            if not isinstance(a, torch.Tensor):
                a = torch.from_numpy(np.ascontiguousarray(a))
            return a.to(device=device, dtype=dtype).contiguous()
        # the index check runs on the HOST copy of the faces (a device tensor's max would wait for everything queued on the device: one stall per frame of an
        # animation); faces that arrive as a device tensor have been checked by whoever uploaded them (warp_mesh_sequence)
        fmax = None
        if not (isinstance(faces, torch.Tensor) and faces.is_cuda):
            fmax = int(np.asarray(faces.cpu() if isinstance(faces, torch.Tensor) else faces)[:, :3].max()) if len(faces) else -1
        self.verts = dev(verts, _F32).reshape(-1, 3)
        self.faces = dev(faces, torch.int32)[:, :3].contiguous()
        self.T = dev(Ts, torch.float64).reshape(-1, 4, 4)
        if (fmax is not None and fmax >= self.T.shape[0]) or self.T.shape[0] < self.verts.shape[0]:
            raise RuntimeError("WarpMesh: Ts must hold one 4x4 per vertex")
        # exact-culling acceleration structure for the closest-face search (rebuilt per frame: the posed mesh changes)
        self.accel = None
        nbytes = int(L.lib().ac_warp_accel_bytes(self.faces.shape[0])) if accel else 0
        if nbytes:
            self.accel = torch.empty(nbytes, dtype=torch.uint8, device=device)
            L.check(L.lib().ac_warp_accel_build(self.verts.data_ptr(), self.faces.data_ptr(), self.verts.shape[0], self.faces.shape[0],
                                                self.accel.data_ptr(), nbytes, L.current_stream(torch.device(device))), "warp_accel_build")
        self.use_mesh_guide = bool(use_mesh_guide)
        self.c = L.ac_warp_mesh(self.verts.data_ptr(), self.faces.data_ptr(), self.T.data_ptr(), self.verts.shape[0], self.faces.shape[0],
                                float(threshold), float(geo_threshold), int(bool(use_mesh_guide)), L.ptr(self.accel), None, 0)
        self._seeds = None

    def bind_seeds(self, seeds):
        """temporal seeds of the closest-face searches for the NEXT render call(s) with this mesh (ac_warp_mesh.seed_faces): an int32 [n_rays, >= T0 + T]
        device tensor the caller keeps across frames (new_seed_buffer), row r = ray r of the render call; None = off.  Pixels are unchanged bit for bit."""
        if seeds is None or self.accel is None:
            self._seeds = None
            self.c.seed_faces, self.c.seed_stride = None, 0
            return
        if seeds.dtype != torch.int32 or seeds.dim() != 2 or seeds.stride(1) != 1 or not seeds.is_cuda:
            raise RuntimeError("WarpMesh.bind_seeds: an int32 [n_rays, columns] device tensor with contiguous rows")
        self._seeds = seeds                                 # (kept alive)
        self.c.seed_faces, self.c.seed_stride = seeds.data_ptr(), int(seeds.stride(0))

    @staticmethod
    def new_seed_buffer(n_rays, columns, device):
        return torch.full((int(n_rays), int(columns)), -1, dtype=torch.int32, device=device)


    def work_counters(self):
        """what the closest-face searches have done on this frame's structure since it was built (ac_warp_accel_work): dict of counts"""
        if self.accel is None:
            return None
        out = (C.c_ulonglong * 4)()
        L.check(L.lib().ac_warp_accel_work(self.accel.data_ptr(), C.addressof(out), L.current_stream(self.accel.device)), "warp_accel_work")
        return dict(exact_tests=int(out[0]), disc_tests=int(out[1]), subbox_tests=int(out[2]), box_tests=int(out[3]))


_MESH_STREAMS = {}


def warp_mesh_sequence(frames, faces, device, threshold=0.05, geo_threshold=0.05, use_mesh_guide=True, overlap=True):
    """One WarpMesh per frame of an animation (render_warp.py:40-124: the pose sequence is known up front).  frames: an iterable of (verts [V,3], Ts [V,4,4]);
    faces [F,3] are uploaded and checked ONCE.  With overlap (the default on a GPU) the NEXT frame's upload and structure build (ac_warp_accel_build: five small
    launches, one of them a single-workgroup sort) are queued on a side stream before the current frame is handed out, so they run beside the current frame's
    render instead of in front of the next one's; the consumer's stream waits for a frame's mesh through an event.  Same meshes, same pixels."""
    import numpy as np
    dev = torch.device(device)
    fh = np.asarray(faces.cpu() if isinstance(faces, torch.Tensor) else faces)[:, :3]
    faces_d = torch.from_numpy(np.ascontiguousarray(fh)).to(device=dev, dtype=torch.int32).contiguous()
    fmax = int(fh.max()) if len(fh) else -1
    kw = dict(threshold=threshold, geo_threshold=geo_threshold, use_mesh_guide=use_mesh_guide)

    def make(v, T):
        if fmax >= len(T):
            raise RuntimeError("WarpMesh: Ts must hold one 4x4 per vertex")
        return WarpMesh(v, faces_d, T, dev, **kw)
    it = iter(frames)
    if not (overlap and dev.type == "cuda"):
        for v, T in it:
            yield make(v, T)
        return
    side = _MESH_STREAMS.get(str(dev))
    if side is None:
        side = _MESH_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)

    def prepare(v, T):
        with torch.cuda.stream(side):
            wm = make(v, T)
            ev = torch.cuda.Event()
            ev.record(side)
        return wm, ev
    nxt = None
    for v, T in it:
        nxt = prepare(v, T)
        break
    while nxt is not None:
        cur, ev = nxt
        nxt = None
        for v, T in it:                                     # the next frame's mesh is queued BEFORE the consumer renders this one
            nxt = prepare(v, T)
            break
        main = torch.cuda.current_stream(dev)
        main.wait_event(ev)
        for t in (cur.verts, cur.T, cur.accel):             # (allocated under the side stream, consumed on the caller's)
            if t is not None:
                t.record_stream(main)
        yield cur


_CORE_SCRATCH = {}


def core_scratch(field, N, T, dev):
    """device scratch of ac_render_core_backward, one buffer per (device, stream): grown to the largest batch seen, never shrunk
    (free_scratch() releases everything).  Keyed by stream so that two streams never share queues."""
    need = int(L.lib().ac_render_core_backward_scratch(C.byref(field.c), int(N), int(T)))
    key = (str(dev), int(L.current_stream(dev) or 0))
    cur = _CORE_SCRATCH.get(key)
    if cur is None or cur.numel() < need:
        _CORE_SCRATCH[key] = None
        cur = _CORE_SCRATCH[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    return cur, need


def free_scratch():
    """drop every cached scratch buffer of this module and of the hash encoder back end (multi-GB queues of the table-gradient scatter)"""
    _CORE_SCRATCH.clear()
    from .encoder.hashencoder import backend as BK
    BK._SCRATCH.clear()


# a training render keeps the hash features of every sample's 7-point stencil (0.9 KB per sample) so that the backward streams them back instead of
# gathering them again; AC_NO_FEAT7=1 restores the re-gathering backward (A/B timing)
SAVE_STENCIL_FEATURES = os.environ.get("AC_NO_FEAT7", "0") != "1"


# see _RenderCore.backward: False = the table gradient goes through autograd (default); True = accumulated into table.grad in place
ACCUMULATE_TABLE_GRAD_IN_PLACE = False


class _RenderCore(torch.autograd.Function):
    """NeRFRenderer.run with gradients (reference models/instant_nsr.py:133-299 under torch.enable_grad) as ONE operator:
    forward  = the fused renderer itself (ac_render_rays: sampling + render core, the launch an inference render makes) with its
               per-sample outputs kept;
    backward = ac_render_core_backward: compositing -> colour MLP -> normal normalisation + eikonal -> SDF query -> table scatter.
    Differentiable w.r.t. the hash table, the EFFECTIVE MLP matrices (weight norm stays in torch) and inv_s; the sample positions
    are constants, as in the reference (they are computed under no_grad, :176-184)."""

    @staticmethod
    def forward(ctx, table, W1, b1, W2, b2, Wc1, Wc2, Wc3, inv_s, rays_o, rays_d, bg, noise, cfg):
        offsets, pls, H, T0, up, bound, car, ner, precision, warp = cfg
        ctx.set_materialize_grads(False)              # an output the loss does not use arrives as None -> a NULL upstream pointer
        d = lambda t: t.detach().contiguous()
        Wc1_21, Wc1_sh = split_viewdir_weight(d(Wc1))           # use_viewdirs: [64,37] -> the 21 per-sample columns + the 16 view-direction columns
        field = Field(d(table), offsets, pls, H, d(W1), d(b1), d(W2), d(b2), Wc1_21, d(Wc2), d(Wc3), Wc1_sh=Wc1_sh).prepare()
        # warp = WarpMesh: posed space (run(render_can=False), instant_nsr.py:166-172,198-207,246-249) -- the launch sequence of an inference render
        # (sampling, SMPL inverse warp, final pass; every sample evaluated: skip_masked off) with the per-sample outputs kept.  The warped points,
        # the mask and the mesh-guided range are constants of the differentiation, as in the reference (the warp is numpy there).
        out = render_rays(field, rays_o, rays_d, T0, up, bound, inv_s, bg=bg, noise=noise, cos_anneal_ratio=car, normal_epsilon_ratio=ner,
                          extras=True, train_extras=True, precision=precision, warp=warp)
        ctx.field, ctx.cfg = field, cfg
        ctx.opts = out.opts                                        # the launch's ac_render_opts (+ the tensors its pointers refer to)
        ctx.posed = None
        if warp is not None:
            op_b = type(out.opts[0]).from_buffer_copy(out.opts[0])  # the backward's options: + the range the forward computed for itself
            if "near_m" in out:
                op_b.near_m, op_b.far_m = out["near_m"].data_ptr(), out["far_m"].data_ptr()
            ctx.opts = (op_b,) + tuple(out.opts[1:])
            ctx.posed = (out["mask"], out.get("near_m"), out.get("far_m"), out["_warp_scratch"], warp)
        ctx.has_bg = bg is not None
        # outputs among the saved tensors (z_vals, color) must go through save_for_backward (no reference cycle through ctx)
        ctx.save_for_backward(out["z_vals"], out["pts"], out["sdf"], out["sdf_out16"], out["gradient"], out["color"], out["eik_res"], rays_o, rays_d,
                              bg if bg is not None else rays_o, out["feat7"] if "feat7" in out else rays_o)
        ctx.has_feat = "feat7" in out
        ctx.table = table
        ctx.inv_s_shape = inv_s.shape
        ctx.mark_non_differentiable(out["weights"], out["alpha"], out["color"], out["z_vals"])
        return (out["image"], out["weights_sum"], out["depth"], out["normal_map"], out["eik_res"][0], out["weights"], out["alpha"], out["color"],
                out["z_vals"])

    @staticmethod
    def backward(ctx, g_image, g_wsum, g_depth, g_nmap, g_eik, *_unused):
        field = ctx.field
        z_vals, pts, sdf, sdf16, gradient, color, eik_res, rays_o, rays_d, bg, feat7 = ctx.saved_tensors
        if not ctx.has_bg:
            bg = None
        N, T = z_vals.shape
        dev = rays_o.device
        c = lambda g: None if g is None else g.contiguous().to(_F32)
        g_image, g_wsum, g_depth, g_nmap, g_eik = c(g_image), c(g_wsum), c(g_depth), c(g_nmap), c(g_eik)
        table = ctx.table
        # The table gradient is RETURNED to autograd like every other gradient (torch.autograd.grad, backward(inputs=...), hooks and DDP-style
        # reducers see it).  ACCUMULATE_TABLE_GRAD_IN_PLACE (opt-in, off by default) adds it straight into an existing contiguous fp32 table.grad
        # and returns None for that input instead: it saves one 49 MB zero-fill + add per backward pass for callers that only ever read .grad.
        in_place = ACCUMULATE_TABLE_GRAD_IN_PLACE and table.grad is not None and table.grad.is_contiguous() and table.grad.dtype == _F32
        g_table = table.grad if in_place else torch.zeros_like(table)       # accumulated into, like hash_encode_backward (hashgrid.py:61-68)
        g_sdf_p = torch.empty(64 * 36 + 16 * 64 + 16, dtype=_F32, device=dev)
        g_col_p = torch.empty(64 * 32 + 64 * 64 + 16 * 64, dtype=_F32, device=dev)
        g_invs = torch.empty(N, dtype=_F32, device=dev)
        sv = L.ac_core_saved(z_vals.data_ptr(), pts.data_ptr(), sdf.data_ptr(), sdf16.data_ptr(), gradient.data_ptr(), color.data_ptr(),
                             eik_res[1:].data_ptr(), feat7.data_ptr() if ctx.has_feat else None, ctx.posed[0].data_ptr() if ctx.posed else None)
        upg = L.ac_core_upstream(L.ptr(g_image), L.ptr(g_wsum), L.ptr(g_depth), L.ptr(g_nmap), L.ptr(g_eik), 0, 0)
        gr = L.ac_core_grads(g_table.data_ptr(), g_sdf_p.data_ptr(), g_col_p.data_ptr(), g_invs.data_ptr())
        scratch, need = core_scratch(field, N, T, dev)
        op = ctx.opts[0]
        vd = _viewdir_args(field, rays_d, N, T, sv, gr)
        L.check(L.lib().ac_render_core_backward(C.byref(field.c), C.byref(op), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(bg), C.byref(sv), C.byref(upg),
                                                C.byref(gr), scratch.data_ptr(), need, L.current_stream(dev)), "render_core_backward")
        gW1b = g_sdf_p[:64 * 36].view(64, 36)
        g_Wc1 = g_col_p[:2048].view(64, 32)[:, :21]
        if vd is not None:
            g_Wc1 = join_viewdir_grad(g_Wc1, _viewdir_weight_grad(vd, N, T))
        return (None if in_place else g_table, gW1b[:, :35], gW1b[:, 35], g_sdf_p[64 * 36:64 * 36 + 1024].view(16, 64), g_sdf_p[64 * 36 + 1024:],
                g_Wc1, g_col_p[2048:6144].view(64, 64), g_col_p[6144:].view(16, 64)[:3],
                g_invs.sum().reshape(ctx.inv_s_shape), None, None, None, None, None)


def render_core(table, W1, b1, W2, b2, Wc1, Wc2, Wc3, inv_s, rays_o, rays_d, bg, noise, offsets, per_level_scale, base_resolution, num_steps,
                upsample_steps, bound, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, precision="exact", warp=None):
    """-> image [N,3], weights_sum [N], depth [N], normal_map [N,3], gradient_error [], weights [N,T], alpha [N,T], color [N,T,3], z_vals [N,T]
    warp = WarpMesh(...): the posed-space render (run(render_can=False)) under autograd."""
    cfg = ([int(v) for v in offsets], per_level_scale, int(base_resolution), int(num_steps), int(upsample_steps), float(bound),
           float(cos_anneal_ratio), float(normal_epsilon_ratio), precision, warp)
    return _RenderCore.apply(table, W1, b1, W2, b2, Wc1, Wc2, Wc3, inv_s, rays_o, rays_d, bg, noise, cfg)


def weight_norm_forward(pairs, outs):
    """outs[i][r, :] = v[r, :] * g[r] / ||v[r, :]|| for every (v, g) in pairs, ONE launch (ac_weight_norm_forward); outs may be row-strided views"""
    n = len(pairs)
    arr = (L.ac_wn_layer * n)()
    for i, ((v, g), w) in enumerate(zip(pairs, outs)):
        assert v.is_contiguous() and g.is_contiguous() and v.dtype == _F32 and w.dtype == _F32 and w.stride(1) == 1 and tuple(w.shape) == tuple(v.shape)
        arr[i] = L.ac_wn_layer(v.data_ptr(), g.data_ptr(), w.data_ptr(), v.shape[0], v.shape[1], w.stride(0), 0)
    L.check(L.lib().ac_weight_norm_forward(arr, n, L.current_stream(pairs[0][0].device)), "weight_norm_forward")
    return outs


PG_WEIGHT_NORM, PG_ADD, PG_VARIANCE = 0, 1, 2


class _WeightNormAll(torch.autograd.Function):
    """torch.nn.utils.weight_norm (dim 0) of several layers as ONE operator: forward = ac_weight_norm_forward, backward = ac_param_grads -- the same
    effective matrices, bit for bit, for every path that renders a NeRFNetwork (inference, the training operators, the step without autograd)."""

    @staticmethod
    def forward(ctx, *vg):
        pairs = [(vg[2 * i].detach().contiguous(), vg[2 * i + 1].detach().contiguous()) for i in range(len(vg) // 2)]
        outs = [torch.empty_like(v) for v, _ in pairs]
        weight_norm_forward(pairs, outs)
        ctx.save_for_backward(*vg)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gw):
        vg = ctx.saved_tensors
        dev = vg[0].device
        flat = torch.zeros(sum(t.numel() for t in vg), dtype=_F32, device=dev)
        res, entries, off = [], [], 0
        for i in range(len(vg) // 2):
            v, g = vg[2 * i].detach().contiguous(), vg[2 * i + 1].detach().contiguous()
            gv = flat[off:off + v.numel()].view_as(v); off += v.numel()
            gg = flat[off:off + g.numel()].view_as(g); off += g.numel()
            if gw[i] is None:
                res += [None, None]
                continue
            gwi = gw[i].contiguous().to(_F32)
            entries.append((PG_WEIGHT_NORM, gwi, gwi.shape[1], v.shape[0], v.shape[1], v, g, gv, gg.reshape(-1)))
            res += [gv, gg]
        if entries:
            param_grads(entries, dev)
        return tuple(res)


def weight_norm_all(layers):
    """[W_l = weight_norm(l.weight_v, l.weight_g)] for the given nn.Linear-like modules, differentiable, one launch each way"""
    args = [t for l in layers for t in (l.weight_v, l.weight_g)]
    return list(_WeightNormAll.apply(*args))


def param_grads(entries, device):
    """entries: (kind, src tensor / (tensor, element offset), src_stride, rows, cols, v, g, dst, dst2); everything ACCUMULATED into dst / dst2, ONE launch"""
    n = len(entries)
    arr = (L.ac_pg_entry * n)()
    keep = []
    for i, (kind, src, stride, rows, cols, v, g, dst, dst2) in enumerate(entries):
        off = 0
        if isinstance(src, tuple):
            src, off = src
        for t in (src, v, g, dst, dst2):
            assert t is None or (t.dtype == _F32 and t.is_contiguous())
        keep.append((src, v, g, dst, dst2))
        arr[i] = L.ac_pg_entry(src.data_ptr() + 4 * off, L.ptr(v), L.ptr(g), dst.data_ptr(), L.ptr(dst2), rows, cols, stride, kind)
    L.check(L.lib().ac_param_grads(arr, n, L.current_stream(device)), "param_grads")


def sds_upstream(weights_sum, weights_sum_gt, scale, want_grad=True):
    """opacity term of the stylisation loss: -> (d loss / d weights_sum [N] or None, loss [1]); loss = sum smooth_l1(clamp, clamp) * scale"""
    ws, wg = weights_sum.reshape(-1).contiguous(), weights_sum_gt.reshape(-1).contiguous()
    N, dev = ws.shape[0], ws.device
    g = torch.empty(N, dtype=_F32, device=dev) if want_grad else None
    loss = torch.empty(1, dtype=_F32, device=dev)
    L.check(L.lib().ac_sds_upstream(ws.data_ptr(), wg.data_ptr(), N, float(scale), L.ptr(g), loss.data_ptr(), L.current_stream(dev)), "sds_upstream")
    return g, loss


VIEWDIR_COLS = 16                      # degree-4 spherical harmonics of the ray direction: columns 3 .. 18 of a use_viewdirs color_net.0 weight [64, 37]


def split_viewdir_weight(Wc1):
    """the effective color_net.0 weight of NeRFNetwork(use_viewdirs=True), [64, 37] = [x(3) | sh(d)(16) | normal(3) | geo_feat(15)] (models/instant_nsr.py:648-650)
    -> (Wc1 [64,21] = [x | normal | geo_feat] as the fused kernels keep it, Wc1_sh [64,16]); a [64,21] weight -> (Wc1, None)"""
    if Wc1.shape[1] == 21:
        return Wc1, None
    if Wc1.shape[1] != 21 + VIEWDIR_COLS:
        raise RuntimeError(f"colour layer 1 has {Wc1.shape[1]} inputs: the fused renderer takes 21 (no view directions) or 37 (degree-4 spherical harmonics)")
    return torch.cat([Wc1[:, :3], Wc1[:, 3 + VIEWDIR_COLS:]], dim=1).contiguous(), Wc1[:, 3:3 + VIEWDIR_COLS].contiguous()


def join_viewdir_grad(g21, g_sh):
    """inverse of split_viewdir_weight for gradients: ([64,21], [64,16]) -> [64,37]"""
    return torch.cat([g21[:, :3], g_sh, g21[:, 3:]], dim=1)


def sh_bias(field, rays_d, want_sh=True):
    """ac_sh_bias: the per-ray layer-1 bias of the colour network for the view directions rays_d [N,3] -> (bias [N,64], sh [N,16] or None)"""
    rays_d = _chk(rays_d.reshape(-1, 3), "rays_d")
    N, dev = rays_d.shape[0], rays_d.device
    bias = torch.empty((N, 64), dtype=_F32, device=dev)
    sh = torch.empty((N, VIEWDIR_COLS), dtype=_F32, device=dev) if want_sh else None
    L.check(L.lib().ac_sh_bias(C.byref(field.c), rays_d.data_ptr(), N, bias.data_ptr(), L.ptr(sh), L.current_stream(dev)), "sh_bias")
    return bias, sh


def _viewdir_args(field, rays_d, N, T, sv, gr):
    """a field with view directions: fill ac_core_saved.sh_bias / ac_core_grads.g_sh_tiles; returns what viewdir_weight_grad needs (None otherwise)"""
    if not getattr(field, "has_viewdirs", False):
        return None
    bias, sh = sh_bias(field, rays_d)
    tiles = torch.empty((N * T // 16, 64), dtype=_F32, device=rays_d.device)
    sv.sh_bias, gr.g_sh_tiles = bias.data_ptr(), tiles.data_ptr()
    return bias, sh, tiles


def _viewdir_weight_grad(vd, N, T):
    """d Wc1_sh [64,16] = sum over rays of (the ray's T / 16 tile rows of g_sh_tiles, summed) (x) sh(d_ray)"""
    if vd is None:
        return None
    _, sh, tiles = vd
    return tiles.view(N, T // 16, 64).sum(1).t().matmul(sh)


def eikonal_groups(eik, group_rays):
    """gradient_error (instant_nsr.py:266-272) per group of `group_rays` consecutive rays of a launch, from its per-ray partial sums eik [N,2]
    (ac_render_out.eik): [groups, 2] = (error, denominator), each by ac_eikonal_reduce2 -- the bits a launch of that group alone reports as eik_res"""
    N = eik.shape[0]
    G = (N + group_rays - 1) // group_rays
    res = torch.empty((G, 2), dtype=_F32, device=eik.device)
    st = L.current_stream(eik.device)
    for k in range(G):
        sl = eik[k * group_rays:(k + 1) * group_rays]
        L.check(L.lib().ac_eikonal_reduce2(sl.data_ptr(), sl.shape[0], res[k].data_ptr(), st), "eikonal_reduce2")
    return res


def render_core_backward(field, opts, out, rays_o, rays_d, bg, g_image, g_wsum, g_depth, g_nmap, g_eik, g_table, split=None, eik_groups=None):
    """ac_render_core_backward on the outputs of a render_rays(..., train_extras=True) launch (`out`, its .opts): the table gradient is accumulated
    into g_table; returns (g_sdf_params [3344], g_color_params [7168], g_inv_s_per_ray [N][, g_Wc1_sh [64,16] for a field with view directions]) w.r.t. the
    EFFECTIVE matrices.
    split = (level, torch.cuda.Stream): the scatter completes the table gradient of levels >= level first and orders that stream behind exactly that
    (ac_core_grads.side_stream / split_level): work enqueued there afterwards (the all-reduce of that slice) overlaps the rest of the backward."""
    z_vals = out["z_vals"]
    N, T = z_vals.shape
    dev = rays_o.device
    c = lambda g: None if g is None else g.contiguous().to(_F32)
    g_image, g_wsum, g_depth, g_nmap, g_eik = c(g_image), c(g_wsum), c(g_depth), c(g_nmap), c(g_eik)
    g_sdf_p = torch.empty(64 * 36 + 16 * 64 + 16, dtype=_F32, device=dev)
    g_col_p = torch.empty(64 * 32 + 64 * 64 + 16 * 64, dtype=_F32, device=dev)
    g_invs = torch.empty(N, dtype=_F32, device=dev)
    # eik_groups = (group_rays, res [G,2] from eikonal_groups): the launch holds G patches whose eikonal terms are separate ratios; g_eik is then [G]
    den_ptr = out["eik_res"][1:].data_ptr() if eik_groups is None else eik_groups[1][:, 1:].data_ptr()
    sv = L.ac_core_saved(z_vals.data_ptr(), out["pts"].data_ptr(), out["sdf"].data_ptr(), out["sdf_out16"].data_ptr(), out["gradient"].data_ptr(),
                         out["color"].data_ptr(), den_ptr, L.ptr(out.get("feat7") if hasattr(out, "get") else None))
    upg = L.ac_core_upstream(L.ptr(g_image), L.ptr(g_wsum), L.ptr(g_depth), L.ptr(g_nmap), L.ptr(g_eik), 0, 0)
    if eik_groups is not None:
        if g_eik is not None and g_eik.numel() != eik_groups[1].shape[0]:
            raise RuntimeError("render_core_backward: one g_eik per group of rays")
        upg.eik_group_rays, upg.eik_den_stride = int(eik_groups[0]), 2
    gr = L.ac_core_grads(g_table.data_ptr(), g_sdf_p.data_ptr(), g_col_p.data_ptr(), g_invs.data_ptr())
    if split is not None:
        gr.split_level, gr.side_stream = int(split[0]), int(split[1].cuda_stream)
    scratch, need = core_scratch(field, N, T, dev)
    vd = _viewdir_args(field, rays_d, N, T, sv, gr)
    L.check(L.lib().ac_render_core_backward(C.byref(field.c), C.byref(opts[0]), rays_o.data_ptr(), rays_d.data_ptr(), L.ptr(bg), C.byref(sv), C.byref(upg),
                                            C.byref(gr), scratch.data_ptr(), need, L.current_stream(dev)), "render_core_backward")
    if vd is not None:
        return g_sdf_p, g_col_p, g_invs, _viewdir_weight_grad(vd, N, T)            # (+ d Wc1_sh [64,16] for a field with view directions)
    return g_sdf_p, g_col_p, g_invs


class _PackedShading(torch.autograd.Function):
    """normal, cos-annealed NeuS alpha and the two per-sample eikonal terms of PACKED samples (ac_packed_shading_forward / _backward): the glue of
    NeRFRenderer.run_cuda's train() branch between the fused SDF query and the packed compositor, one launch each way.
    (sdf_out16 [M,16], gradient [M,3], inv_s) carry gradients; xyzs, dirs, deltas, n_valid (device int32) do not."""

    @staticmethod
    def forward(ctx, sdf16, gradient, inv_s, xyzs, dirs, deltas, n_valid, car):
        sdf16, gradient, xyzs, dirs, deltas = (t.contiguous() for t in (sdf16, gradient, xyzs, dirs, deltas))
        M, dev = sdf16.shape[0], sdf16.device
        stride = 1 if deltas.dim() == 1 else int(deltas.shape[1])
        inv_t = inv_s.detach().reshape(1).to(_F32).contiguous()
        nv = n_valid.detach().reshape(1).to(torch.int32).contiguous()
        alpha = torch.empty(M, dtype=_F32, device=dev)
        normal = torch.empty((M, 3), dtype=_F32, device=dev)
        eik = torch.empty((M, 2), dtype=_F32, device=dev)
        L.check(L.lib().ac_packed_shading_forward(sdf16.data_ptr(), gradient.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), stride, M, nv.data_ptr(),
                                                  0.0, inv_t.data_ptr(), float(car), alpha.data_ptr(), normal.data_ptr(), eik.data_ptr(), L.current_stream(dev)),
                "packed_shading_forward")
        ctx.save_for_backward(sdf16, gradient, xyzs, dirs, deltas, inv_t, nv)
        ctx.meta = (stride, float(car), inv_s.shape)
        return alpha, normal, eik

    @staticmethod
    def backward(ctx, g_alpha, g_normal, g_eik):
        sdf16, gradient, xyzs, dirs, deltas, inv_t, nv = ctx.saved_tensors
        stride, car, inv_shape = ctx.meta
        M, dev = sdf16.shape[0], sdf16.device
        c = lambda g: None if g is None else g.contiguous().to(_F32)
        g_alpha, g_normal, g_eik = c(g_alpha), c(g_normal), c(g_eik)
        g_sdf16 = torch.zeros((M, 16), dtype=_F32, device=dev)
        g_grad = torch.empty((M, 3), dtype=_F32, device=dev)
        g_rows = torch.empty(M, dtype=_F32, device=dev)
        L.check(L.lib().ac_packed_shading_backward(sdf16.data_ptr(), gradient.data_ptr(), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), stride, M, nv.data_ptr(),
                                                   0.0, inv_t.data_ptr(), car, L.ptr(g_alpha), L.ptr(g_normal), L.ptr(g_eik), g_sdf16.data_ptr(), g_grad.data_ptr(),
                                                   g_rows.data_ptr(), L.current_stream(dev)), "packed_shading_backward")
        return g_sdf16, g_grad, g_rows.sum().reshape(inv_shape), None, None, None, None, None


def packed_shading(sdf16, gradient, inv_s, xyzs, dirs, deltas, n_valid, cos_anneal_ratio=1.0):
    """-> (alpha [M], normal [M,3], eik [M,2] = (relax (|g| - 1)^2, relax)); see _PackedShading"""
    return _PackedShading.apply(sdf16, gradient, inv_s, xyzs, dirs, deltas, n_valid, float(cos_anneal_ratio))


class _SdfStencil(torch.autograd.Function):
    """forward_sdf(x) + finite_difference_normals_approximator(x) of the render core as one fused op with a fused backward
    (csrc/sdf_train.hip).  Inputs: x [B,3], the hash table, the EFFECTIVE sdf_net matrices (weight norm stays in torch).
    x may require grad (the curvature term's perturbed points, models/instant_nsr.py:276-288: positions that are a function of the normal): the backward then
    also returns d loss / d x = the share through the MLP's own xyz inputs (ac_sdf_stencil_backward_inputs) + the share through the encodings (the
    reference's dy_dx path, hashencoder.cu:177-220,311-337: ac_hash_stencil_input_backward)."""

    @staticmethod
    def forward(ctx, x, table, W1, b1, W2, b2, cfg):
        offsets, pls, H, bound, eps = cfg
        x = x.contiguous()
        dev = x.device
        dummy = lambda *shape: torch.zeros(shape, dtype=_F32, device=dev)
        field = Field(table.detach().contiguous(), offsets, pls, H, W1.detach().contiguous(), b1.detach().contiguous(), W2.detach().contiguous(),
                      b2.detach().contiguous(), dummy(64, 21), dummy(64, 64), dummy(3, 64))
        B = x.shape[0]
        out16 = torch.empty((B, 16), dtype=_F32, device=dev)
        grad = torch.empty((B, 3), dtype=_F32, device=dev)
        L.check(L.lib().ac_sdf_stencil_forward(C.byref(field.c), x.data_ptr(), B, float(bound), float(eps), out16.data_ptr(), grad.data_ptr(),
                                               L.current_stream(dev)), "sdf_stencil_forward")
        ctx.save_for_backward(x, table, W1, b1, W2, b2)
        ctx.cfg = cfg
        return out16, grad

    @staticmethod
    def backward(ctx, g_out, g_grad):
        import numpy as np
        x, table, W1, b1, W2, b2 = ctx.saved_tensors
        offsets, pls, H, bound, eps = ctx.cfg
        dev = x.device
        dummy = lambda *shape: torch.zeros(shape, dtype=_F32, device=dev)
        field = Field(table.detach().contiguous(), offsets, pls, H, W1.detach().contiguous(), b1.detach().contiguous(), W2.detach().contiguous(),
                      b2.detach().contiguous(), dummy(64, 21), dummy(64, 64), dummy(3, 64))
        B = x.shape[0]
        g_out = g_out.contiguous().float(); g_grad = g_grad.contiguous().float()
        gfeat = torch.empty((7, 16, B, 2), dtype=_F32, device=dev)
        gparams = torch.empty(64 * 36 + 16 * 64 + 16, dtype=_F32, device=dev)
        nbytes = int(L.lib().ac_sdf_stencil_backward_scratch(B))
        scratch = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        st = L.current_stream(dev)
        oh = np.asarray(offsets, dtype=np.int32)
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty((B, 3), dtype=_F32, device=dev)
            L.check(L.lib().ac_sdf_stencil_backward_inputs(C.byref(field.c), x.data_ptr(), g_out.data_ptr(), g_grad.data_ptr(), B, float(bound), float(eps),
                                                           gfeat.data_ptr(), gparams.data_ptr(), g_x.data_ptr(), scratch.data_ptr(), nbytes, st),
                    "sdf_stencil_backward_inputs")
            part = torch.empty(((16 + 3) // 4, B, 3), dtype=_F32, device=dev)
            L.check(L.lib().ac_hash_stencil_input_backward(gfeat.data_ptr(), x.data_ptr(), field.t["table"].data_ptr(), oh.ctypes.data, part.data_ptr(), B, 2, 16,
                                                           field.S, H, float(eps), float(bound), st), "hash_stencil_input_backward")
            g_x = g_x + part.sum(0)
        else:
            L.check(L.lib().ac_sdf_stencil_backward(C.byref(field.c), x.data_ptr(), g_out.data_ptr(), g_grad.data_ptr(), B, float(bound), float(eps),
                                                    gfeat.data_ptr(), gparams.data_ptr(), scratch.data_ptr(), nbytes, st), "sdf_stencil_backward")
        g_table = torch.zeros_like(table)
        from .encoder.hashencoder.backend import stencil_scratch
        hs, hbytes = stencil_scratch(oh, 16, field.S, H, dev, B)
        L.check(L.lib().ac_hash_stencil_backward(gfeat.data_ptr(), x.data_ptr(), oh.ctypes.data, g_table.data_ptr(), B, 2, 16, field.S, H, float(eps),
                                                 float(bound), L.ptr(hs), hbytes, st), "hash_stencil_backward")
        gW1b = gparams[:64 * 36].view(64, 36)
        return (g_x, g_table, gW1b[:, :35].contiguous(), gW1b[:, 35].contiguous(), gparams[64 * 36:64 * 36 + 1024].view(16, 64),
                gparams[64 * 36 + 1024:], None)


class _ColorMlp(torch.autograd.Function):
    """forward_color of the render core (use_viewdirs = False) as a fused op with a fused backward (csrc/sdf_train.hip).
    Inputs: x [B,3] (no grad), normal [B,3], sdf_out [B,16] (column 0 unused), the EFFECTIVE colour matrices."""

    @staticmethod
    def _field(dev, Wc1, Wc2, Wc3):
        z = lambda *shape: torch.zeros(shape, dtype=_F32, device=dev)
        offs = list(range(0, 17))                  # a dummy 16-level layout of one entry per level: the colour kernels never touch the table
        return Field(z(16, 2), offs, 2.0, 16, z(64, 35), z(64), z(16, 64), z(16), Wc1.detach().contiguous(), Wc2.detach().contiguous(),
                     Wc3.detach().contiguous())

    @staticmethod
    def forward(ctx, x, normal, sdf_out, Wc1, Wc2, Wc3):
        x, normal, sdf_out = x.contiguous(), normal.contiguous(), sdf_out.contiguous()
        B, dev = x.shape[0], x.device
        field = _ColorMlp._field(dev, Wc1, Wc2, Wc3)
        rgb = torch.empty((B, 3), dtype=_F32, device=dev)
        L.check(L.lib().ac_color_forward(C.byref(field.c), x.data_ptr(), normal.data_ptr(), sdf_out.data_ptr(), B, rgb.data_ptr(), L.current_stream(dev)),
                "color_forward")
        ctx.save_for_backward(x, normal, sdf_out, Wc1, Wc2, Wc3)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        x, normal, sdf_out, Wc1, Wc2, Wc3 = ctx.saved_tensors
        B, dev = x.shape[0], x.device
        field = _ColorMlp._field(dev, Wc1, Wc2, Wc3)
        g_rgb = g_rgb.contiguous().float()
        g_n = torch.empty((B, 3), dtype=_F32, device=dev)
        g_s = torch.empty((B, 16), dtype=_F32, device=dev)
        gp = torch.empty(64 * 32 + 64 * 64 + 16 * 64, dtype=_F32, device=dev)
        nbytes = int(L.lib().ac_color_backward_scratch(B))
        scratch = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        L.check(L.lib().ac_color_backward(C.byref(field.c), x.data_ptr(), normal.data_ptr(), sdf_out.data_ptr(), g_rgb.data_ptr(), B, g_n.data_ptr(),
                                          g_s.data_ptr(), gp.data_ptr(), scratch.data_ptr(), nbytes, L.current_stream(dev)), "color_backward")
        return (None, g_n, g_s, gp[:2048].view(64, 32)[:, :21].contiguous(), gp[2048:6144].view(64, 64), gp[6144:].view(16, 64)[:3].contiguous())


def color_mlp(x, normal, sdf_out, Wc1, Wc2, Wc3):
    """-> rgb [B,3]; differentiable w.r.t. normal, sdf_out[:, 1:], Wc1, Wc2, Wc3"""
    return _ColorMlp.apply(x, normal, sdf_out, Wc1, Wc2, Wc3)


class _Composite(torch.autograd.Function):
    """NeuS alpha + compositing of the render core as one fused op each way (csrc/sdf_train.hip)"""

    @staticmethod
    def forward(ctx, z_vals, sdf, normal, color, inv_s, rays_o, rays_d, bg, num_steps, bound, car):
        N, T = z_vals.shape
        dev = z_vals.device
        z_vals, sdf, normal, color = z_vals.contiguous(), sdf.contiguous(), normal.contiguous(), color.contiguous()
        f = lambda *s: torch.empty(s, dtype=_F32, device=dev)
        image, wsum, depth, nmap, weights, alpha = f(N, 3), f(N), f(N), f(N, 3), f(N, T), f(N, T)
        s = float(inv_s.detach().reshape(-1)[0])
        L.check(L.lib().ac_composite_forward(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), sdf.data_ptr(), normal.data_ptr(), color.data_ptr(),
                                             L.ptr(bg), N, int(num_steps), T, float(bound), s, float(car), image.data_ptr(), wsum.data_ptr(),
                                             depth.data_ptr(), nmap.data_ptr(), weights.data_ptr(), alpha.data_ptr(), L.current_stream(dev)),
                "composite_forward")
        ctx.save_for_backward(z_vals, sdf, normal, color, inv_s, rays_o, rays_d, bg if bg is not None else torch.empty(0, device=dev))
        ctx.cfg = (int(num_steps), float(bound), float(car), s, bg is not None)
        ctx.mark_non_differentiable(weights, alpha)
        return image, wsum, depth, nmap, weights, alpha

    @staticmethod
    def backward(ctx, g_image, g_wsum, g_depth, g_nmap, _gw, _ga):
        z_vals, sdf, normal, color, inv_s, rays_o, rays_d, bg = ctx.saved_tensors
        num_steps, bound, car, s, has_bg = ctx.cfg
        N, T = z_vals.shape
        dev = z_vals.device
        c = lambda g, *shape: (g.contiguous().float() if g is not None else torch.zeros(shape, dtype=_F32, device=dev))
        g_image, g_wsum, g_depth, g_nmap = c(g_image, N, 3), c(g_wsum, N), c(g_depth, N), c(g_nmap, N, 3)
        f = lambda *sh: torch.empty(sh, dtype=_F32, device=dev)
        g_sdf, g_nrm, g_col, g_s = f(N, T), f(N, T, 3), f(N, T, 3), f(N)
        L.check(L.lib().ac_composite_backward(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), sdf.data_ptr(), normal.data_ptr(), color.data_ptr(),
                                              bg.data_ptr() if has_bg else None, N, num_steps, T, bound, s, car, g_image.data_ptr(), g_wsum.data_ptr(),
                                              g_depth.data_ptr(), g_nmap.data_ptr(), g_sdf.data_ptr(), g_nrm.data_ptr(), g_col.data_ptr(), g_s.data_ptr(),
                                              L.current_stream(dev)), "composite_backward")
        return None, g_sdf, g_nrm, g_col, g_s.sum().reshape(inv_s.shape), None, None, None, None, None, None


def composite(z_vals, sdf, normal, color, inv_s, rays_o, rays_d, bg, num_steps, bound, cos_anneal_ratio):
    """-> image [N,3], weights_sum [N], depth [N], normal_map [N,3], weights [N,T], alpha [N,T]; differentiable w.r.t. sdf, normal, color, inv_s"""
    return _Composite.apply(z_vals, sdf, normal, color, inv_s, rays_o, rays_d, bg, num_steps, bound, cos_anneal_ratio)


def sdf_stencil(x, table, W1, b1, W2, b2, offsets, per_level_scale, base_resolution, bound, eps):
    """-> (sdf_out [B,16], gradient [B,3]); differentiable w.r.t. table, W1, b1, W2, b2"""
    cfg = ([int(v) for v in offsets], per_level_scale, int(base_resolution), float(bound), float(eps))
    return _SdfStencil.apply(x, table, W1, b1, W2, b2, cfg)


def variance_forward(variance):
    """forward_variance() without a graph: clip(exp(10 * variance), 1e-6, 1e6) as a [1, 1] tensor, one launch (ac_variance_forward)"""
    v = variance.detach()
    _chk(v, "variance")
    if v.dtype != torch.float32 or v.numel() != 1:
        raise RuntimeError("variance_forward: a single float32 value")
    out = torch.empty((1, 1), dtype=torch.float32, device=v.device)
    L.check(L.lib().ac_variance_forward(v.data_ptr(), out.data_ptr(), L.current_stream(v.device)), "variance_forward")
    return out


def field_samples(field, xyzs, dirs, deltas, bound, eps, inv_s, cos_anneal_ratio=1.0, want_sdf=False, want_gradient=False):
    """ac_field_samples: the field on packed samples (what run_cuda evaluates between the marcher and the packed compositor).
    xyzs, dirs [M,3]; deltas [M] (march_rays_train) or [M,2] (march_rays; column 0 is the step).  inv_s: float or a CUDA tensor (read on the device).
    -> dict(alpha [M], rgb [M,3], normal [M,3] (+ sdf [M], gradient [M,3]))"""
    xyzs = _chk(xyzs.reshape(-1, 3), "xyzs"); dirs = _chk(dirs.reshape(-1, 3), "dirs"); deltas = _chk(deltas, "deltas")
    M, dev = xyzs.shape[0], xyzs.device
    if dirs.shape[0] != M or deltas.shape[0] != M or deltas.dim() not in (1, 2):
        raise RuntimeError("field_samples: xyzs [M,3], dirs [M,3], deltas [M] or [M,k]")
    stride = 1 if deltas.dim() == 1 else int(deltas.shape[1])
    f = lambda *sh: torch.empty(sh, dtype=_F32, device=dev)
    out = dict(alpha=f(M), rgb=f(M, 3), normal=f(M, 3))
    if want_sdf:
        out["sdf"] = f(M)
    if want_gradient:
        out["gradient"] = f(M, 3)
    inv_f, inv_t = _inv_s_arg(inv_s)
    L.check(L.lib().ac_field_samples(C.byref(field.c), xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), stride, M, float(bound), float(eps), inv_f, L.ptr(inv_t),
                                     float(cos_anneal_ratio), out["alpha"].data_ptr(), out["rgb"].data_ptr(), out["normal"].data_ptr(),
                                     L.ptr(out.get("sdf")), L.ptr(out.get("gradient")), L.current_stream(dev)), "field_samples")
    return out


# from this ray count on the inference launch runs in phases (same bits): 1.01 against 1.94 ms on a 256 x 256 view, 0.54 against 0.87 ms on a 4096-ray batch of
# its middle rows; a few dozen rays are quicker through the one-wave-per-group kernel (no grid barriers)
OCCUPANCY_PHASED_MIN_RAYS = int(os.environ.get("AC_OCC_PHASED_MIN_RAYS", "2048"))
_OP_SCRATCH = {}

# The phased kernels (inference and training form) synchronise their grid with barriers in global memory.  The launch is sized and checked for co-residency
# (csrc/ac_common.hpp: launch_resident); what is left is a FOREIGN kernel holding compute units for longer than a barrier's bounded spin -- then the launch
# gives up and counts itself in its scratch's sticky word.  Neither form hands out the partial result (the reference's loop, raymarching/raymarching.py:136-188,
# cannot produce one):
#   inference -- on the device: ac_render_rays_occupancy_phased queues the barrier-free kernel behind the phased one, conditional on that launch's verdict
#       word (a few microseconds when it is not needed, no host round trip); it renders every ray again, the same bits;
#   training form -- OCCUPANCY_VERIFY (default on; AC_OCC_VERIFY=0 for callers that poll occupancy_launch_failures() themselves): the wrapper reads the
#       sticky word behind the launch (one 4-byte read-back, +0.08 ms on a 0.34 ms launch) and raises OccupancyBarrierTimeout, which instant_nsr.run_cuda
#       answers with the chain of operators (the same pixels).
OCCUPANCY_VERIFY = os.environ.get("AC_OCC_VERIFY", "1") != "0"
_OCC_FAILURES = {"dropped": 0, "fallbacks": 0}        # sticky words of scratch buffers that were dropped or replaced | launches answered by the fallback


class OccupancyBarrierTimeout(RuntimeError):
    """a grid barrier of a phased occupancy launch timed out (compute units held by a foreign kernel): its outputs are partial and were NOT returned"""


def _sticky_word(sc):
    return int(sc[32:36].view(torch.int32).item())


def _retire_scratch(cache, key):
    """drop a scratch buffer, keeping its failure count in the process total (clearing or re-allocating a buffer lost its sticky word)"""
    sc = cache.pop(key, None)
    if sc is not None:
        _OCC_FAILURES["dropped"] += _sticky_word(sc)


def render_rays_occupancy(field, rays_o, rays_d, density_grid, mean_density, bound, eps, inv_s, cos_anneal_ratio=1.0, count_samples=False, max_steps=0,
                          phased=None):
    """ac_render_rays_occupancy / ac_render_rays_occupancy_phased: the inference form of run_cuda in one launch (no host round trips).
    -> dict(weights_sum [N], depth [N] (raw sum of w t), image [N,3] (no background), normal_map [N,3] (+ n_samples, a [1] int32 device tensor))
    max_steps: a ray stops after that many samples (0 = no cap); see include/avatarcraft_hip.h for how that relates to the loop of rounds.
    phased: None = by ray count (OCCUPANCY_PHASED_MIN_RAYS); the two kernels give the same bits."""
    rays_o = _chk(rays_o.reshape(-1, 3), "rays_o"); rays_d = _chk(rays_d.reshape(-1, 3), "rays_d"); grid = _chk(density_grid, "density_grid")
    N, dev = rays_o.shape[0], rays_o.device
    if grid.dim() != 3 or grid.shape[0] != grid.shape[1] or grid.shape[0] != grid.shape[2]:
        raise RuntimeError("render_rays_occupancy: density_grid must be [H, H, H]")
    f = lambda *sh: torch.empty(sh, dtype=_F32, device=dev)
    out = dict(weights_sum=f(N), depth=f(N), image=f(N, 3), normal_map=f(N, 3))
    if count_samples:
        out["n_samples"] = torch.zeros(1, dtype=torch.int32, device=dev)
    inv_f, inv_t = _inv_s_arg(inv_s)
    if phased is None:
        phased = N >= OCCUPANCY_PHASED_MIN_RAYS
    if phased and N > 0:
        need = int(L.lib().ac_render_rays_occupancy_phased_scratch(N))
        key = (str(dev), int(L.current_stream(dev) or 0))
        sc = _OP_SCRATCH.get(key)
        if sc is None or sc.numel() < need:
            _retire_scratch(_OP_SCRATCH, key)
            sc = _OP_SCRATCH[key] = torch.zeros(need, dtype=torch.uint8, device=dev)          # zeroed once: every launch re-arms its sync words
            sc._ac_seen_failures = 0
        L.check(L.lib().ac_render_rays_occupancy_phased(C.byref(field.c), rays_o.data_ptr(), rays_d.data_ptr(), N, grid.data_ptr(), int(grid.shape[0]),
                                                        float(mean_density), float(bound), float(eps), inv_f, L.ptr(inv_t), float(cos_anneal_ratio),
                                                        out["weights_sum"].data_ptr(), out["depth"].data_ptr(), out["image"].data_ptr(),
                                                        out["normal_map"].data_ptr(), L.ptr(out.get("n_samples")), max(0, int(max_steps)), sc.data_ptr(),
                                                        sc.numel(), L.current_stream(dev)), "render_rays_occupancy_phased")
        # (a launch whose grid barrier timed out is answered ON THE DEVICE: the library queues the barrier-free kernel behind the phased one, conditional on that
        #  launch's verdict word -- no read-back here, never a partial result; occupancy_launch_failures() counts such launches)
        return out
    L.check(L.lib().ac_render_rays_occupancy(C.byref(field.c), rays_o.data_ptr(), rays_d.data_ptr(), N, grid.data_ptr(), int(grid.shape[0]), float(mean_density),
                                             float(bound), float(eps), inv_f, L.ptr(inv_t), float(cos_anneal_ratio), out["weights_sum"].data_ptr(),
                                             out["depth"].data_ptr(), out["image"].data_ptr(), out["normal_map"].data_ptr(), L.ptr(out.get("n_samples")),
                                             max(0, int(max_steps)), L.current_stream(dev)), "render_rays_occupancy")
    return out


_OT_SCRATCH = {}


def render_rays_occupancy_train(field, rays_o, rays_d, density_grid, mean_density, bound, eps, inv_s, cos_anneal_ratio=1.0, perturb=False, capacity=0,
                                composite_capacity=0, counter=None, bg=None):
    """ac_render_rays_occupancy_train: the training form of run_cuda without autograd in one launch (walk + record, offsets + packed samples, field on
    tiles dealt to all waves, the packed compositor per ray; grid barriers between the phases).  capacity / composite_capacity: the M of march_rays_train /
    composite_rays_train, both > 0 (a budgeted call: the packed layout lives in the launch's scratch); counter: the [2] int32 step counter (+= samples,
    += N); bg: None (raw sums), a number, a [3] / [1,3] or an [N,3] tensor -- image + (1 - weights_sum) * bg.
    -> dict(weights_sum [N], image [N,3], normal_map [N,3], gradient_error [1])"""
    rays_o = _chk(rays_o.reshape(-1, 3), "rays_o"); rays_d = _chk(rays_d.reshape(-1, 3), "rays_d"); grid = _chk(density_grid, "density_grid")
    N, dev = rays_o.shape[0], rays_o.device
    if grid.dim() != 3 or grid.shape[0] != grid.shape[1] or grid.shape[0] != grid.shape[2]:
        raise RuntimeError("render_rays_occupancy_train: density_grid must be [H, H, H]")
    if counter is not None and (counter.dtype != torch.int32 or counter.numel() < 2 or not counter.is_cuda or not counter.is_contiguous()):
        raise RuntimeError("render_rays_occupancy_train: counter must be a contiguous [2] int32 device tensor")
    cap, ccap = int(capacity), int(composite_capacity) or int(capacity)
    if cap <= 0:
        raise RuntimeError("render_rays_occupancy_train: capacity must be > 0 (the un-budgeted form of march_rays_train sizes its layout after counting: use the operators)")
    bg_mode, bg_value, bg_t = 0, 0.0, None
    if bg is not None:
        if isinstance(bg, torch.Tensor):
            if bg.numel() == 1 and not bg.is_cuda:
                bg_mode, bg_value = 1, float(bg.reshape(()).item())
            elif bg.numel() == 1:                                  # a device scalar: as a broadcast triple, without a read-back
                bg_mode, bg_t = 2, _chk(bg.reshape(1).expand(3).contiguous(), "bg")
            elif bg.numel() == 3:
                bg_mode, bg_t = 2, _chk(bg.reshape(3), "bg")
            elif bg.numel() == 3 * N:
                bg_mode, bg_t = 3, _chk(bg.reshape(N, 3), "bg")
            else:
                raise RuntimeError(f"render_rays_occupancy_train: background of {tuple(bg.shape)} for {N} rays")
        else:
            bg_mode, bg_value = 1, float(bg)
    f = lambda *sh: torch.empty(sh, dtype=_F32, device=dev)
    out = dict(weights_sum=f(N), image=f(N, 3), normal_map=f(N, 3), gradient_error=f(1))
    need = int(L.lib().ac_render_rays_occupancy_train_scratch(N, cap))
    key = (str(dev), int(L.current_stream(dev) or 0), N, cap)      # (the layout depends on N and the capacity: a buffer is re-armed for calls of ITS shape)
    sc = _OT_SCRATCH.get(key)
    if sc is None:
        if len(_OT_SCRATCH) > 8:
            for k in list(_OT_SCRATCH):
                _retire_scratch(_OT_SCRATCH, k)
        sc = _OT_SCRATCH[key] = torch.zeros(need, dtype=torch.uint8, device=dev)      # zeroed once: every launch re-arms it
        sc._ac_seen_failures = 0
    inv_f, inv_t = _inv_s_arg(inv_s)
    L.check(L.lib().ac_render_rays_occupancy_train(C.byref(field.c), rays_o.data_ptr(), rays_d.data_ptr(), N, grid.data_ptr(), int(grid.shape[0]),
                                                   float(mean_density), float(bound), float(eps), inv_f, L.ptr(inv_t), float(cos_anneal_ratio),
                                                   1 if perturb else 0, cap, ccap, L.ptr(counter), L.ptr(bg_t), bg_mode, float(bg_value),
                                                   out["weights_sum"].data_ptr(), out["image"].data_ptr(), out["normal_map"].data_ptr(),
                                                   out["gradient_error"].data_ptr(), sc.data_ptr(), sc.numel(), L.current_stream(dev)),
            "render_rays_occupancy_train")
    if OCCUPANCY_VERIFY and N > 0 and _sticky_word(sc) != getattr(sc, "_ac_seen_failures", 0):
        _retire_scratch(_OT_SCRATCH, key)                 # (phases after the failed barrier were skipped: the buffer is not re-armed)
        _OCC_FAILURES["fallbacks"] += 1
        raise OccupancyBarrierTimeout("render_rays_occupancy_train: a grid barrier timed out (compute units held by a foreign kernel); no outputs were returned -- "
                                      "render through the operators (instant_nsr.run_cuda does)")
    return out


def occupancy_launch_failures():
    """launches of the phased occupancy kernels (inference and training form) whose grid barrier timed out, over the life of this process: the sticky word 8
    of every scratch buffer held now + the words of the buffers dropped or replaced since (kept in _OCC_FAILURES).  0 on a healthy run.  With
    OCCUPANCY_VERIFY (default) every one of them was answered by the barrier-free path and no partial result left the wrappers; without it such a launch leaves
    NaN in gradient_error / weights_sum[0].  Synchronises."""
    n = _OCC_FAILURES["dropped"]
    for sc in list(_OT_SCRATCH.values()) + list(_OP_SCRATCH.values()):
        n += _sticky_word(sc)
    return n


def occupancy_fallbacks():
    """phased launches whose outputs were withheld and rendered again / raised (OCCUPANCY_VERIFY)"""
    return _OCC_FAILURES["fallbacks"]


def debug_hold_cus(blocks, lds_bytes, millis, stream=None):
    """test utility: ac_debug_hold_cus on `stream` (a torch.cuda.Stream; default: the current one)"""
    st = stream.cuda_stream if stream is not None else L.current_stream(None)
    L.check(L.lib().ac_debug_hold_cus(int(blocks), int(lds_bytes), int(millis), st), "debug_hold_cus")


def field_sdf(field, x, bound):
    """forward_sdf (instant_nsr.py:627-642): x [B,3] -> [B,16] (sdf, 15 features)"""
    x = _chk(x.reshape(-1, 3), "x")
    out = torch.empty((x.shape[0], 16), dtype=_F32, device=x.device)
    L.check(L.lib().ac_field_sdf(C.byref(field.c), x.data_ptr(), x.shape[0], float(bound), out.data_ptr(),
                                 L.current_stream(x.device)), "field_sdf")
    return out


def field_color(field, x, n, sdfout, dirs=None):
    """forward_color (instant_nsr.py:644-663): -> rgb [B,3]; dirs [B,3] = the view direction of every point for a field with view directions"""
    x = _chk(x.reshape(-1, 3), "x"); n = _chk(n.reshape(-1, 3), "n"); sdfout = _chk(sdfout.reshape(-1, 16), "sdfout")
    if dirs is not None:
        dirs = _chk(dirs.reshape(-1, 3), "dirs", (x.shape[0], 3))
    out = torch.empty((x.shape[0], 3), dtype=_F32, device=x.device)
    L.check(L.lib().ac_field_color_dirs(C.byref(field.c), x.data_ptr(), L.ptr(dirs), n.data_ptr(), sdfout.data_ptr(), x.shape[0], out.data_ptr(),
                                        L.current_stream(x.device)), "field_color")
    return out


# ---------------------------------------------------------------------------------------------------- geometry (csrc/geometry.hip)
def field_sdf_grid(field, axis_x, axis_y, axis_z, bound, negate=False, out=None):
    """ac_field_sdf_grid: forward_sdf(.)[0] on the grid axis_x x axis_y x axis_z (three 1-D float32 device tensors: the coordinates
    torch.linspace gives the reference's extract_fields, models/instant_nsr.py:728-745) -> [nx, ny, nz] float32 on the device; negate: -sdf"""
    ax, ay, az = _chk(axis_x.reshape(-1), "axis_x"), _chk(axis_y.reshape(-1), "axis_y"), _chk(axis_z.reshape(-1), "axis_z")
    nx, ny, nz = ax.shape[0], ay.shape[0], az.shape[0]
    if out is None:
        out = torch.empty((nx, ny, nz), dtype=_F32, device=ax.device)
    elif tuple(out.shape) != (nx, ny, nz) or out.dtype != _F32 or not out.is_contiguous():
        raise RuntimeError("field_sdf_grid: out must be a contiguous float32 [nx, ny, nz] tensor")
    L.check(L.lib().ac_field_sdf_grid(C.byref(field.c), ax.data_ptr(), ay.data_ptr(), az.data_ptr(), nx, ny, nz, float(bound), int(bool(negate)),
                                      out.data_ptr(), L.current_stream(ax.device)), "field_sdf_grid")
    return out


def marching_cubes(volume, iso=0.0, den=1.0, span=(1.0, 1.0, 1.0), lo=(0.0, 0.0, 0.0)):
    """ac_marching_cubes_count + _emit on a device volume [nx, ny, nz] (float32): the surface u = iso, corner flagged <=> u <= iso (PyMCubes' convention),
    one shared vertex per sign-changing grid edge.  -> (vertices [V,3] float64 = index / den * span + lo, triangles [F,3] int32), both on the device.
    One 8-byte device-to-host read between the two calls (the caller owns the output buffers, so it has to know their size)."""
    vol = _chk(volume, "volume")
    if vol.dim() != 3:
        raise RuntimeError("marching_cubes: volume must be [nx, ny, nz]")
    nx, ny, nz = vol.shape
    dev = vol.device
    need = int(L.lib().ac_marching_cubes_scratch(nx, ny, nz))
    if need == 0:
        raise RuntimeError(f"marching_cubes: grid {nx} x {ny} x {nz} unsupported (every side >= 2, fewer than 2^31 points)")
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    st = L.current_stream(dev)
    L.check(L.lib().ac_marching_cubes_count(vol.data_ptr(), nx, ny, nz, float(iso), scratch.data_ptr(), need, counts.data_ptr(), st), "marching_cubes_count")
    nv, nt = (int(v) for v in counts.tolist())
    verts = torch.empty((nv, 3), dtype=torch.float64, device=dev)
    tris = torch.empty((nt, 3), dtype=torch.int32, device=dev)
    span_c, lo_c = (C.c_double * 3)(*[float(v) for v in span]), (C.c_double * 3)(*[float(v) for v in lo])
    if nv or nt:
        L.check(L.lib().ac_marching_cubes_emit(vol.data_ptr(), nx, ny, nz, float(iso), scratch.data_ptr(), need, float(den), span_c, lo_c,
                                               L.ptr(verts) if nv else None, nv, L.ptr(tris) if nt else None, nt, st), "marching_cubes_emit")
    return verts, tris


_DG_SCRATCH = {}


def density_grid_update(field, axis, grid, bound, inv_s=512.0, decay=0.95):
    """ac_density_grid_update: the grid update of update_extra_state (models/instant_nsr.py:303-346) in one launch, IN PLACE on grid [H,H,H];
    axis = torch.linspace(-bound, bound, H) on the device.  -> mean(grid) as a [1] float64 device tensor."""
    axis = _chk(axis.reshape(-1), "axis"); grid = _chk(grid, "density_grid")
    H = axis.shape[0]
    if tuple(grid.shape) != (H, H, H):
        raise RuntimeError("density_grid_update: grid must be [H, H, H] with H = len(axis)")
    dev = grid.device
    need = int(L.lib().ac_density_grid_update_scratch(H))
    if need == 0:
        raise RuntimeError(f"density_grid_update: H = {H} unsupported (2 <= H <= 1024)")
    key = (str(dev), int(L.current_stream(dev) or 0), H)
    sc = _DG_SCRATCH.get(key)
    if sc is None:
        if len(_DG_SCRATCH) > 16:
            _DG_SCRATCH.clear()
        sc = _DG_SCRATCH[key] = torch.zeros(need, dtype=torch.uint8, device=dev)        # zeroed once: the launch re-arms its ticket
    mean = torch.empty(1, dtype=torch.float64, device=dev)
    L.check(L.lib().ac_density_grid_update(C.byref(field.c), axis.data_ptr(), H, float(bound), float(inv_s), float(decay), grid.data_ptr(), mean.data_ptr(),
                                           sc.data_ptr(), need, L.current_stream(dev)), "density_grid_update")
    return mean
