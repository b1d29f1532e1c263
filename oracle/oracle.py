"""oracle/oracle.py -- TEST INFRASTRUCTURE: numpy/ctypes front end of the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (avatarcraft_amd) never does.  See ac_oracle.c for what each entry
restates (reference file:line) and how the oracle is pinned.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)


def build(force=False):
    """Compile libac_oracle.so with the committed Makefile (gcc, seconds)."""
    so = os.path.join(_HERE, "libac_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ac_oracle.c", "ac_oracle_ops.c", "ac_oracle_bwd.c", "ac_oracle_typed.c", "ac_oracle_geometry.c", "ac_oracle.h", "ac_math.h",
                                                 "ac_sh_table.h", "ac_sp_table.h", "ac_mc_table.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libac_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libac_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_pcg32_next_uint.restype = C.c_uint32
        _LIB.orc_pcg32_next_float.restype = C.c_float
        _LIB.orc_fast_hash.restype = C.c_uint32
        _LIB.orc_grid_index.restype = C.c_uint32
        _LIB.orc_eikonal_reduce.restype = C.c_float
        for n in ("orc_test_expf", "orc_test_log1pf", "orc_test_softplus100", "orc_test_sigmoid"):
            getattr(_LIB, n).restype = C.c_float
            getattr(_LIB, n).argtypes = [C.c_float]
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=f32p):
    return a.ctypes.data_as(t) if a is not None else None


# ------------------------------------------------------------------ pcg32 / hash helpers
class Pcg32:
    def __init__(self, initstate, initseq=1):
        self.st = (C.c_uint64 * 2)()
        lib().orc_pcg32_seed(self.st, C.c_uint64(initstate), C.c_uint64(initseq))

    def next_uint(self):
        return int(lib().orc_pcg32_next_uint(self.st))

    def next_float(self):
        return float(lib().orc_pcg32_next_float(self.st))


def fast_hash(pos):
    a = (C.c_uint32 * len(pos))(*pos)
    return int(lib().orc_fast_hash(a, C.c_uint32(len(pos))))


def grid_index(D, Cc, ch, hashmap_size, resolution, pos):
    a = (C.c_uint32 * len(pos))(*pos)
    return int(lib().orc_grid_index(C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(ch), C.c_uint32(hashmap_size),
                                    C.c_uint32(resolution), a))


def hash_level_table(L, S, H):
    scale = np.zeros(L, np.float32)
    res = np.zeros(L, np.uint32)
    lib().orc_hash_level_table(C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(scale), _p(res, u32p))
    return scale, res


def hash_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
    """HashEncoder.__init__ allocation rule (encoder/hashencoder/hashgrid.py:83-108)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        offsets.append(offset)
        offset += min(max_params, (resolution + 1) ** input_dim)
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


# ------------------------------------------------------------------ hash encoder
def hash_encode_forward(inputs, grid, offsets, S, H, calc_grad_inputs=False, want_corner_idx=False):
    inputs = _f(inputs); grid = _f(grid); offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = grid.shape[1]
    out = np.empty((L, B, Cc), np.float32)
    dy_dx = np.empty((B, L * D * Cc), np.float32) if calc_grad_inputs else np.empty(1, np.float32)
    cidx = np.empty((L, B, 1 << D), np.uint32) if want_corner_idx else None
    rc = lib().orc_hash_encode_forward(_p(inputs), _p(grid), _p(offsets, i32p), _p(out), C.c_uint32(B), C.c_uint32(D),
                                       C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                       C.c_int(int(calc_grad_inputs)), _p(dy_dx), _p(cidx, u32p))
    if rc:
        raise RuntimeError("GridEncoding: unsupported D/C")
    return out, (dy_dx if calc_grad_inputs else None), cidx


def hash_encode_backward(grad, inputs, grid, offsets, S, H, dy_dx=None):
    grad = _f(grad); inputs = _f(inputs); grid = _f(grid); offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = grid.shape[1]
    gg = np.zeros_like(grid)
    gi = np.zeros_like(inputs) if dy_dx is not None else np.zeros(1, np.float32)
    dd = _f(dy_dx) if dy_dx is not None else np.zeros(1, np.float32)
    rc = lib().orc_hash_encode_backward(_p(grad), _p(inputs), _p(grid), _p(offsets, i32p), _p(gg), C.c_uint32(B),
                                        C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                        C.c_int(int(dy_dx is not None)), _p(dd), _p(gi))
    if rc:
        raise RuntimeError("GridEncoding: unsupported D/C")
    return gg, (gi if dy_dx is not None else None)


# ------------------------------------------------------------------ SH
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _f(inputs)
    B = inputs.shape[0]
    out = np.empty((B, degree * degree), np.float32)
    dy_dx = np.empty((B, 3 * degree * degree), np.float32) if calc_grad_inputs else np.empty(1, np.float32)
    rc = lib().orc_sh_encode_forward(_p(inputs), _p(out), C.c_uint32(B), C.c_uint32(inputs.shape[1]), C.c_uint32(degree),
                                     C.c_int(int(calc_grad_inputs)), _p(dy_dx))
    if rc:
        raise RuntimeError("SH encoder: unsupported input_dim/degree")
    return out, (dy_dx if calc_grad_inputs else None)


def sh_encode_backward(grad, inputs, degree, dy_dx):
    grad = _f(grad); inputs = _f(inputs); dy_dx = _f(dy_dx)
    gi = np.zeros_like(inputs)
    lib().orc_sh_encode_backward(_p(grad), _p(inputs), C.c_uint32(inputs.shape[0]), C.c_uint32(3), C.c_uint32(degree),
                                 _p(dy_dx), _p(gi))
    return gi


# ------------------------------------------------------------------ the encoders on half / double tensors
# The reference dispatches both extensions over the dtype of their tensors (hashencoder.cu:352,391; shencoder.cu:337,380).  float64: the C restatement
# in ac_oracle_typed.c.  float16: widen, run the fp32 routine, round ONCE -- positions and weights are fp32 in every instantiation
# (hashencoder.cu:125-154), only the accumulation differs: the reference rounds every partial sum to half through c10::Half's operators, which cannot be
# reproduced without its CUDA build and which nothing exercises; the single rounding is what the HIP kernels do and at least as
# accurate.  PARITY UNPINNED for both (see ac_oracle_typed.c).
f64p = C.POINTER(C.c_double)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def hash_encode_forward_typed(inputs, grid, offsets, S, H, calc_grad_inputs=False):
    """(outputs [L,B,C], dy_dx or None) in the dtype of `inputs` (float32 / float16 / float64; grid must share it)"""
    dt = np.asarray(inputs).dtype
    assert np.asarray(grid).dtype == dt, "one dtype per call (the reference instantiates its kernel on inputs.scalar_type())"
    if dt == np.float32:
        out, dy, _ = hash_encode_forward(inputs, grid, offsets, S, H, calc_grad_inputs)
        return out, dy
    if dt == np.float16:
        out, dy, _ = hash_encode_forward(np.asarray(inputs, np.float32), np.asarray(grid, np.float32), offsets, S, H, calc_grad_inputs)
        return out.astype(np.float16), (None if dy is None else dy.astype(np.float16))
    assert dt == np.float64, "inputs must be a floating tensor"
    inputs = _d(inputs); grid = _d(grid); offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = grid.shape[1]
    out = np.empty((L, B, Cc), np.float64)
    dy_dx = np.empty((B, L * D * Cc), np.float64) if calc_grad_inputs else np.empty(1, np.float64)
    rc = lib().orc_hash_encode_forward_f64(_p(inputs, f64p), _p(grid, f64p), _p(offsets, i32p), _p(out, f64p), C.c_uint32(B), C.c_uint32(D),
                                           C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H), C.c_int(int(calc_grad_inputs)), _p(dy_dx, f64p))
    if rc:
        raise RuntimeError("GridEncoding: unsupported D/C")
    return out, (dy_dx if calc_grad_inputs else None)


def hash_encode_backward_typed(grad, inputs, grid, offsets, S, H, dy_dx=None):
    """(grad_grid, grad_inputs or None) in the dtype of `grad`.  float16: the fp32 sums rounded once (the GPU adds half2 atomics in arbitrary order:
    compare with a tolerance that covers one half rounding per contribution)"""
    dt = np.asarray(grad).dtype
    if dt == np.float32:
        return hash_encode_backward(grad, inputs, grid, offsets, S, H, dy_dx)
    if dt == np.float16:
        gg, gi = hash_encode_backward(np.asarray(grad, np.float32), np.asarray(inputs, np.float32), np.asarray(grid, np.float32), offsets, S, H,
                                      None if dy_dx is None else np.asarray(dy_dx, np.float32))
        return gg.astype(np.float16), (None if gi is None else gi.astype(np.float16))
    assert dt == np.float64, "grad must be a floating tensor"
    grad = _d(grad); inputs = _d(inputs); offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = np.asarray(grid).shape[1]
    gg = np.zeros(np.asarray(grid).shape, np.float64)
    gi = np.zeros_like(inputs) if dy_dx is not None else np.zeros(1, np.float64)
    dd = _d(dy_dx) if dy_dx is not None else np.zeros(1, np.float64)
    rc = lib().orc_hash_encode_backward_f64(_p(grad, f64p), _p(inputs, f64p), _p(offsets, i32p), _p(gg, f64p), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                            C.c_uint32(L), C.c_float(S), C.c_uint32(H), C.c_int(int(dy_dx is not None)), _p(dd, f64p), _p(gi, f64p))
    if rc:
        raise RuntimeError("GridEncoding: unsupported D/C")
    return gg, (gi if dy_dx is not None else None)


def sh_encode_forward_typed(inputs, degree, calc_grad_inputs=False):
    dt = np.asarray(inputs).dtype
    if dt == np.float32:
        return sh_encode_forward(inputs, degree, calc_grad_inputs)
    if dt == np.float16:
        out, dy = sh_encode_forward(np.asarray(inputs, np.float32), degree, calc_grad_inputs)
        return out.astype(np.float16), (None if dy is None else dy.astype(np.float16))
    assert dt == np.float64, "inputs must be a floating tensor"
    inputs = _d(inputs)
    B = inputs.shape[0]
    out = np.empty((B, degree * degree), np.float64)
    dy_dx = np.empty((B, 3 * degree * degree), np.float64) if calc_grad_inputs else np.empty(1, np.float64)
    rc = lib().orc_sh_encode_forward_f64(_p(inputs, f64p), _p(out, f64p), C.c_uint32(B), C.c_uint32(inputs.shape[1]), C.c_uint32(degree),
                                         C.c_int(int(calc_grad_inputs)), _p(dy_dx, f64p))
    if rc:
        raise RuntimeError("SH encoder: unsupported input_dim/degree")
    return out, (dy_dx if calc_grad_inputs else None)


def sh_encode_backward_typed(grad, inputs, degree, dy_dx):
    dt = np.asarray(grad).dtype
    if dt == np.float32:
        return sh_encode_backward(grad, inputs, degree, dy_dx)
    if dt == np.float16:
        return sh_encode_backward(np.asarray(grad, np.float32), np.asarray(inputs, np.float32), degree, np.asarray(dy_dx, np.float32)).astype(np.float16)
    assert dt == np.float64, "grad must be a floating tensor"
    grad = _d(grad); dy_dx = _d(dy_dx)
    gi = np.zeros((grad.shape[0], 3), np.float64)
    lib().orc_sh_encode_backward_f64(_p(grad, f64p), C.c_uint32(grad.shape[0]), C.c_uint32(3), C.c_uint32(degree), _p(dy_dx, f64p), _p(gi, f64p))
    return gi


# ------------------------------------------------------------------ raymarching
def march_rays_train(rays_o, rays_d, grid, mean_density, bound, M=None, perturb=0, counter=None):
    rays_o = _f(rays_o).reshape(-1, 3); rays_d = _f(rays_d).reshape(-1, 3); grid = _f(grid)
    N, H = rays_o.shape[0], grid.shape[0]
    M = N * 1024 if M is None else M
    xyzs = np.zeros((M, 3), np.float32); dirs = np.zeros((M, 3), np.float32); deltas = np.zeros(M, np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32) if counter is None else counter
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), C.c_float(mean_density), C.c_int(0), C.c_float(bound),
                               C.c_uint32(N), C.c_uint32(H), C.c_uint32(M), _p(xyzs), _p(dirs), _p(deltas),
                               _p(rays, i32p), _p(counter, i32p), C.c_uint32(int(perturb)))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, bound=1.0):
    sigmas = _f(sigmas); rgbs = _f(rgbs); deltas = _f(deltas); rays = np.ascontiguousarray(rays, np.int32)
    M, N = sigmas.shape[0], rays.shape[0]
    ws = np.empty(N, np.float32); img = np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays, i32p), C.c_float(bound),
                                           C.c_uint32(M), C.c_uint32(N), _p(ws), _p(img))
    return ws, img


def composite_rays_train_backward(grad_ws, grad_img, sigmas, rgbs, deltas, rays, ws, img, bound=1.0):
    args = [_f(a) for a in (grad_ws, grad_img, sigmas, rgbs, deltas)]
    rays = np.ascontiguousarray(rays, np.int32); ws = _f(ws); img = _f(img)
    M, N = args[2].shape[0], rays.shape[0]
    gs = np.zeros(M, np.float32); gc = np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), _p(args[4]),
                                            _p(rays, i32p), _p(ws), _p(img), C.c_float(bound), C.c_uint32(M),
                                            C.c_uint32(N), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, grid, mean_density, near, far, perturb=0):
    rays_alive = np.ascontiguousarray(rays_alive, np.int32); rays_t = _f(rays_t)
    rays_o = _f(rays_o).reshape(-1, 3); rays_d = _f(rays_d).reshape(-1, 3); grid = _f(grid); near = _f(near); far = _f(far)
    M = n_alive * n_step
    xyzs = np.zeros((M, 3), np.float32); dirs = np.zeros((M, 3), np.float32); deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(rays_alive, i32p), _p(rays_t), _p(rays_o), _p(rays_d),
                         C.c_float(bound), C.c_uint32(grid.shape[0]), _p(grid), C.c_float(mean_density), _p(near), _p(far),
                         _p(xyzs), _p(dirs), _p(deltas), C.c_uint32(int(perturb)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map):
    """in place on rays_t, weights, depth, image, normal_map (float32 contiguous numpy arrays)"""
    rays_alive = np.ascontiguousarray(rays_alive, np.int32)
    lib().orc_composite_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(rays_alive, i32p), _p(rays_t), _p(_f(sigmas)),
                             _p(_f(rgbs)), _p(_f(normals)), _p(_f(deltas)), _p(weights), _p(depth), _p(image), _p(normal_map))


def compact_rays(n_alive, rays_alive_old, rays_t_old):
    rays_alive_old = np.ascontiguousarray(rays_alive_old, np.int32); rays_t_old = _f(rays_t_old)
    ra = np.zeros_like(rays_alive_old); rt = np.zeros_like(rays_t_old); cnt = np.zeros(1, np.int32)
    lib().orc_compact_rays(C.c_uint32(n_alive), _p(ra, i32p), _p(rays_alive_old, i32p), _p(rt), _p(rays_t_old), _p(cnt, i32p))
    return ra, rt, int(cnt[0])


# ------------------------------------------------------------------ Instant-NSR field + renderer
class _Field(C.Structure):
    _fields_ = [("table", f32p), ("offsets", i32p), ("scale", C.c_float * 16), ("res", C.c_uint32 * 16),
                ("W1", f32p), ("b1", f32p), ("W2", f32p), ("b2", f32p), ("Wc1", f32p), ("Wc2", f32p), ("Wc3", f32p), ("Wsh", f32p)]


class _Opts(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("num_steps", C.c_int32), ("upsample_steps", C.c_int32), ("bound", C.c_float),
                ("inv_s", C.c_float), ("cos_anneal_ratio", C.c_float), ("fd_eps", C.c_float), ("perturb", C.c_int32)]


class _Out(C.Structure):
    _fields_ = [("image", f32p), ("weights_sum", f32p), ("depth", f32p), ("normal_map", f32p), ("eik", f32p),
                ("z_vals", f32p), ("weights", f32p), ("alpha", f32p), ("color", f32p), ("sdf", f32p),
                ("gradient", f32p), ("ss_inds", i32p), ("sort_index", i32p)]


class Field:
    """Effective (weight-normed) parameters of the default NeRFNetwork (models/instant_nsr.py:478-591):
    hash table [6119857,2] + offsets[17], W1[64,35], b1[64], W2[16,64], b2[16], Wc1[64,21], Wc2[64,64], Wc3[3,64]."""

    def __init__(self, table, offsets, W1, b1, W2, b2, Wc1, Wc2, Wc3, per_level_scale, base_resolution=16):
        """Wc1 [64,21] = [x | n | geo_feat], or the [64,37] matrix of NeRFNetwork(use_viewdirs=True) = [x | sh(d) (16) | n | geo_feat] (models/instant_nsr.py:648-650):
        its 16 view-direction columns are then kept apart (orc_field.Wsh) and enter layer 1 as a per-ray bias"""
        Wc1 = _f(Wc1)
        Wsh = None
        if Wc1.shape == (64, 37):
            Wsh = np.ascontiguousarray(Wc1[:, 3:19])
            Wc1 = np.ascontiguousarray(np.concatenate([Wc1[:, :3], Wc1[:, 19:]], 1))
        self.arrs = dict(table=_f(table), W1=_f(W1), b1=_f(b1), W2=_f(W2), b2=_f(b2), Wc1=_f(Wc1), Wc2=_f(Wc2), Wc3=_f(Wc3))
        if Wsh is not None:
            self.arrs["Wsh"] = Wsh
        self.has_viewdirs = Wsh is not None
        self.offsets = np.ascontiguousarray(offsets, np.int32)
        assert self.offsets.shape[0] == 17 and self.arrs["table"].shape[1] == 2
        assert self.arrs["W1"].shape == (64, 35) and self.arrs["W2"].shape == (16, 64)
        assert self.arrs["Wc1"].shape == (64, 21) and self.arrs["Wc2"].shape == (64, 64) and self.arrs["Wc3"].shape == (3, 64)
        self.S = float(np.log2(per_level_scale))
        self.H = int(base_resolution)
        self.scale, self.res = hash_level_table(16, self.S, self.H)
        s = _Field()
        s.table = _p(self.arrs["table"]); s.offsets = _p(self.offsets, i32p)
        for i in range(16):
            s.scale[i] = float(self.scale[i]); s.res[i] = int(self.res[i])
        for k in ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3"):
            setattr(s, k, _p(self.arrs[k]))
        if self.has_viewdirs:
            s.Wsh = _p(self.arrs["Wsh"])
        self.c = s

    def sdf(self, x, bound):
        x = _f(x).reshape(-1, 3)
        out = np.empty((x.shape[0], 16), np.float32)
        lib().orc_field_sdf(C.byref(self.c), _p(x), C.c_uint32(x.shape[0]), C.c_float(bound), _p(out))
        return out

    def color(self, x, n, sdfout, dirs=None):
        x = _f(x).reshape(-1, 3); n = _f(n).reshape(-1, 3); sdfout = _f(sdfout).reshape(-1, 16)
        out = np.empty((x.shape[0], 3), np.float32)
        if dirs is not None:
            dirs = _f(dirs).reshape(-1, 3)
        lib().orc_field_color_dirs(C.byref(self.c), _p(x), _p(dirs), _p(n), _p(sdfout), C.c_uint32(x.shape[0]), _p(out))
        return out


def field_samples(field, xyzs, dirs, deltas, bound, eps, inv_s, cos_anneal_ratio=1.0):
    """per-sample (alpha, rgb, normal, sdf, gradient) of packed samples (orc_field_samples); deltas [M] or [M,2] (column 0)"""
    xyzs = _f(xyzs).reshape(-1, 3); dirs = _f(dirs).reshape(-1, 3); deltas = _f(deltas)
    M = xyzs.shape[0]
    stride = 1 if deltas.ndim == 1 else deltas.shape[1]
    out = dict(alpha=np.empty(M, np.float32), rgb=np.empty((M, 3), np.float32), normal=np.empty((M, 3), np.float32), sdf=np.empty(M, np.float32),
               gradient=np.empty((M, 3), np.float32))
    if M:
        lib().orc_field_samples(C.byref(field.c), _p(xyzs), _p(dirs), _p(deltas), C.c_uint32(stride), C.c_uint32(M), C.c_float(bound), C.c_float(eps),
                                C.c_float(inv_s), C.c_float(cos_anneal_ratio), _p(out["alpha"]), _p(out["rgb"]), _p(out["normal"]), _p(out["sdf"]),
                                _p(out["gradient"]))
    return out


def _near_far_cube(rays_o, rays_d, bound):
    """near_far_from_bound(type='cube'), models/instant_nsr.py:58-77, fp32"""
    o, d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        tmin = (np.float32(-bound) - o) / (d + np.float32(1e-15))
        tmax = (np.float32(bound) - o) / (d + np.float32(1e-15))
    near = np.where(tmin < tmax, tmin, tmax).max(axis=1)
    far = np.where(tmin > tmax, tmin, tmax).min(axis=1)
    return np.maximum(near, np.float32(0.05)).astype(np.float32), far.astype(np.float32)


def run_cuda_train(field, rays_o, rays_d, grid, mean_density, bound, eps, inv_s, cos_anneal_ratio=1.0, bg=1.0, perturb=0, mean_count=-1, align=128):
    """training form of the occupancy-grid render (NeRFRenderer.run_cuda, train()): march_rays_train -> field_samples -> composite_rays_train_forward.
    Returns dict(image, weights_sum, normal_map, gradient_error, rays, counter, xyzs, dirs, deltas, alpha, rgb, normal, gradient)."""
    N = _f(rays_o).reshape(-1, 3).shape[0]
    M = None
    if mean_count > 0:
        M = mean_count + (align - mean_count % align) if align > 0 else mean_count
    xyzs, dirs, deltas, rays, counter = march_rays_train(rays_o, rays_d, grid, mean_density, bound, M=M, perturb=perturb)
    if mean_count <= 0:
        m = int(counter[0])
        m = m + (align - m % align) if align > 0 else m
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    fs = field_samples(field, xyzs, dirs, deltas, bound, eps, inv_s, cos_anneal_ratio)
    ws, img = composite_rays_train_forward(fs["alpha"], fs["rgb"], deltas, rays, bound)
    _, nmap = composite_rays_train_forward(fs["alpha"], fs["normal"], deltas, rays, bound)
    n_valid = int(counter[0])
    if mean_count > 0:          # rays the budget left out wrote nothing: the marched samples end with the last ray that fitted (rows behind it are zeros, not samples)
        ends = rays[:, 1].astype(np.int64) + rays[:, 2]
        fit = (rays[:, 2] > 0) & (ends < xyzs.shape[0])
        n_valid = int(ends[fit].max()) if fit.any() else 0
    valid = (np.arange(xyzs.shape[0]) < n_valid).astype(np.float32)
    relax = (np.sqrt((xyzs.astype(np.float64) ** 2).sum(1)) < 1.2).astype(np.float64) * valid
    gerr = (np.sqrt((fs["gradient"].astype(np.float64) ** 2).sum(1)) - 1.0) ** 2
    res = dict(fs)
    res.update(image=(img + (1 - ws)[:, None] * np.asarray(bg, np.float32)).astype(np.float32), weights_sum=ws, normal_map=nmap,
               gradient_error=float((relax * gerr).sum() / (relax.sum() + 1e-5)), rays=rays, counter=counter, xyzs=xyzs, dirs=dirs, deltas=deltas)
    return res


def run_cuda_eval(field, rays_o, rays_d, grid, mean_density, bound, eps, inv_s, cos_anneal_ratio=1.0, bg=1.0, max_steps=1024, align=128):
    """inference form (eval()): rounds of compact_rays / march_rays / field_samples / composite_rays with n_step = clamp(N // n_alive, 1, 8).
    Returns dict(image, weights_sum, depth, normal_map, rounds, alive_per_round)."""
    o, d = _f(rays_o).reshape(-1, 3), _f(rays_d).reshape(-1, 3)
    N = o.shape[0]
    ws, depth = np.zeros(N, np.float32), np.zeros(N, np.float32)
    image, nmap = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    near, far = _near_far_cube(o, d, bound)
    rays_alive = np.arange(N, dtype=np.int32); rays_t = near.copy()
    n_alive, step, alive_log = N, 0, []
    while step < max_steps:
        if step > 0:
            rays_alive, rays_t, n_alive = compact_rays(n_alive, rays_alive, rays_t)
        if n_alive <= 0:
            break
        alive_log.append(n_alive)
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, grid, mean_density, near, far)
        fs = field_samples(field, xyzs, dirs, deltas, bound, eps, inv_s, cos_anneal_ratio)
        rays_t = np.ascontiguousarray(rays_t, np.float32)
        composite_rays(n_alive, n_step, rays_alive, rays_t, fs["alpha"], fs["rgb"], fs["normal"], deltas, ws, depth, image, nmap)
        step += n_step
    with np.errstate(divide="ignore", invalid="ignore"):
        dn = (np.maximum(depth - near, np.float32(0)) / (far - near)).astype(np.float32)
    return dict(image=(image + (1 - ws)[:, None] * np.asarray(bg, np.float32)).astype(np.float32), weights_sum=ws, depth=dn, normal_map=nmap,
                rounds=len(alive_log), alive_per_round=alive_log)


def linspace_tables(num_steps):
    """The two torch.linspace tables the renderer consumes (instant_nsr.py:155, :34), made by torch on
    the CPU exactly as the reference makes them."""
    import torch
    lin_z = torch.linspace(0.0, 1.0, num_steps).numpy().astype(np.float32)
    lin_u = torch.linspace(0. + 0.5 / 16, 1. - 0.5 / 16, steps=16).numpy().astype(np.float32)
    return lin_z, lin_u


class _WarpCtx(C.Structure):
    _fields_ = [("verts", f32p), ("faces", i32p), ("T", C.POINTER(C.c_double)), ("V", C.c_uint32), ("F", C.c_uint32),
                ("threshold", C.c_double), ("geo_threshold", C.c_float), ("use_mesh_guide", C.c_int32),
                ("can_mid", f32p), ("mask", C.POINTER(C.c_uint8))]


def render_rays(field, rays_o, rays_d, num_steps=64, upsample_steps=64, bound=1.6, inv_s=None, bg=None, noise=None,
                cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, extras=True, warp=None):
    """NeRFRenderer.run (models/instant_nsr.py:133-299).  render_can=True unless warp = dict(verts, faces, Ts[, threshold,
    use_mesh_guide]) is given (render_can=False: SMPL-guided inverse warp of the samples).  Returns a dict."""
    rays_o = _f(rays_o).reshape(-1, 3); rays_d = _f(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    T = num_steps + upsample_steps
    nup = upsample_steps // 16
    lin_z, lin_u = linspace_tables(num_steps)
    op = _Opts(N, num_steps, upsample_steps, bound, float(inv_s), float(cos_anneal_ratio),
               float(np.float32(0.005 * (1.0 - normal_epsilon_ratio))), int(noise is not None))
    res = dict(image=np.empty((N, 3), np.float32), weights_sum=np.empty(N, np.float32), depth=np.empty(N, np.float32),
               normal_map=np.empty((N, 3), np.float32), eik=np.empty((N, 2), np.float32))
    if extras:
        res.update(z_vals=np.empty((N, T), np.float32), weights=np.empty((N, T), np.float32),
                   alpha=np.empty((N, T), np.float32), color=np.empty((N, T, 3), np.float32),
                   sdf=np.empty((N, T), np.float32), gradient=np.empty((N, T, 3), np.float32),
                   ss_inds=np.empty((N, max(nup, 1), 16), np.int32), sort_index=np.empty((N, max(nup, 1), 128), np.int32))
    o = _Out()
    for k, v in res.items():
        setattr(o, k, _p(v, i32p if v.dtype == np.int32 else f32p))
    bgc = _f(bg).reshape(-1, 3) if bg is not None else None
    nz = _f(noise).reshape(N, num_steps) if noise is not None else None
    if warp is None:
        rc = lib().orc_render_rays(C.byref(field.c), C.byref(op), _p(rays_o), _p(rays_d), _p(bgc), _p(nz), _p(lin_z), _p(lin_u),
                                   C.byref(o))
    else:
        verts = _f(warp["verts"]).reshape(-1, 3)
        faces = np.ascontiguousarray(np.asarray(warp["faces"])[:, :3], np.int32)
        Ts = np.ascontiguousarray(warp["Ts"], np.float64)
        thr = warp.get("threshold", 0.05)
        res["can_mid"] = np.empty((N, T, 3), np.float32); res["mask"] = np.empty((N, T), np.uint8)
        wc = _WarpCtx(_p(verts), _p(faces, i32p), Ts.ctypes.data_as(C.POINTER(C.c_double)), verts.shape[0], faces.shape[0], float(thr),
                      float(thr), int(bool(warp.get("use_mesh_guide", True))), _p(res["can_mid"]),
                      res["mask"].ctypes.data_as(C.POINTER(C.c_uint8)))
        rc = lib().orc_render_rays_warped(C.byref(field.c), C.byref(op), _p(rays_o), _p(rays_d), _p(bgc), _p(nz), _p(lin_z), _p(lin_u),
                                          C.byref(wc), C.byref(o))
    if rc:
        raise RuntimeError("render_rays: unsupported num_steps/upsample_steps")
    res["gradient_error"] = float(lib().orc_eikonal_reduce(_p(res["eik"]), C.c_int32(N)))
    return res


# ------------------------------------------------------------------ SMPL-guided warp
def mesh_near_far(rays_o, rays_d, verts, geo_threshold=0.05):
    rays_o = _f(rays_o).reshape(-1, 3); rays_d = _f(rays_d).reshape(-1, 3); verts = _f(verts).reshape(-1, 3)
    N = rays_o.shape[0]
    near = np.empty(N, np.float32); far = np.empty(N, np.float32)
    lib().orc_mesh_near_far(_p(rays_o), _p(rays_d), _p(verts), C.c_uint32(N), C.c_uint32(verts.shape[0]), C.c_float(geo_threshold), _p(near), _p(far))
    return near, far


def warp_samples(pts, verts, faces, T, threshold=0.05):
    """warp_samples_to_canonical (utils/ray_utils.py:62-90): returns can_pts f64, closest f64, dist2 f64, face_id, mask"""
    pts = _f(pts).reshape(-1, 3); verts = _f(verts).reshape(-1, 3)
    faces = np.ascontiguousarray(faces[:, :3], np.int32); T = np.ascontiguousarray(T, np.float64)
    P = pts.shape[0]
    can = np.empty((P, 3), np.float64); clo = np.empty((P, 3), np.float64); d2 = np.empty(P, np.float64)
    fid = np.empty(P, np.int32); mask = np.empty(P, np.uint8)
    dp = C.POINTER(C.c_double)
    lib().orc_warp_samples(_p(pts), _p(verts), _p(faces, i32p), T.ctypes.data_as(dp), C.c_uint32(P), C.c_uint32(faces.shape[0]),
                           C.c_double(threshold), can.ctypes.data_as(dp), clo.ctypes.data_as(dp), d2.ctypes.data_as(dp), _p(fid, i32p),
                           mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return can, clo, d2, fid, mask.astype(bool)


def update_density_grid(field, grid, bound, decay=0.95, inv_s=512.0, resolution=129):
    """NeRFRenderer.update_extra_state's grid update (models/instant_nsr.py:309-343): sdf on linspace(-bound, bound, 129)^3 -> logistic
    density inv_s * e^(-inv_s |sdf|) / (1 + e^(-inv_s |sdf|)) -> zero-pad by one at the far ends, 2x2x2 max pool (stride 1) ->
    maximum(grid * decay, new).  Returns (new grid, mean density).  numpy fp32 over the C oracle's forward_sdf."""
    ax = np.linspace(-bound, bound, resolution, dtype=np.float32)            # torch.linspace(-b, b, 129) in fp32
    xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")
    pts = np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1).astype(np.float32)
    sdf = field.sdf(pts, bound)[:, 0].astype(np.float32)
    a = np.float32(inv_s) * np.abs(sdf)
    e = np.exp(-a, dtype=np.float32)
    dens = (np.float32(inv_s) * e / (np.float32(1) + e)).reshape(resolution, resolution, resolution)
    pad = np.zeros((resolution + 1,) * 3, np.float32)
    pad[:-1, :-1, :-1] = dens
    pooled = pad[:-1, :-1, :-1]
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                pooled = np.maximum(pooled, pad[dx:dx + resolution, dy:dy + resolution, dz:dz + resolution])
    new = np.maximum(np.asarray(grid, np.float32) * np.float32(decay), pooled)
    return new, float(new.mean(dtype=np.float64))


# ------------------------------------------------------------------ backward of the render core (fp64 witness, ac_oracle_bwd.c)
class _CoreGrads(C.Structure):
    _fields_ = [("g_table", C.POINTER(C.c_double)), ("g_params", C.POINTER(C.c_double)), ("g_inv_s", C.POINTER(C.c_double)),
                ("fwd", C.POINTER(C.c_double)), ("gradient_error", C.POINTER(C.c_double))]


CORE_PARAM_SHAPES = (("W1", (64, 35)), ("b1", (64,)), ("W2", (16, 64)), ("b2", (16,)), ("Wc1", (64, 21)), ("Wc2", (64, 64)), ("Wc3", (3, 64)))


def render_core_backward(field, rays_o, rays_d, z_vals, num_steps, upsample_steps, bound, inv_s, bg=None, g_image=None, g_weights_sum=None,
                         g_depth=None, g_normal_map=None, g_eik=0.0, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, ext_pts=None, mask=None, near_far=None):
    """ext_pts [N,T,3], mask [N,T], near_far = (near [N], far [N]): posed space (run(render_can=False)) -- the warped mid points the field is
    evaluated at, the alpha mask, the mesh-guided sampling range (inf = the cube's); all constants of the differentiation.
    d loss / d (hash table, effective MLP matrices, inv_s) for loss = <g_image, image> + <g_weights_sum, weights_sum> + <g_depth, depth> +
    <g_normal_map, normal_map> + g_eik * gradient_error of NeRFRenderer.run's render core at the given (constant) sample positions z_vals [N,T]
    -- float64, analytic (reference models/instant_nsr.py:190-299 under autograd).  Returns a dict: g_table [n_entries,2], g_W1 ... g_Wc3,
    g_inv_s, and the fp64 forward (image, weights_sum, depth, normal_map, gradient_error)."""
    rays_o = _f(rays_o).reshape(-1, 3); rays_d = _f(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    T = num_steps + upsample_steps
    z = _f(z_vals).reshape(N, T)
    op = _Opts(N, num_steps, upsample_steps, bound, float(inv_s), float(cos_anneal_ratio), float(np.float32(0.005 * (1.0 - normal_epsilon_ratio))), 0)
    dp = C.POINTER(C.c_double)
    g_table = np.zeros(field.arrs["table"].shape, np.float64)
    npar = sum(int(np.prod(s)) for _, s in CORE_PARAM_SHAPES)
    g_par = np.zeros(npar + (64 * 16 if getattr(field, "has_viewdirs", False) else 0), np.float64); g_s = np.zeros(1, np.float64); fwd = np.zeros((N, 8), np.float64); ge = np.zeros(1, np.float64)
    cg = _CoreGrads(g_table.ctypes.data_as(dp), g_par.ctypes.data_as(dp), g_s.ctypes.data_as(dp), fwd.ctypes.data_as(dp), ge.ctypes.data_as(dp))
    opt = lambda a, shape: None if a is None else _f(a).reshape(shape)
    gi, gw, gd, gm, bgc = opt(g_image, (N, 3)), opt(g_weights_sum, (N,)), opt(g_depth, (N,)), opt(g_normal_map, (N, 3)), opt(bg, (N, 3))
    fn = lib().orc_render_core_backward_posed
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, f32p, f32p, f32p, f32p, f32p, C.c_void_p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_double, C.c_void_p]
    ep = opt(ext_pts, (N, T, 3))
    mk = None if mask is None else np.ascontiguousarray(np.asarray(mask).reshape(N, T) != 0, dtype=np.uint8)
    nm, fm = (None, None) if near_far is None else (opt(near_far[0], (N,)), opt(near_far[1], (N,)))
    rc = fn(C.byref(field.c), C.byref(op), _p(rays_o), _p(rays_d), _p(bgc), _p(z), _p(ep), None if mk is None else mk.ctypes.data, _p(nm), _p(fm),
            _p(gi), _p(gw), _p(gd), _p(gm), float(g_eik), C.byref(cg))
    if rc:
        raise RuntimeError("render_core_backward: unsupported configuration")
    res = dict(g_table=g_table, g_inv_s=float(g_s[0]), image=fwd[:, 0:3], weights_sum=fwd[:, 3], depth=fwd[:, 4], normal_map=fwd[:, 5:8],
               gradient_error=float(ge[0]))
    off = 0
    for name, shape in CORE_PARAM_SHAPES:
        n = int(np.prod(shape))
        res["g_" + name] = g_par[off:off + n].reshape(shape).copy(); off += n
    if getattr(field, "has_viewdirs", False):
        # a field with view directions: the gradient of the reference's [64,37] color_net.0 matrix, columns in ITS order [x | sh(d) | n | geo_feat]
        g_sh = g_par[off:off + 64 * 16].reshape(64, 16)
        res["g_Wsh"] = g_sh.copy()
        res["g_Wc1_37"] = np.concatenate([res["g_Wc1"][:, :3], g_sh, res["g_Wc1"][:, 3:]], 1)
    return res


def weight_norm_backward(v, g, g_w):
    """float64 chain rule of torch.nn.utils.weight_norm (dim 0): w = g v / |v|_row  ->  (d/dv, d/dg) from d/dw"""
    v = np.asarray(v, np.float64); g = np.asarray(g, np.float64).reshape(-1, 1); g_w = np.asarray(g_w, np.float64)
    nrm = np.sqrt((v * v).sum(1, keepdims=True))
    dot = (g_w * v).sum(1, keepdims=True)
    return g / nrm * (g_w - v * dot / (nrm * nrm)), dot / nrm


# ------------------------------------------------------------------ mesh export (ac_oracle_geometry.c)
def field_sdf_grid(field, bound, resolution, negate=False):
    """extract_fields (models/instant_nsr.py:728-745): forward_sdf on linspace(-bound, bound, resolution)^3 -> [res, res, res] fp32 (negate: -sdf)"""
    import torch
    ax = torch.linspace(-bound, bound, resolution).numpy()                  # the reference's own axis (host, fp32)
    xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")
    pts = np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], 1).astype(np.float32)
    u = field.sdf(pts, bound)[:, 0].astype(np.float32).reshape(resolution, resolution, resolution)
    return -u if negate else u


def marching_cubes(volume, iso=0.0, den=1.0, span=(1.0, 1.0, 1.0), lo=(0.0, 0.0, 0.0)):
    """mcubes.marching_cubes(u, iso) restated (see ac_oracle_geometry.c): -> vertices [V,3] float64 (= index / den * span + lo), triangles [F,3] int32"""
    vol = _f(volume)
    nx, ny, nz = vol.shape
    dp = C.POINTER(C.c_double)
    span_a, lo_a = np.asarray(span, dtype=np.float64), np.asarray(lo, dtype=np.float64)
    counts = np.zeros(2, dtype=np.uint32)
    args = lambda v, t: (_p(vol), C.c_uint32(nx), C.c_uint32(ny), C.c_uint32(nz), C.c_float(iso), C.c_double(den), span_a.ctypes.data_as(dp),
                         lo_a.ctypes.data_as(dp), v, t, _p(counts, u32p))
    if lib().orc_marching_cubes(*args(None, None)) != 0:
        raise MemoryError("orc_marching_cubes")
    verts = np.zeros((int(counts[0]), 3), dtype=np.float64)
    tris = np.zeros((int(counts[1]), 3), dtype=np.int32)
    lib().orc_marching_cubes(*args(verts.ctypes.data_as(dp), _p(tris, i32p)))
    return verts, tris
