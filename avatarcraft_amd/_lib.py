"""ctypes binding of libavatarcraft_hip.so (the C ABI declared in include/avatarcraft_hip.h).

This is the only place where Python touches the native library.  It fails LOUDLY when the
library is missing or cannot be loaded: there is no CPU / eager fallback for the hot path.
PyTorch is used by the callers for device memory and streams only; nothing here takes torch
types -- pointers are plain integers (tensor.data_ptr()).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AC_LIB_PATH") or os.path.join(_HERE, "libavatarcraft_hip.so")      # AC_LIB_PATH: ablation builds (tools/)

_lib = None

vp = C.c_void_p
u32 = C.c_uint32
i32 = C.c_int32
f32 = C.c_float


class ac_field(C.Structure):
    _fields_ = [("table", vp), ("offsets", i32 * 17), ("S", f32), ("H", u32),
                ("W1", vp), ("b1", vp), ("W2", vp), ("b2", vp), ("Wc1", vp), ("Wc2", vp), ("Wc3", vp), ("prepared", vp), ("Wc1_sh", vp)]


class ac_render_opts(C.Structure):
    _fields_ = [("n_rays", i32), ("num_steps", i32), ("upsample_steps", i32), ("bound", f32), ("inv_s", f32),
                ("cos_anneal_ratio", f32), ("fd_eps", f32), ("perturb", i32), ("inv_s_dev", vp), ("near_m", vp), ("far_m", vp), ("precision", i32), ("skip_masked", i32), ("opacity_only", i32)]


class ac_render_out(C.Structure):
    _fields_ = [("image", vp), ("weights_sum", vp), ("depth", vp), ("normal_map", vp), ("eik", vp), ("z_vals", vp),
                ("weights", vp), ("alpha", vp), ("color", vp), ("sdf", vp), ("gradient", vp), ("ss_inds", vp),
                ("sort_index", vp), ("sdf_out16", vp), ("pts", vp), ("feat7", vp), ("eik_reduced", vp)]


class ac_core_saved(C.Structure):
    _fields_ = [("z_vals", vp), ("pts", vp), ("sdf", vp), ("sdf_out16", vp), ("gradient", vp), ("color", vp), ("eik_den", vp), ("feat7", vp), ("mask", vp), ("sh_bias", vp)]


class ac_core_upstream(C.Structure):
    _fields_ = [("g_image", vp), ("g_weights_sum", vp), ("g_depth", vp), ("g_normal_map", vp), ("g_eik", vp), ("eik_group_rays", i32), ("eik_den_stride", i32)]


class ac_core_grads(C.Structure):
    _fields_ = [("g_table", vp), ("g_sdf_params", vp), ("g_color_params", vp), ("g_inv_s_per_ray", vp), ("side_stream", vp), ("split_level", C.c_int32),
                ("reserved", C.c_int32), ("g_sh_tiles", vp)]


class ac_wn_layer(C.Structure):
    _fields_ = [("v", vp), ("g", vp), ("w", vp), ("rows", u32), ("cols", u32), ("w_stride", u32), ("reserved_", u32)]


class ac_pg_entry(C.Structure):
    _fields_ = [("src", vp), ("v", vp), ("g", vp), ("dst", vp), ("dst2", vp), ("rows", u32), ("cols", u32), ("src_stride", u32), ("kind", i32)]


class ac_adam_entry(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_uint64)]


AC_ADAM_MAX_TENSORS = 16


class ac_warp_mesh(C.Structure):
    _fields_ = [("verts", vp), ("faces", vp), ("T", vp), ("V", u32), ("F", u32), ("threshold", C.c_double), ("geo_threshold", f32),
                ("use_mesh_guide", i32), ("accel", vp), ("seed_faces", vp), ("seed_stride", u32)]


_SIGS = {
    "ac_version": ([], C.c_int),
    "ac_last_error": ([], C.c_char_p),
    "ac_debug_hold_cus": ([u32, u32, u32, vp], C.c_int),
    "ac_packed_shading_forward": ([vp, vp, vp, vp, vp, u32, u32, vp, f32, vp, f32, vp, vp, vp, vp], C.c_int),
    "ac_packed_shading_backward": ([vp, vp, vp, vp, vp, u32, u32, vp, f32, vp, f32, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_set_occupancy_barrier_ms": ([u32], u32),
    "ac_hash_level_table": ([u32, f32, u32, vp, vp], None),
    "ac_hash_encode_forward": ([vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, C.c_int, vp, vp], C.c_int),
    "ac_hash_encode_backward": ([vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, C.c_int, vp, vp, vp], C.c_int),
    "ac_hash_encode_backward_scratch": ([vp, u32, u32, u32, f32, u32, u32], C.c_size_t),
    "ac_hash_encode_backward_ws": ([vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, C.c_int, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "ac_hash_corner_indices": ([vp, vp, vp, u32, u32, u32, f32, u32, vp], C.c_int),
    "ac_hash_encode_forward_typed": ([C.c_int, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, C.c_int, vp, vp], C.c_int),
    "ac_hash_encode_backward_typed": ([C.c_int, vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, f32, u32, C.c_int, vp, vp, vp], C.c_int),
    "ac_sh_encode_forward_typed": ([C.c_int, vp, vp, u32, u32, u32, C.c_int, vp, vp], C.c_int),
    "ac_sh_encode_backward_typed": ([C.c_int, vp, vp, u32, u32, u32, vp, vp, vp], C.c_int),
    "ac_variance_forward": ([vp, vp, vp], C.c_int),
    "ac_adam_step": ([C.POINTER(ac_adam_entry), u32, f32, f32, f32, f32, f32, f32, f32, C.c_int, vp], C.c_int),
    "ac_sh_encode_forward": ([vp, vp, u32, u32, u32, C.c_int, vp, vp], C.c_int),
    "ac_sh_encode_backward": ([vp, vp, u32, u32, u32, vp, vp, vp], C.c_int),
    "ac_march_rays_train_scratch": ([u32], C.c_size_t),
    "ac_march_rays_train": ([vp, vp, vp, f32, C.c_int, f32, u32, u32, u32, vp, vp, vp, vp, vp, u32, vp, vp], C.c_int),
    "ac_composite_rays_train_forward": ([vp, vp, vp, vp, f32, u32, u32, vp, vp, vp], C.c_int),
    "ac_composite_rays_train_backward": ([vp, vp, vp, vp, vp, vp, vp, vp, f32, u32, u32, vp, vp, vp], C.c_int),
    "ac_march_rays": ([u32, u32, vp, vp, vp, vp, f32, u32, vp, f32, vp, vp, vp, vp, vp, u32, vp], C.c_int),
    "ac_composite_rays": ([u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_compact_rays": ([u32, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_field_prepare": ([C.POINTER(ac_field), vp, vp], C.c_int),
    "ac_render_rays": ([C.POINTER(ac_field), C.POINTER(ac_render_opts), vp, vp, vp, vp, vp, vp, C.POINTER(ac_render_out), vp], C.c_int),
    "ac_render_rays_pair": ([C.POINTER(ac_field), C.POINTER(ac_render_opts), vp, vp, vp, vp, vp, vp, C.POINTER(ac_render_out), vp], C.c_int),
    "ac_sample_rays": ([C.POINTER(ac_field), C.POINTER(ac_render_opts), vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_eikonal_reduce": ([vp, i32, vp, vp], C.c_int),
    "ac_eikonal_reduce2": ([vp, i32, vp, vp], C.c_int),
    "ac_weight_norm_forward": ([C.POINTER(ac_wn_layer), u32, vp], C.c_int),
    "ac_param_grads": ([C.POINTER(ac_pg_entry), u32, vp], C.c_int),
    "ac_sds_upstream": ([vp, vp, u32, f32, vp, vp, vp], C.c_int),
    "ac_render_core_backward_scratch": ([C.POINTER(ac_field), i32, i32], C.c_size_t),
    "ac_render_core_backward": ([C.POINTER(ac_field), C.POINTER(ac_render_opts), vp, vp, vp, C.POINTER(ac_core_saved), C.POINTER(ac_core_upstream),
                                 C.POINTER(ac_core_grads), vp, C.c_size_t, vp], C.c_int),
    "ac_field_sdf": ([C.POINTER(ac_field), vp, u32, f32, vp, vp], C.c_int),
    "ac_field_color": ([C.POINTER(ac_field), vp, vp, vp, u32, vp, vp], C.c_int),
    "ac_mesh_near_far": ([vp, vp, vp, u32, u32, f32, vp, vp, vp], C.c_int),
    "ac_warp_accel_bytes": ([u32], C.c_size_t),
    "ac_warp_accel_build": ([vp, vp, u32, u32, vp, C.c_size_t, vp], C.c_int),
    "ac_warp_samples_accel": ([vp, vp, vp, vp, u32, u32, u32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_hash_stencil_forward": ([vp, vp, vp, vp, u32, u32, u32, f32, u32, f32, f32, vp], C.c_int),
    "ac_hash_stencil_backward_scratch": ([vp, u32, f32, u32, u32, u32], C.c_size_t),
    "ac_hash_stencil_backward": ([vp, vp, vp, vp, u32, u32, u32, f32, u32, f32, f32, vp, C.c_size_t, vp], C.c_int),
    "ac_sdf_stencil_forward": ([C.POINTER(ac_field), vp, u32, f32, f32, vp, vp, vp], C.c_int),
    "ac_sdf_stencil_backward_scratch": ([u32], C.c_size_t),
    "ac_sdf_stencil_backward": ([C.POINTER(ac_field), vp, vp, vp, u32, f32, f32, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "ac_debug_unit_div_check": ([u32, u32, f32, vp, vp], C.c_int),
    "ac_sdf_stencil_backward_inputs": ([C.POINTER(ac_field), vp, vp, vp, u32, f32, f32, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "ac_hash_stencil_input_backward": ([vp, vp, vp, vp, vp, u32, u32, u32, f32, u32, f32, f32, vp], C.c_int),
    "ac_color_forward": ([C.POINTER(ac_field), vp, vp, vp, u32, vp, vp], C.c_int),
    "ac_render_handoff_timeouts": ([vp, vp], C.c_int),
    "ac_warp_accel_work": ([vp, vp, vp], C.c_int),
    "ac_debug_warped_phases": ([C.c_int], None),
    "ac_debug_warped_phase_ms": ([vp], C.c_int),
    "ac_render_rays_occupancy": ([C.POINTER(ac_field), vp, vp, u32, vp, u32, f32, f32, f32, f32, vp, f32, vp, vp, vp, vp, vp, u32, vp], C.c_int),
    "ac_render_rays_occupancy_phased_scratch": ([u32], C.c_size_t),
    "ac_render_rays_occupancy_phased": ([C.POINTER(ac_field), vp, vp, u32, vp, u32, f32, f32, f32, f32, vp, f32, vp, vp, vp, vp, vp, u32, vp, C.c_size_t, vp], C.c_int),
    "ac_render_rays_occupancy_train_scratch": ([u32, u32], C.c_size_t),
    "ac_render_rays_occupancy_train": ([C.POINTER(ac_field), vp, vp, u32, vp, u32, f32, f32, f32, f32, vp, f32, u32, u32, u32, vp, vp, u32, f32, vp, vp, vp, vp,
                                        vp, C.c_size_t, vp], C.c_int),
    "ac_field_samples": ([C.POINTER(ac_field), vp, vp, vp, u32, u32, f32, f32, f32, vp, f32, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_color_backward_scratch": ([u32], C.c_size_t),
    "ac_color_backward": ([C.POINTER(ac_field), vp, vp, vp, vp, u32, vp, vp, vp, vp, C.c_size_t, vp], C.c_int),
    "ac_composite_forward": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_composite_backward": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_render_rays_warped_scratch": ([i32, i32, C.POINTER(C.c_size_t)], C.c_size_t),
    "ac_render_rays_warped": ([C.POINTER(ac_field), C.POINTER(ac_render_opts), vp, vp, vp, vp, vp, vp, C.POINTER(ac_warp_mesh), vp, C.c_size_t,
                               C.POINTER(ac_render_out), vp], C.c_int),
    "ac_warp_samples": ([vp, vp, vp, vp, u32, u32, u32, C.c_double, vp, vp, vp, vp, vp, vp, vp], C.c_int),
    "ac_sh_bias": ([C.POINTER(ac_field), vp, u32, vp, vp, vp], C.c_int),
    "ac_field_color_dirs": ([C.POINTER(ac_field), vp, vp, vp, vp, u32, vp, vp], C.c_int),
    "ac_field_sdf_grid": ([C.POINTER(ac_field), vp, vp, vp, u32, u32, u32, f32, C.c_int, vp, vp], C.c_int),
    "ac_marching_cubes_scratch": ([u32, u32, u32], C.c_size_t),
    "ac_marching_cubes_count": ([vp, u32, u32, u32, f32, vp, C.c_size_t, vp, vp], C.c_int),
    "ac_marching_cubes_emit": ([vp, u32, u32, u32, f32, vp, C.c_size_t, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), vp, u32, vp, u32, vp], C.c_int),
    "ac_density_grid_update_scratch": ([u32], C.c_size_t),
    "ac_density_grid_update": ([C.POINTER(ac_field), vp, u32, f32, f32, f32, vp, vp, vp, C.c_size_t, vp], C.c_int),
}
EXPORTS = tuple(_SIGS)
FIELD_PREPARED_BYTES = 98304          # AC_FIELD_PREPARED_BYTES of include/avatarcraft_hip.h


def lib():
    """The loaded library.  Raises RuntimeError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"avatarcraft_amd: native library {LIB_PATH} is missing. Build it with "
                f"`python -m avatarcraft_amd.build` (hipcc, gfx950). There is no CPU fallback for the hot path.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError(f"avatarcraft_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (args, res) in _SIGS.items():
            fn = getattr(handle, name)      # AttributeError here = ABI mismatch, also loud
            fn.argtypes = args
            fn.restype = res
        if handle.ac_version() != 10:
            raise RuntimeError("avatarcraft_amd: libavatarcraft_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc, what=""):
    """Map a non-zero status to the RuntimeError the reference's TORCH_CHECK / std::runtime_error gives."""
    if rc != 0:
        msg = lib().ac_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}" if what else msg)


def ptr(t):
    """device (or host) address of a torch tensor / None"""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


DTYPE_CODES = {}       # torch dtype -> AC_DTYPE_* (include/avatarcraft_hip.h), filled on first use (torch is imported lazily by the callers)


def dtype_code(t, name="inputs", like=None):
    """AC_DTYPE_F32 / _F16 / _F64 of a tensor; the reference's CHECK_IS_FLOATING (hashencoder.cu:19) for anything else.  like: tensors that must share
    the dtype (the reference instantiates its kernels on ONE scalar_t for every pointer of the call)."""
    import torch
    if not DTYPE_CODES:
        DTYPE_CODES.update({torch.float32: 0, torch.float16: 1, torch.float64: 2})
    code = DTYPE_CODES.get(t.dtype)
    if code is None:
        raise RuntimeError(f"{name} must be a floating tensor")
    for other, oname in (like or ()):
        if other.dtype != t.dtype:
            raise RuntimeError(f"{oname} must have the dtype of {name} ({t.dtype}), got {other.dtype}")
    return code


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("must be a CUDA tensor")      # reference: CHECK_CUDA
        if t is not None and not t.is_contiguous():
            raise RuntimeError("must be a contiguous tensor")  # reference: CHECK_CONTIGUOUS
