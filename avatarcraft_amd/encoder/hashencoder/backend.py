"""`_backend` of the hash encoder: same two functions, argument order and in-place convention as the
reference's pybind module (encoder/hashencoder/src/bindings.cpp:6-7, hashencoder.cu:413-468), served
by libavatarcraft_hip.so.  No JIT on import; raises RuntimeError when the library is absent."""
import numpy as np
import torch

from ... import _lib as L


def _floating(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float32 tensor")


PRIVATE_COPIES = 16         # private copies of the small dense levels in the stencil backward (same-address atomic bursts)
_SCRATCH = {}               # (kind, device, stream, layout) -> uint8 tensor, grown to the largest batch seen (free_scratch() drops them)


BINNED_SCATTER = True       # hashed levels through the binned two-pass scatter (needs ~8 KB of device scratch per sample)


def _scratch_buffer(key, nbytes, device):
    """One buffer per (kind, device, STREAM, table layout): two streams never share queues (concurrent backward passes would race on the
    slot counters), a smaller batch (the last partial batch of an epoch) re-uses the buffer of the largest one instead of re-allocating."""
    key = key + (int(L.current_stream(device) or 0),)
    cur = _SCRATCH.get(key)
    if nbytes and (cur is None or cur.numel() < nbytes):
        _SCRATCH[key] = None                       # release the old one first
        cur = _SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return cur


def free_scratch():
    """release the cached scatter queues (multi-GB for training batches)"""
    _SCRATCH.clear()


def stencil_scratch(offsets_host, L_, S, H, device, B=0):
    """(tensor, nbytes): device scratch for ac_hash_stencil_backward on the current stream; (None, 0) if not needed"""
    B = int(B) if BINNED_SCATTER else 0
    nbytes = int(L.lib().ac_hash_stencil_backward_scratch(offsets_host.ctypes.data, L_, S, H, PRIVATE_COPIES, B))
    if not nbytes:
        return None, 0
    return _scratch_buffer(("stencil", str(device), int(offsets_host[-1]), L_, S, H, bool(B)), nbytes, device), nbytes


class _Backend:
    @staticmethod
    def _offsets_host(offsets):
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be an int tensor")
        cache = getattr(offsets, "_ac_host", None)
        if cache is None:
            cache = np.ascontiguousarray(offsets.detach().cpu().numpy(), dtype=np.int32)
            try:
                offsets._ac_host = cache
            except Exception:
                pass
        return cache

    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L_, S, H, calc_grad_inputs, dy_dx):
        L.require_cuda(inputs, embeddings, offsets, outputs, dy_dx)
        # the reference dispatches on inputs.scalar_type() over Float / Half / Double (hashencoder.cu:352): one dtype for every tensor of the call
        code = L.dtype_code(inputs, "inputs", ((embeddings, "embeddings"), (outputs, "outputs"), (dy_dx, "dy_dx")))
        oh = _Backend._offsets_host(offsets)
        if code != 0:
            L.check(L.lib().ac_hash_encode_forward_typed(code, inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(), oh.ctypes.data,
                                                         outputs.data_ptr(), B, D, C, L_, float(np.float32(S)), H, int(bool(calc_grad_inputs)),
                                                         dy_dx.data_ptr(), L.current_stream(inputs.device)), "hash_encode_forward")
            return
        L.check(L.lib().ac_hash_encode_forward(inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(), oh.ctypes.data,
                                               outputs.data_ptr(), B, D, C, L_, float(np.float32(S)), H, int(bool(calc_grad_inputs)),
                                               dy_dx.data_ptr(), L.current_stream(inputs.device)), "hash_encode_forward")

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L_, S, H, calc_grad_inputs, dy_dx,
                             grad_inputs):
        L.require_cuda(grad, inputs, embeddings, offsets, grad_embeddings, dy_dx, grad_inputs)
        code = L.dtype_code(grad, "grad", ((inputs, "inputs"), (embeddings, "embeddings"), (grad_embeddings, "grad_embeddings"), (dy_dx, "dy_dx"),
                                           (grad_inputs, "grad_inputs")))
        oh = _Backend._offsets_host(offsets)
        Sf = float(np.float32(S))
        if code != 0:           # half / double: hardware atomics (packed half2 / fp64), no binned path
            L.check(L.lib().ac_hash_encode_backward_typed(code, grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(),
                                                          oh.ctypes.data, grad_embeddings.data_ptr(), B, D, C, L_, Sf, H, int(bool(calc_grad_inputs)),
                                                          dy_dx.data_ptr(), grad_inputs.data_ptr(), L.current_stream(inputs.device)),
                    "hash_encode_backward")
            return
        scratch, nbytes = None, 0
        if BINNED_SCATTER and not calc_grad_inputs:
            nbytes = int(L.lib().ac_hash_encode_backward_scratch(oh.ctypes.data, D, C, L_, Sf, H, B))
            scratch = _scratch_buffer(("enc", str(inputs.device), int(oh[-1]), D, C, L_, Sf, H), nbytes, inputs.device) if nbytes else None
        L.check(L.lib().ac_hash_encode_backward_ws(grad.data_ptr(), inputs.data_ptr(), embeddings.data_ptr(), offsets.data_ptr(),
                                                   oh.ctypes.data, grad_embeddings.data_ptr(), B, D, C, L_, Sf, H,
                                                   int(bool(calc_grad_inputs)), dy_dx.data_ptr(), grad_inputs.data_ptr(), L.ptr(scratch), nbytes,
                                                   L.current_stream(inputs.device)), "hash_encode_backward")


    # ---- the 7-point finite-difference stencil in one launch (csrc/hash_stencil.hip); not part of the reference's pybind surface
    @staticmethod
    def hash_stencil_forward(x, embeddings, offsets, outputs, B, C, L_, S, H, eps, bound):
        L.require_cuda(x, embeddings, offsets, outputs)
        for t, n in ((x, "x"), (embeddings, "embeddings"), (outputs, "outputs")):
            _floating(t, n)                 # the stencil operators are not part of the reference's surface: fp32 only
        oh = _Backend._offsets_host(offsets)
        L.check(L.lib().ac_hash_stencil_forward(x.data_ptr(), embeddings.data_ptr(), oh.ctypes.data, outputs.data_ptr(), B, C, L_,
                                                float(np.float32(S)), H, float(eps), float(bound), L.current_stream(x.device)),
                "hash_stencil_forward")

    @staticmethod
    def hash_stencil_backward(grad, x, offsets, grad_embeddings, B, C, L_, S, H, eps, bound):
        L.require_cuda(grad, x, offsets, grad_embeddings)
        oh = _Backend._offsets_host(offsets)
        scratch, nbytes = stencil_scratch(oh, L_, float(np.float32(S)), H, x.device, B)
        L.check(L.lib().ac_hash_stencil_backward(grad.data_ptr(), x.data_ptr(), oh.ctypes.data, grad_embeddings.data_ptr(), B, C, L_,
                                                 float(np.float32(S)), H, float(eps), float(bound), L.ptr(scratch), nbytes,
                                                 L.current_stream(x.device)), "hash_stencil_backward")


_backend = _Backend()
__all__ = ["_backend"]
