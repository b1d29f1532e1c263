"""`_backend` of the SH encoder (reference encoder/shencoder/src/bindings.cpp, shencoder.cu:403-441)."""
import torch

from ... import _lib as L


class _Backend:
    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
        L.require_cuda(inputs, outputs, dy_dx)
        # Float / Half / Double like shencoder.cu:337 (AT_DISPATCH_FLOATING_TYPES_AND_HALF over inputs.scalar_type())
        code = L.dtype_code(inputs, "inputs", ((outputs, "outputs"), (dy_dx, "dy_dx")))
        L.check(L.lib().ac_sh_encode_forward_typed(code, inputs.data_ptr(), outputs.data_ptr(), B, D, C, int(bool(calc_grad_inputs)),
                                                   dy_dx.data_ptr(), L.current_stream(inputs.device)), "sh_encode_forward")

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        L.require_cuda(grad, inputs, dy_dx, grad_inputs)
        code = L.dtype_code(grad, "grad", ((inputs, "inputs"), (dy_dx, "dy_dx"), (grad_inputs, "grad_inputs")))
        L.check(L.lib().ac_sh_encode_backward_typed(code, grad.data_ptr(), inputs.data_ptr(), B, D, C, dy_dx.data_ptr(), grad_inputs.data_ptr(),
                                                    L.current_stream(inputs.device)), "sh_encode_backward")


_backend = _Backend()
__all__ = ["_backend"]
