"""`_backend` of the raymarching operators: the six functions of the reference's pybind module
(raymarching/src/bindings.cpp:5-10, raymarching.h:8-17), same argument order, in place on
caller-allocated tensors, served by libavatarcraft_hip.so."""
import torch

from .. import _lib as L


def _scratch(n, device):
    return torch.empty(n, dtype=torch.int32, device=device)


class _Backend:
    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, mean_density, iter_density, bound, N, H, M, xyzs, dirs, deltas, rays, counter, perturb):
        L.require_cuda(rays_o, rays_d, grid)
        scratch = _scratch(int(L.lib().ac_march_rays_train_scratch(N)), rays_o.device)
        L.check(L.lib().ac_march_rays_train(rays_o.data_ptr(), rays_d.data_ptr(), grid.data_ptr(), float(mean_density),
                                            int(iter_density), float(bound), N, H, M, xyzs.data_ptr(), dirs.data_ptr(),
                                            deltas.data_ptr(), rays.data_ptr(), counter.data_ptr(), int(perturb),
                                            scratch.data_ptr(), L.current_stream(rays_o.device)), "march_rays_train")

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, bound, M, N, weights_sum, image):
        L.require_cuda(sigmas, rgbs, deltas, rays, weights_sum, image)
        if rays.dtype != torch.int32:
            raise RuntimeError("rays must be an int tensor")
        L.check(L.lib().ac_composite_rays_train_forward(sigmas.data_ptr(), rgbs.data_ptr(), deltas.data_ptr(), rays.data_ptr(),
                                                        float(bound), M, N, weights_sum.data_ptr(), image.data_ptr(),
                                                        L.current_stream(sigmas.device)), "composite_rays_train_forward")

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad, sigmas, rgbs, deltas, rays, weights_sum, image, bound, M, N,
                                      grad_sigmas, grad_rgbs):
        L.require_cuda(grad_weights_sum, grad, sigmas, rgbs, deltas, rays, weights_sum, image, grad_sigmas, grad_rgbs)
        L.check(L.lib().ac_composite_rays_train_backward(grad_weights_sum.data_ptr(), grad.data_ptr(), sigmas.data_ptr(),
                                                         rgbs.data_ptr(), deltas.data_ptr(), rays.data_ptr(), weights_sum.data_ptr(),
                                                         image.data_ptr(), float(bound), M, N, grad_sigmas.data_ptr(),
                                                         grad_rgbs.data_ptr(), L.current_stream(sigmas.device)),
                "composite_rays_train_backward")

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, H, density_grid, mean_density, near, far, xyzs, dirs,
                   deltas, perturb):
        L.check(L.lib().ac_march_rays(n_alive, n_step, rays_alive.data_ptr(), rays_t.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(),
                                      float(bound), H, density_grid.data_ptr(), float(mean_density), near.data_ptr(), far.data_ptr(),
                                      xyzs.data_ptr(), dirs.data_ptr(), deltas.data_ptr(), int(perturb),
                                      L.current_stream(rays_o.device)), "march_rays")

    @staticmethod
    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, normals, deltas, weights, depth, image, normal_map):
        L.check(L.lib().ac_composite_rays(n_alive, n_step, rays_alive.data_ptr(), rays_t.data_ptr(), sigmas.data_ptr(), rgbs.data_ptr(),
                                          normals.data_ptr(), deltas.data_ptr(), weights.data_ptr(), depth.data_ptr(), image.data_ptr(),
                                          normal_map.data_ptr(), L.current_stream(sigmas.device)), "composite_rays")

    @staticmethod
    def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
        scratch = _scratch(n_alive + 2, rays_t.device)
        L.check(L.lib().ac_compact_rays(n_alive, rays_alive.data_ptr(), rays_alive_old.data_ptr(), rays_t.data_ptr(),
                                        rays_t_old.data_ptr(), alive_counter.data_ptr(), scratch.data_ptr(),
                                        L.current_stream(rays_t.device)), "compact_rays")


_backend = _Backend()
__all__ = ["_backend"]
