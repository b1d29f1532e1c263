"""Occupancy path timing probe (GPU): the training form (one launch vs the chain of operators) on the SDS step's 4096-ray view and the inference launch on
the bench view in 4096-ray batches / one piece -- HIP-event times of the calls, for A/B runs over library variants (AC_LIB_PATH, e.g. -DAC_RM_BATCH=16).
    python tools/occ_train_probe.py            (under gpurun; prints one JSON line)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    p, field, table, ro, rd = bench.make_inputs(dev, 0)
    from avatarcraft_amd.render_utils import NSR_BOUND
    net = bench.make_net(p, table, dev, False, cuda_ray=True)
    with torch.no_grad():
        net.deviation_net.variance.fill_(float(np.log(512.0) / 10.0))
        net.update_extra_state(NSR_BOUND)
    kw = dict(num_steps=64, bound=NSR_BOUND, upsample_steps=64, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0)
    ro_t, rd_t = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    so, sd = bench.sds_view(0)
    so, sd = torch.from_numpy(so).to(dev), torch.from_numpy(sd).to(dev)
    res = {"lib": os.environ.get("AC_LIB_PATH", "product")}

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        t0 = time.perf_counter()
        for s, e in evs:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        return {"gpu_ms": float(np.median([s.elapsed_time(e) for s, e in evs])), "wall_ms": wall}
    with torch.no_grad():
        net.eval()
        res["eval_4096"] = timed(lambda: net.render(ro_t[None, :4096], rd_t[None, :4096], **kw), 20)
        res["eval_view"] = timed(lambda: net.render(ro_t[None], rd_t[None], **kw), 10)
        from avatarcraft_amd import nsr_ops
        fa = (net._field(), ro_t, rd_t, net.density_grid, net.mean_density, NSR_BOUND, 0.005, net.forward_variance(), 1.0)
        res["eval_view_groups"] = timed(lambda: nsr_ops.render_rays_occupancy(*fa, phased=False), 10)
        res["eval_view_phased"] = timed(lambda: nsr_ops.render_rays_occupancy(*fa, phased=True), 10)
        fb = (net._field(), ro_t[:4096], rd_t[:4096], net.density_grid, net.mean_density, NSR_BOUND, 0.005, net.forward_variance(), 1.0)
        res["eval_4096_phased"] = timed(lambda: nsr_ops.render_rays_occupancy(*fb, phased=True), 20)
        mid = slice(30000, 34096)
        fc = (net._field(), ro_t[mid], rd_t[mid], net.density_grid, net.mean_density, NSR_BOUND, 0.005, net.forward_variance(), 1.0)
        res["eval_mid4096_groups"] = timed(lambda: nsr_ops.render_rays_occupancy(*fc, phased=False), 20)
        res["eval_mid4096_phased"] = timed(lambda: nsr_ops.render_rays_occupancy(*fc, phased=True), 20)
        res["view_samples"] = {k: int(nsr_ops.render_rays_occupancy(*fa, phased=ph, count_samples=True)["n_samples"]) for k, ph in (("groups", False), ("phased", True))}
        net.train()
        net.mean_count = 0
        net.render(so[None, :4096], sd[None, :4096], perturb=True, **kw)
        samples = int(net.step_counter[(net.local_step - 1) % 64, 0].item())
        net.mean_count = samples + 4096
        res["samples"] = samples
        for one in (True, False):
            net.occupancy_train_one_launch = one
            res["train_one_launch" if one else "train_chain"] = timed(lambda: net.render(so[None, :4096], sd[None, :4096], perturb=True, **kw), 30)
        net.occupancy_train_one_launch = True
        for glog in os.environ.get("PROBE_GLOGS", "").split():
            pass
    print(json.dumps(res))


if __name__ == "__main__":
    main()
