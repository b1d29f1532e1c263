"""One rank of tests/test_gpu_multi.py (BASELINE config 5: data-parallel stylisation, one view per rank, ONE all-reduce of the flat gradient).

    python tests/dp_worker.py <backend> <outdir>          with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment

Every rank, on its own GPU (rank % device_count: with `gloo` on a 1-GPU box the ranks share the device -- a plumbing check of the same code):
  1. BEFORE the process group exists: the REAL sds_step on this rank's view with a stand-in optimizer that does not step -> this rank's own
     gradient g_r (no collective anywhere);
  2. process group (nccl = RCCL, one rank per GPU), the same step again from the same state and random streams with the real optimizer
     (stylize.Adam) -> the parameters after one data-parallel step, and the averaged flat gradient as the collective left it;
  3. all ranks' g_r are gathered (an all_gather of the test, not of the product) and written down with the results.
The parent test forms the manual average of the g_r, applies one Adam step to the initial parameters and compares."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _NoStep:
    def __init__(self, opt):
        self.param_groups = opt.param_groups

    def zero_grad(self, set_to_none=False):
        pass

    def step(self):
        pass


def main():
    backend, outdir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view, Adam
    from avatarcraft_amd.synthetic import make_rays
    import tests.test_gpu_model as TM
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    TM.DEV = str(dev)
    # one view per rank (pose indices 0, 12, ... on the 100-view ring -> yaw steps of 12 * 3.6 degrees), 32 x 32 rays
    ro, rd = make_rays(32, 32, dist=1.8, f=25.0, yaw=float(np.deg2rad(12 * 3.6 * rank)))
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)

    def fresh():
        net, _ = TM.golden_net(train=True)
        net_gt, _ = TM.golden_net(train=False)
        opt = Adam(net.parameters(), lr=5e-3, zero_grad_in_step=False)
        flat = flat_grad_view(net.parameters())
        torch.manual_seed(42 + rank)                       # per-rank streams for jitter noise, identical weights
        return net, net_gt, opt, flat, SyntheticGuidance(42 + rank)

    net, net_gt, opt, flat, guide = fresh()
    init = {k: v.detach().clone() for k, v in net.named_parameters()}
    assert not dist.is_initialized()
    sds_step(net, net_gt, ro, rd, (32, 32), _NoStep(opt), guide, batch_size=4096, flat_grad=flat)
    torch.cuda.synchronize()
    g_own = flat.detach().clone()
    for k, v in net.named_parameters():
        assert torch.equal(v.detach(), init[k]), k         # the stand-in did not step

    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        net, net_gt, opt, flat, guide = fresh()
        marks = []
        sds_step(net, net_gt, ro, rd, (32, 32), opt, guide, batch_size=4096, flat_grad=flat, timers=marks)
        torch.cuda.synchronize()
        g_avg = flat.detach().clone()
        after = {k: v.detach().clone() for k, v in net.named_parameters()}
        gathered = [torch.empty_like(g_own) for _ in range(world)]
        dist.all_gather(gathered, g_own)
        torch.cuda.synchronize()
        if rank == 0:
            np.savez(os.path.join(outdir, "rank0.npz"), g_all=torch.stack(gathered).cpu().numpy(), g_avg=g_avg.cpu().numpy(),
                     names=np.array(list(after.keys())), marks=np.array([n for n, _ in marks]),
                     **{"init." + k: v.cpu().numpy() for k, v in init.items()}, **{"after." + k: v.cpu().numpy() for k, v in after.items()})
        else:
            np.savez(os.path.join(outdir, f"rank{rank}.npz"), g_avg=g_avg.cpu().numpy(), **{"after." + k: v.cpu().numpy() for k, v in after.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
