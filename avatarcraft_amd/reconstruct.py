"""Fit the bare-SMPL Instant-NSR field from multi-view renders: counterpart of reconstruct.py:29-165 -- the loop that
produces the `bare_smpl` checkpoint stylize.py starts from.  Same kernels as the stylisation step (the training render is the fused
operator nsr_ops.render_core), a different loss:

    per batch of 1600 rays (all views' rays shuffled once per epoch):
        rgb, eikonal = render_instantnsr_naive(net, rays, requires_grad=True, perturb=1.0, render_can=True, white / black background)
        loss = smooth_l1(rgb, rgb_gt, mean) + 0.1 * eikonal                                            reconstruct.py:101-112
        Adam(lr 5e-4, betas (0.9, 0.99), eps 1e-15).step();  CosineAnnealingLR(T_max = epochs, eta_min = lr // 20 (= 0)) per epoch   :47-49,161

`NeusDataset` reads the reference's data/smpl_da_512 layout (transforms_train.json + img/NNNN.png, utils/SMPLDataset.py:11-60); the two
inference drivers only need its camera (avatarcraft_amd.drivers.gen_rays_pose)."""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from .render_utils import render_instantnsr_naive, WHITE_BKG, BLACK_BKG, NSR_BOUND

BATCH_SIZE = 1600                 # reconstruct.py:78
W_EIKONAL = 0.1                   # reconstruct.py:108


def make_optimizer(net, epochs, lr=5e-4):
    """reconstruct.py:47-49 (the scheduler's eta_min = lr // 20 is a float floor division: 0.0)"""
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=epochs, eta_min=lr // 20)
    return opt, sched


def reconstruct_step(net, optimizer, rays_o, rays_d, rgb_gt, white_bkg=True, w_eikonal=W_EIKONAL, batch_size=BATCH_SIZE, num_steps=64, upsample_steps=64,
                     flat_grad=None, process_group=None, grad_divisor=None):
    """one optimisation step on one ray batch (reconstruct.py:92-112).  rays_o, rays_d, rgb_gt: [n, 3] on the net's device.
    With a process group every rank takes its own ray batch and the flat gradient is averaged in ONE collective like stylize.sds_step (stylize.reduce_gradients)."""
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    if flat_grad is None and dist_on and torch.distributed.get_world_size(process_group) > 1:
        # every rank draws its own ray batch: without the all-reduce the replicas would silently drift apart (stylize.sds_step refuses the same way)
        raise RuntimeError("reconstruct_step under torch.distributed (world size > 1) needs flat_grad = stylize.flat_grad_view(net.parameters()): "
                           "the gradients of the ranks are averaged through that buffer")
    from .stylize import reduce_gradients, _check_finite, _collective_verdict
    collective = flat_grad is not None and (process_group is not None or dist_on)
    guard = None
    # (under a process group the render's NaN flag is not judged on this rank alone: it rides in the guard word of the collective -- stylize._collective_verdict)
    with _collective_verdict(collective, net) as verdict:
        with torch.enable_grad():
            rgb, eikonal, _ = render_instantnsr_naive(net, rays_o, rays_d, rays_per_batch=batch_size, requires_grad=True, bkg_key=WHITE_BKG if white_bkg else BLACK_BKG,
                                                      return_torch=True, perturb=1.0, return_raw=True, render_can=True, bound=NSR_BOUND, num_steps=num_steps,
                                                      upsample_steps=upsample_steps)
            if flat_grad is not None:
                flat_grad.zero_()
            else:
                optimizer.zero_grad()
            loss = F.smooth_l1_loss(rgb, rgb_gt, reduction='mean') + eikonal * w_eikonal
            loss.backward()
        if collective:
            guard = reduce_gradients(flat_grad, process_group, grad_divisor, nan_flag=verdict.flag(extra=[eikonal] if isinstance(eikonal, torch.Tensor) else ()))
    # a NaN in this step's normals -- on ANY rank, through the guard word of the collective -- raises here on every rank (reference: the assert at
    # instant_nsr.py:274), not after Adam has consumed it
    _check_finite(net, guard)
    optimizer.step()
    return loss.detach()


def reconstruct_epochs(net, optimizer, scheduler, all_rays_o, all_rays_d, gt_rgb, epochs, batch_size=BATCH_SIZE, white_bkg=True, seed=42, on_step=None,
                       max_steps=None, flat_grad=None, process_group=None):
    """the training loop of main_reconstruct (reconstruct.py:80-162): per epoch one random permutation of ALL rays (every view), batches of
    1600, scheduler.step() per epoch.  all_rays_o / all_rays_d / gt_rgb: [n_views * H * W, 3].  Under torch.distributed every rank draws the
    same permutation and takes the batches rank, rank + world, ...; every batch of an epoch is visited once also when their number is not a multiple of the
    world size (the ranks without a batch in the last round contribute a zero gradient).  Returns the number of optimizer steps taken by this rank."""
    gen = torch.Generator(); gen.manual_seed(seed)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank(process_group) if dist_on else 0
    world = torch.distributed.get_world_size(process_group) if dist_on else 1
    n = all_rays_o.shape[0]
    step = 0
    for epoch in range(epochs):
        perm = torch.randperm(n, generator=gen).to(all_rays_o.device)
        starts = list(range(0, n, batch_size))
        rounds = -(-len(starts) // world)                      # every batch of the epoch exactly once, like the reference (reconstruct.py:86-92): in the last
        for rnd in range(rounds):                              # round the ranks without a batch join the collective with a zero gradient (stylize.sds_idle_step)
            k = rnd * world + rank
            n_active = min(world, len(starts) - rnd * world)
            if k < len(starts):
                idx = perm[starts[k]:starts[k] + batch_size]
                loss = reconstruct_step(net, optimizer, all_rays_o[idx].contiguous(), all_rays_d[idx].contiguous(), gt_rgb[idx], white_bkg=white_bkg,
                                        batch_size=batch_size, flat_grad=flat_grad, process_group=process_group,
                                        grad_divisor=n_active if n_active < world else None)
            else:
                from .stylize import sds_idle_step
                sds_idle_step(net, optimizer, flat_grad, n_active, process_group)
                loss = torch.zeros((), device=all_rays_o.device)
            if on_step is not None:
                on_step(step, epoch, loss)
            step += 1
            if max_steps is not None and step >= max_steps:
                return step
        scheduler.step()
    return step


class NeusDataset:
    """the multi-view set the reference fits (utils/SMPLDataset.py:11-60): `transforms_train.json` (camera_angle_x, frames[file_path,
    transform_matrix]) next to the PNGs.  images [n,H,W,C] in [0,1]; the reference flips them along the WIDTH axis (`images[:, :, ::-1]`,
    :33 -- written as a BGR->RGB swap, applied to axis 2 of [n,H,W,C]); kept, so that a field fitted here matches one fitted there."""

    def __init__(self, path, device="cuda", max_views=None):
        from PIL import Image
        self.device = torch.device(device)
        with open(os.path.join(path, 'transforms_train.json'), 'r') as fp:
            meta = json.load(fp)
        frames = meta['frames'][:max_views] if max_views else meta['frames']
        imgs, poses = [], []
        for fr in frames:
            imgs.append(np.asarray(Image.open(os.path.join(path, fr['file_path'] + '.png'))))
            poses.append(np.array(fr['transform_matrix']))
        images = (np.array(imgs) / 255.).astype(np.float32)[:, :, ::-1]
        self.images = torch.from_numpy(images.copy())
        self.poses = torch.from_numpy(np.array(poses).astype(np.float32)).to(self.device)
        self.n_images = len(imgs)
        self.H, self.W = self.images[0].shape[:2]
        self.camera_angle_x = float(meta['camera_angle_x'])
        self.focal = .5 * self.W / np.tan(.5 * self.camera_angle_x)

    def gen_rays_pose(self, pose, resolution_level=1):
        from .drivers import gen_rays_pose
        return gen_rays_pose(pose, resolution_level, H=self.H, W=self.W, camera_angle_x=self.camera_angle_x, device=self.device)

    def gen_rays_at(self, img_idx, resolution_level=1):
        return self.gen_rays_pose(self.poses[img_idx], resolution_level)

    def all_rays(self):
        """(rays_o, rays_d, rgb) of every pixel of every view, [n*H*W, 3] each, on the device (reconstruct.py:60-72)"""
        ro, rd = zip(*[self.gen_rays_pose(self.poses[i]) for i in range(self.n_images)])
        return (torch.stack(ro).reshape(-1, 3).contiguous(), torch.stack(rd).reshape(-1, 3).contiguous(),
                self.images[..., :3].reshape(-1, 3).to(self.device))
