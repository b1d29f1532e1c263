"""The two inference drivers of the reference as functions (no file I/O, no option parsing): the main loops of render_canonical.py:38-99
(360-degree views of the canonical avatar, body and head rings) and render_warp.py:40-125 (SMPL-driven animation / shape
interpolation through the posed-space renderer), plus the camera of the latter, SMPLDataset.gen_rays_pose (utils/SMPLDataset.py:86-103)."""
import numpy as np
import torch

from .render_utils import (default_360_path, pose2cap, cap2rays, render_instantnsr_naive, WHITE_BKG, BLACK_BKG, NSR_BOUND)
from .smpl import calc_local_trans

CANONICAL_CAMERA_DIST_VAL = 1.7         # render_canonical.py:35 (overrides utils/constant.py for the supplementary video)
CAN_HEAD_OFFSET = 0.47 * 0.9            # utils/constant.py:35,43
CAN_HEAD_CAMERA_DIST = 0.5 * 0.9        # utils/constant.py:36,42


def gen_rays_pose(pose, resolution_level=1, H=512, W=512, camera_angle_x=np.pi / 3, device="cuda"):
    """rays of the 512x512 dataset camera with field of view camera_angle_x (focal = 0.5 W / tan(0.5 fov)) for a camera-to-world
    `pose` [4,4], sub-sampled on linspace(0, W-1, W // level): -> rays_o, rays_d [H//level, W//level, 3] fp32"""
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    pose = torch.as_tensor(np.asarray(pose, dtype=np.float32) if not isinstance(pose, torch.Tensor) else pose).to(device=device, dtype=torch.float32)
    l = resolution_level
    tx = torch.linspace(0, W - 1, int(W // l))
    ty = torch.linspace(0, H - 1, int(H // l))
    px, py = torch.meshgrid(tx, ty, indexing="ij")
    px, py = px.t().to(device), py.t().to(device)
    K = torch.tensor([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=torch.float64)
    p = torch.stack([(px - K[0][2]) / K[0][0], -(py - K[1][2]) / K[1][1], -torch.ones_like(px)], -1).float()
    v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    v = torch.sum(v[..., None, :] * pose[:3, :3], -1)
    o = pose[None, None, :3, 3].expand(v.shape)
    return o, v


def _rank_world(rank, world):
    """(rank, world) of this process: explicit arguments win, else the default process group, else (0, 1)"""
    if rank is None or world is None:
        on = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank = (torch.distributed.get_rank() if on else 0) if rank is None else rank
        world = (torch.distributed.get_world_size() if on else 1) if world is None else world
    rank, world = int(rank), int(world)
    if not (world >= 1 and 0 <= rank < world):
        raise ValueError(f"rank {rank} / world {world}: need 0 <= rank < world")
    return rank, world


def shard_indices(n, rank=None, world=None):
    """Multi-GPU inference ("partition the views across ranks", every rank a full replica of the 49 MB table + MLPs, no collective on
    the data path): the view / frame indices rank `rank` of `world` renders -- round robin (rank, rank + world, ...), so that consecutive frames of
    an animation land on different GPUs and the ranks finish together whatever n % world is.  Defaults: the default process group, else everything."""
    rank, world = _rank_world(rank, world)
    return list(range(rank, int(n), world))


def render_canonical_360(net, n_views=100, render_hw=(256, 256), center=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), white_bkg=True, with_head=True,
                         rays_per_batch=4096, device="cuda", rank=None, world=None):
    """yields (ring name, view index, rgb [H,W,3] float32 in [0,1], depth [H,W]) for the body ring and, with_head, the head ring.
    rank / world: this process renders the views shard_indices(n_views, rank, world) of every ring (default: its rank in the default process group;
    without one, all views) -- the view index yielded is the GLOBAL one, so the ranks' outputs interleave into the reference's file sequence."""
    center, up = np.asarray(center, dtype=np.float64), np.asarray(up, dtype=np.float64)
    mine = set(shard_indices(n_views, rank, world))
    rings = [("body", default_360_path(center, up, CANONICAL_CAMERA_DIST_VAL, n_views)[0])]
    if with_head:
        rings.append(("head", default_360_path(center + up * CAN_HEAD_OFFSET, up, CAN_HEAD_CAMERA_DIST, n_views)[0]))
    h, w = render_hw
    for name, poses in rings:
        for i, pose in enumerate(poses):
            if i not in mine:
                continue
            ro, rd = cap2rays(pose2cap([h, w], pose), device=device)
            rgb, _, extra = render_instantnsr_naive(net, ro, rd, rays_per_batch, requires_grad=False, bkg_key=WHITE_BKG if white_bkg else BLACK_BKG,
                                                    return_torch=True, perturb=False, return_raw=True, render_can=True)
            yield name, i, rgb.reshape(h, w, 3), extra["depth"].reshape(h, w)


def render_animation(net, body_model, cam_pose, poses=None, render_type="animate", shape_from=None, shape_to=None, resolution=256, max_frames=100,
                     white_bkg=True, rays_per_batch=None, device="cuda", rank=None, world=None):
    """yields (frame index, rgb [res,res,3]) for an SMPL pose sequence (render_type "animate", poses [F,72]) or a shape interpolation
    ("interp_shape", shape_from / shape_to [1,10]), seen from the dataset camera `cam_pose` [4,4]; 32 + 32 samples per ray like the reference.
    rays_per_batch: the reference cuts a frame into 64 * 128 = 8192-ray batches (render_warp.py) to bound its memory; the default here is the whole
    frame in one batch (0.13 GB of scratch at 256 x 256): same pixels, and the launches are full when the body covers a fraction of the image.
    rank / world: this process renders the frames shard_indices(n_frames, rank, world) (default: its rank in the default process group; without one,
    every frame); the frame index yielded is the global one.  The SMPL forward of the sequence (calc_local_trans: a few ms) runs on every rank."""
    if rays_per_batch is None:
        rays_per_batch = resolution * resolution
    world_verts, Ts, n_frames = calc_local_trans(body_model, render_type=render_type, poses=poses, shape_from=shape_from, shape_to=shape_to,
                                                 max_frames=max_frames)
    faces = np.asarray(body_model.faces)
    ro, rd = gen_rays_pose(cam_pose, int(512 / resolution), device=device)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    mine = list(shard_indices(n_frames, rank, world))
    # one WarpMesh per frame, the next frame's upload + culling structure prepared beside the current frame's render (nsr_ops.warp_mesh_sequence)
    if torch.device(device).type == "cuda":
        from . import nsr_ops
        meshes = nsr_ops.warp_mesh_sequence(((world_verts[i], Ts[i]) for i in mine), faces, device)
    else:
        meshes = (None for _ in mine)
    for i, mesh in zip(mine, meshes):
        # the loop keeps rgb only: samples the warp masks out (alpha * 0) need no field evaluation (bit-identical pixels).  The switch is set around
        # each frame's render and restored (also when the consumer abandons the generator): the caller's net keeps its documented default
        prev = getattr(net, "skip_masked_samples", None)
        if prev is not None:
            net.skip_masked_samples = True
        try:
            rgb, _, _ = render_instantnsr_naive(net, ro, rd, rays_per_batch, requires_grad=False, bkg_key=WHITE_BKG if white_bkg else BLACK_BKG,
                                                return_torch=True, perturb=False, return_raw=True, render_can=False,
                                                verts=mesh if mesh is not None else world_verts[i], faces=faces, Ts=Ts[i], num_steps=32, upsample_steps=32,
                                                bound=NSR_BOUND)
        finally:
            if prev is not None:
                net.skip_masked_samples = prev
        yield i, rgb.reshape(resolution, resolution, 3)
