"""Render harness and camera helpers: counterpart of the parts of utils/render_utils.py that call the hot path
(SURVEY section 8a rows a16, a20): render_instantnsr_naive :514-600, select_background :953-987,
sparse_ray_sampling :989-1011, pose_spherical :57-76, default_360_path :137-154, pose2cap / cap2rays :323-376
(+ cameras/*, geometry/pcd_projector.py:85-120, utils/ray_utils.py:25-37 restated as plain numpy).
Host-side code: the per-ray work is in libavatarcraft_hip.so via NeRFNetwork.render."""
import random

import numpy as np
import torch

from . import nsr_ops

WHITE_BKG, BLACK_BKG, NOISE_BKG, CHESSBOARD_BKG = 0, 1, 2, 3        # utils/constant.py:27-30
CANONICAL_ZOOM_FACTOR = 1000 / 1280                                  # utils/constant.py:9
NSR_BOUND = 1.6                                                      # utils/constant.py:21


# ------------------------------------------------------------------ backgrounds / ray sub-sampling
def select_background(shape, key) -> torch.Tensor:
    """(H*W, 3) background; 4 kinds (render_utils.py:953-987).  The chessboard's Gaussian blur needs torchvision in
    the reference; here it is a separable 5x9 Gaussian with the reference's default sigma range midpoint."""
    key = key % 4
    if key == WHITE_BKG:
        return torch.ones(shape)
    if key == BLACK_BKG:
        return torch.zeros(shape)
    if key == NOISE_BKG:
        bg = torch.nn.init.normal_(torch.ones(shape[0]), mean=0.5, std=0.1).clamp(0, 1)
        return bg[:, None].repeat(1, 3)
    H = W = int(np.sqrt(shape[0]))
    board = torch.zeros([H, W]) + 0.2
    L = max(H // 10, 1)
    i, j = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32), indexing='xy')
    white = ((i // L + j // L) % 2 == 0)
    board[torch.from_numpy(i[white]).long(), torch.from_numpy(j[white]).long()] = 0.8
    def gauss(n, s):
        x = torch.arange(n, dtype=torch.float32) - (n - 1) / 2
        k = torch.exp(-0.5 * (x / s) ** 2); return k / k.sum()
    ky, kx = gauss(5, 1.05), gauss(9, 1.05)
    b = torch.nn.functional.pad(board[None, None], (4, 4, 2, 2), mode="reflect")
    b = torch.nn.functional.conv2d(b, ky.view(1, 1, 5, 1)); b = torch.nn.functional.conv2d(b, kx.view(1, 1, 1, 9))
    return b.reshape(-1, 1).repeat(1, 3)


def sparse_ray_sampling(rays_o: torch.Tensor, rays_d: torch.Tensor, stride: int = 1):
    """every stride-th pixel from a random top-left offset (render_utils.py:989-1011); rays_* are [H, W, 3]"""
    assert stride > 0
    if stride == 1:
        return rays_o, rays_d
    x_off, y_off = random.randint(0, stride - 1), random.randint(0, stride - 1)
    return rays_o[x_off::stride, y_off::stride, ...], rays_d[x_off::stride, y_off::stride, ...]


_CONST_BG = {}


def _background_on(device, shape, key):
    """select_background(...).to(device); the two constant backgrounds are kept on the device (a pageable host-to-device copy per ray
    batch otherwise: ~50 us of an 8 ms training step each)"""
    k = key % 4
    if k in (WHITE_BKG, BLACK_BKG):
        ck = (str(device), tuple(shape), k)
        if ck not in _CONST_BG:
            if len(_CONST_BG) > 64:
                _CONST_BG.clear()
            _CONST_BG[ck] = select_background(shape, k).to(device)
        return _CONST_BG[ck]
    return select_background(shape, key).to(device)


# ------------------------------------------------------------------ the batching harness
def render_instantnsr_naive(net, rays_o, rays_d, rays_per_batch=6400, requires_grad=False, return_torch=True, bkg_key: int = WHITE_BKG,
                            render_can: bool = False, perturb: bool = True, return_raw: bool = False, verts=None, faces=None, Ts=None,
                            num_steps: int = 64, upsample_steps=64, bound: float = 1.6, opacity_only: bool = False):
    """Same signature, defaults and outputs as the reference (render_utils.py:514-600):
    returns (rgb[Ntot,3], sum of eikonal terms[, {depth[Ntot,1], weight_sum[Ntot,1], normal[Ntot,3]}]).
    opacity_only (not in the reference): the caller will only read weight_sum / depth / normal -- a no-grad render then skips the colour network."""
    device = rays_o.device
    total = rays_o.shape[0]
    rgbs, depths, wsums, normals = [], [], [], []
    total_eikonal = 0.0
    if not render_can and verts is not None and not isinstance(verts, nsr_ops.WarpMesh):
        verts = nsr_ops.WarpMesh(verts, faces, Ts, device)       # upload the frame's mesh once, not once per ray batch
    # temporal seeds of the closest-face searches (round 6): the net keeps, per ray and sample slot of a VIEW, the face the previous frame's search found;
    # each batch of this frame starts its searches from those faces' distances and leaves its own answers for the next frame.  Same pixels bit for bit
    # (ac_warp_mesh.seed_faces); what it buys is in bench.py's posed_frame leg.  net.warp_temporal_seeds = False switches it off.
    seed_rows = None
    if (not render_can and isinstance(verts, nsr_ops.WarpMesh) and not requires_grad and getattr(net, "warp_temporal_seeds", False)
            and verts.accel is not None and rays_o.is_cuda):
        cache = net.__dict__.setdefault("_warp_seed_rows", {})
        key = (str(device), int(total), int(num_steps), int(upsample_steps), int(verts.faces.shape[0]))
        seed_rows = cache.get(key)
        if seed_rows is None:
            cache.clear()                                        # (one view shape at a time: 25 MB per 256 x 256 view of 32 + 64 slots)
            seed_rows = cache[key] = nsr_ops.WarpMesh.new_seed_buffer(total, 2 * num_steps + upsample_steps, device)
    # the harness keeps rgb / depth / weight_sum / normal only: a no-grad render of this package's NeRFNetwork skips the per-sample outputs
    lean = {"per_sample": False} if (not requires_grad and getattr(net, "supports_lean_render", False)) else {}
    if opacity_only and not requires_grad and getattr(net, "supports_opacity_only", False):
        lean["opacity_only"] = True
    if (getattr(net, "cuda_ray", False) and not requires_grad and not net.training and render_can and total > rays_per_batch
            and not getattr(net, "occupancy_rounds", True) and (bkg_key % 4) in (WHITE_BKG, BLACK_BKG)):
        # An occupancy-grid net in eval(): its render is ONE launch whatever the ray count (ac_render_rays_occupancy keeps per-ray state in registers
        # and allocates nothing per sample), and a launch costs the latency of its longest ray -- a chain of ~200 dependent grid look-ups, 0.26 ms --
        # however few rays it holds: sixteen 4096-ray launches per 256 x 256 view took 12.1 ms, the view in one launch 2.4 ms (round 4's figures).
        # The reference batches to bound the memory of ITS per-sample tensors; rays are independent, so the pixels are the same bit for bit.
        rays_per_batch = total
    with torch.set_grad_enabled(requires_grad):
        for i in range(0, total, rays_per_batch):
            ro, rd = rays_o[i:i + rays_per_batch], rays_d[i:i + rays_per_batch]
            background_rgb = _background_on(device, ro.shape, bkg_key)
            if seed_rows is not None:
                verts.bind_seeds(seed_rows[i:i + rays_per_batch])
            out = net.render(ro.unsqueeze(0), rd.unsqueeze(0), num_steps=num_steps, upsample_steps=upsample_steps, bound=bound, staged=False,
                             bg_color=background_rgb, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=render_can, verts=verts,
                             faces=faces, Ts=Ts, perturb=perturb, **lean)
            total_eikonal = total_eikonal + out["gradient_error"]
            rgbs.append(out['rgb']); wsums.append(out['weight_sum']); depths.append(out['depth']); normals.append(out['normal'])
        if seed_rows is not None:
            verts.bind_seeds(None)
        cat = lambda ts, dim=0: ts[0] if len(ts) == 1 else torch.cat(ts, dim=dim)      # one batch (every training patch): nothing to copy
        rgb = cat(rgbs, 1).squeeze(0).reshape(-1, 3)
        extra = {"depth": cat(depths, 1).squeeze(0).reshape(-1, 1), "weight_sum": cat(wsums).reshape(-1, 1), "normal": cat(normals).reshape(-1, 3)}
    if not return_torch:
        rgb = rgb.detach().cpu().numpy()
        extra = {k: v.detach().cpu().numpy() for k, v in extra.items()}
    if return_raw:
        return rgb, total_eikonal, extra
    return rgb, total_eikonal


# ------------------------------------------------------------------ cameras (numpy, fp64 like the reference)
def _trans_t(t):
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], dtype=np.float64)


def _rot_phi(phi):
    return np.array([[1, 0, 0, 0], [0, np.cos(phi), -np.sin(phi), 0], [0, np.sin(phi), np.cos(phi), 0], [0, 0, 0, 1]], dtype=np.float64)


def _rot_theta(th):
    return np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=np.float64)


class CameraPose:
    """camera-to-world pose; the reference stores world-to-camera as (translation f32, rotation) and inverts on demand
    (cameras/camera_pose.py:15-116).  The fp32 round trip of the translation is reproduced."""

    def __init__(self, c2w):
        w2c = np.linalg.inv(np.asarray(c2w, dtype=np.float64))
        w2c /= w2c[3, 3]
        t = w2c[:3, 3].astype(np.float32).astype(np.float64)        # Translation(vec.astype(float32))
        R = w2c[:3, :3]                                              # UnstableRotation keeps the matrix
        self._w2c = np.eye(4); self._w2c[:3, :3] = R; self._w2c[:3, 3] = t

    @property
    def world_to_camera(self):
        return self._w2c

    @property
    def camera_to_world(self):
        M = np.linalg.inv(self._w2c); return M / M[3, 3]

    @property
    def camera_center_in_world(self):
        return self.camera_to_world[:3, 3]


def pose_spherical(theta, phi, radius, add_noise=False, noise_scale=1.0):
    """render_utils.py:57-76; add_noise: the training-time camera jitter (closer by U(0, 0.2), elevation U(-15, 15) deg, azimuth
    N(0, 1) deg, all times noise_scale), drawn from numpy's global stream in the reference's order"""
    if add_noise:
        radius += np.random.uniform(-0.2, 0) * noise_scale
        phi += np.random.uniform(-15, 15) * noise_scale
        theta += np.random.normal(0, 1) * noise_scale
    c2w = _trans_t(radius)
    c2w = _rot_phi(phi / 180. * np.pi) @ c2w
    c2w = _rot_theta(theta / 180. * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float64) @ c2w
    c2w = c2w @ np.diag([1.0, -1.0, -1.0, 1.0])
    return CameraPose(c2w)


def _rotation_matrix(angle, direction):
    """Rodrigues rotation about `direction` (what geometry/transformations.rotation_matrix returns), 4x4"""
    d = np.asarray(direction, dtype=np.float64)
    n = np.linalg.norm(d)
    R = np.eye(4)
    if n == 0 or angle == 0:
        return R
    d = d / n
    s, c = np.sin(angle), np.cos(angle)
    K = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    R[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(d, d)
    return R


def describe_view(angles, body_part: str = "body"):
    """text prefix per view for the prompt augmentation (render_utils.py:80-90; the camera ring starts behind the avatar)"""
    out = []
    for a in angles:
        where = "front" if (-180 <= a <= -150 or 150 <= a <= 180) else ("back" if -30 <= a <= 30 else "side")
        out.append(f"{where} view of the {body_part} of the")
    return out


def _ring_frame(center, up):
    up = np.asarray(up, dtype=np.float64); up2 = np.array([0, 0, 1.0])
    axis = np.cross(up, up2)
    angle = np.arccos(np.clip(np.dot(up, up2) / (np.linalg.norm(up) * np.linalg.norm(up2)), -1, 1))
    trans = np.eye(4); trans[:3, 3] = np.asarray(center, dtype=np.float64)
    return trans @ _rotation_matrix(-angle, axis)


def default_360_path(center, up, dist, res=40, rad=360, add_noise=False):
    """camera ring around `center` (render_utils.py:137-154) -> (poses, angles)"""
    frame = _ring_frame(center, up)
    angles = np.linspace(-rad / 2, rad / 2, res + 1)[:-1]
    poses = [pose_spherical(a, 0, dist, add_noise, noise_scale=1.0) for a in angles]
    return [CameraPose(frame @ p.camera_to_world) for p in poses], angles


def style_360_path(center, up, dist, res=40, rad=360, add_noise=False, noise_scale=1.0, style_head=False, head_offset=0.0, body_part: str = "body",
                   head_rate=0.0, head_dist=0.5):
    """training views (render_utils.py:157-208): front and back sectors only (res//4 in [-180,-120], res//4 in [120,180], res//2 in
    [-60,60]) and, with style_head, int(res * head_rate) close-ups of the head (always jittered) -> (poses, descriptions)"""
    frame = _ring_frame(center, up)
    angles = np.concatenate([np.linspace(-180, -120, res // 4), np.linspace(120, 180, res // 4), np.linspace(-60, 60, res // 2)])
    poses = [CameraPose(frame @ pose_spherical(a, 0, dist, add_noise, noise_scale).camera_to_world) for a in angles]
    desc = describe_view(angles, body_part)
    if style_head and head_rate > 0.0:
        n = int(res * head_rate)
        hframe = _ring_frame(np.asarray(center, dtype=np.float64) + np.asarray(up, dtype=np.float64) * head_offset, up)
        hangles = np.concatenate([np.linspace(-180, -120, n // 2), np.linspace(120, 180, n // 2)])
        poses = poses + [CameraPose(hframe @ pose_spherical(a, 0, head_dist, True, 1.0).camera_to_world) for a in hangles]
        desc = desc + describe_view(hangles, "face")
    return poses, desc


class PinholeCapture:
    def __init__(self, width, height, fx, fy, cx, cy, pose):
        self.width, self.height, self.fx, self.fy, self.cx, self.cy, self.cam_pose = int(width), int(height), fx, fy, cx, cy, pose

    @property
    def shape(self):
        return (self.height, self.width)

    @property
    def intrinsic_matrix(self):
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])


def pose2cap(hw, pose):
    """render_utils.py:323-337: fx = fy = 0.78125 * w, principal point at the image centre"""
    h, w = hw
    return PinholeCapture(w, h, CANONICAL_ZOOM_FACTOR * w, CANONICAL_ZOOM_FACTOR * w, w / 2.0, h / 2.0, pose)


def shot_rays(cap, xys):
    """unproject pixels at depth 1 and normalise (utils/ray_utils.py:25-37, geometry/pcd_projector.py:85-120)"""
    xys = np.asarray(xys)
    xyz = np.stack([xys[:, 0], xys[:, 1], np.ones(xys.shape[0])], axis=1).astype(np.float64)
    xyz = (np.linalg.inv(cap.intrinsic_matrix) @ xyz.T).T
    c2w = cap.cam_pose.camera_to_world
    xyzw = (c2w @ np.concatenate([xyz, np.ones_like(xyz[:, :1])], axis=1).T).T
    xyzw /= xyzw[:, 3:4]
    pcd = xyzw[:, :3].astype(np.float32)
    orig = np.stack([cap.cam_pose.camera_center_in_world] * xys.shape[0])
    d = pcd - orig
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return orig, d


def cap2rays(cap, device="cuda"):
    """(origins[H*W,3], dirs[H*W,3]) float32 on `device` (render_utils.py:363-376).  On a GPU the rays are generated there (the
    same fp64 arithmetic as shot_rays, elementwise): the numpy path costs ~10 ms per 256x256 view, as much as a whole SDS step."""
    dev = torch.device(device)
    if dev.type != "cuda":
        coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
        o, d = shot_rays(cap, coords)
        return torch.from_numpy(o.astype(np.float32)).to(device), torch.from_numpy(d.astype(np.float32)).to(device)
    h, w = cap.shape
    Ki = torch.from_numpy(np.linalg.inv(cap.intrinsic_matrix)).to(dev)
    c2w = torch.from_numpy(np.ascontiguousarray(cap.cam_pose.camera_to_world, dtype=np.float64)).to(dev)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float64), torch.arange(w, device=dev, dtype=torch.float64), indexing="ij")
    x, y = xs.reshape(-1), ys.reshape(-1)                              # row-major pixels, (x, y) = (column, row) like np.argwhere(...)[:, ::-1]
    cam = [Ki[r, 0] * x + Ki[r, 1] * y + Ki[r, 2] for r in range(3)]
    wld = [c2w[r, 0] * cam[0] + c2w[r, 1] * cam[1] + c2w[r, 2] * cam[2] + c2w[r, 3] for r in range(4)]
    pcd = torch.stack([wld[0] / wld[3], wld[1] / wld[3], wld[2] / wld[3]], dim=1).float()
    orig = torch.from_numpy(np.asarray(cap.cam_pose.camera_center_in_world, dtype=np.float64)).to(dev)
    d = pcd.double() - orig[None]
    d = d / torch.linalg.norm(d, dim=1, keepdim=True)
    return orig[None].expand(h * w, 3).float().contiguous(), d.float().contiguous()
