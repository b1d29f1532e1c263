"""Vanilla NeRF plumbing of BASELINE config 1: counterpart of models/nerf.py:96-153 (NeRF MLP),
utils/ray_utils.py:169-208 (ray_to_samples) and utils/render_utils.py:213-249 (raw2outputs).

Pure PyTorch in the reference and here (it is the reference's CPU-runnable plumbing case, not a kernel): 64x64 rays x 16 samples,
frequency encoding (encoder.freq_encoder), 8x256 MLP, alpha = 1 - exp(-relu(sigma) * delta * |d|), cumprod compositing, white
background.  Parameter names equal the reference's (pts_linears.N, views_linears.0, feature_linear, alpha_linear, rgb_linear,
output_linear), so its checkpoints load."""
import torch
import torch.nn as nn
import torch.nn.functional as F

PERTURB_EPSILON = 0.01          # utils/constant.py:18


class NeRF(nn.Module):
    def __init__(self, depth=8, width=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False, scale=1.0, scale_type='no'):
        super().__init__()
        self.depth, self.width, self.input_ch, self.input_ch_views = depth, width, input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.scale, self.scale_type = skips, use_viewdirs, scale, scale_type
        layers = [nn.Linear(input_ch, width)]
        for i in range(depth - 1):          # the layer AFTER a skip index takes the re-injected input as well
            layers.append(nn.Linear(width + input_ch, width) if i in skips else nn.Linear(width, width))
        self.pts_linears = nn.ModuleList(layers)
        if use_viewdirs:
            self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + width, width // 2)])
            self.feature_linear = nn.Linear(width, width)
            self.alpha_linear = nn.Linear(width, 1)
            self.rgb_linear = nn.Linear(width // 2, 3)
        else:
            self.output_linear = nn.Linear(width, output_ch)

    def forward(self, input_pts, input_views=None):
        assert input_pts.shape[-1] == self.input_ch
        if not self.use_viewdirs:
            input_views = None
        if input_views is not None:
            assert input_views.shape[-1] == self.input_ch_views
        h = input_pts
        for i, layer in enumerate(self.pts_linears):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        if self.use_viewdirs:
            assert input_views is not None
            alpha = self.alpha_linear(h)
            h = torch.cat([self.feature_linear(h), input_views], -1)
            for layer in self.views_linears:
                h = F.relu(layer(h))
            out = torch.cat([self.rgb_linear(h), alpha], -1)
        else:
            out = self.output_linear(h)
        if self.scale_type == 'linear':
            return out * self.scale
        if self.scale_type == 'tanh':
            return torch.tanh(out) * self.scale
        return out


def ray_to_samples(ray_batch, samples_per_ray, lindisp=False, perturb=0., device='cpu', append_t=None):
    """ray_batch: dict(origin [R,3], direction [R,3], near [R,1], far [R,1]) -> pts [R,S,3], dirs [R,S,3], z_vals [R,S]"""
    rays_o, rays_d = ray_batch['origin'], ray_batch['direction']
    near, far = ray_batch['near'], ray_batch['far']
    assert near.shape[0] == far.shape[0] == rays_o.shape[0]
    t = torch.linspace(0., 1., steps=samples_per_ray, device=device)
    z_vals = near * (1. - t) + far * t if not lindisp else 1. / (1. / near * (1. - t) + 1. / far * t)
    if perturb > 0.:                                        # stratified: one uniform draw inside every interval
        mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
        upper = torch.cat([mids, z_vals[..., -1:]], -1)
        lower = torch.cat([z_vals[..., :1], mids], -1)
        u = torch.clip(torch.rand(z_vals.shape, device=device), min=PERTURB_EPSILON, max=1 - PERTURB_EPSILON)
        z_vals = lower + (upper - lower) * u
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z_vals[..., :, None]
    dirs = torch.stack([rays_d] * samples_per_ray, axis=1)
    if append_t is not None:
        pts = torch.cat([pts, append_t.to(device)], dim=-1)
    return pts, dirs, z_vals


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkg=True):
    """raw [R,S,4] (rgb logits, sigma), z_vals [R,S], rays_d [R,3] -> rgb_map [R,3], disp_map [R], acc_map [R], weights [R,S], depth_map [R]"""
    device = raw.device
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    noise = torch.randn(raw[..., 3].shape, device=device) * raw_noise_std if raw_noise_std > 0. else 0.
    alpha = 1. - torch.exp(-F.relu(raw[..., 3] + noise) * dists)
    trans = torch.cumprod(torch.cat([torch.ones((alpha.shape[0], 1), device=device), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1. / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkg:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


def render_rays_vanilla(net, pos_embed, dir_embed, rays_o, rays_d, near, far, samples_per_ray, white_bkg=True, perturb=0.):
    """config 1 in one call: sample, encode, MLP, composite.  near / far: scalars or [R,1] tensors."""
    device = rays_o.device
    R = rays_o.shape[0]
    as_col = lambda v: v if isinstance(v, torch.Tensor) else torch.full((R, 1), float(v), device=device)
    pts, dirs, z_vals = ray_to_samples(dict(origin=rays_o, direction=rays_d, near=as_col(near), far=as_col(far)), samples_per_ray, perturb=perturb,
                                       device=device)
    raw = net(pos_embed(pts.reshape(-1, 3)), dir_embed(dirs.reshape(-1, 3)) if dir_embed is not None else None).reshape(R, samples_per_ray, -1)
    return raw2outputs(raw, z_vals, rays_d, white_bkg=white_bkg)
