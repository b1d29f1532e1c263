"""Long repeat-launch soak at the final build: the same inputs, many launches, every output compared with the first launch bit for bit.
    renders   : bench batch + SDS training view, exact and fast, lean and with every per-sample output (incl. the 7 x 32 stencil features)
    pair      : ac_render_rays_pair
    backward  : ac_render_core_backward on fixed saved tensors / upstream gradients (table gradient through the binned scatter, MLP gradients)
    step      : stylize.sds_step from identical parameters, noise and guidance -> identical parameters after Adam
python tools/soak.py [scale]   (scale 1.0 ~ 3 minutes on one MI355X)"""
import sys, os, time, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from avatarcraft_amd import nsr_ops
from tests.common import make_rays, load_golden, make_table
from tests.gpu_common import device_field

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = torch.device("cuda:0")
p = load_golden("nsr_params.npz")
f, table = device_field(p, device=dev); f.prepare()
inv_s = float(p["inv_s"])
t0 = time.time()
report = []


def same(a, b):
    return all(torch.equal(a[k], b[k]) for k in a)


def soak(name, fn, n):
    first = {k: v.clone() for k, v in fn().items() if isinstance(v, torch.Tensor)}
    bad = 0
    for _ in range(n - 1):
        out = fn()
        bad += 0 if same(first, out) else 1
    torch.cuda.synchronize()
    report.append("%-64s %6d launches, %d differ from the first" % (name, n, bad))
    print(report[-1], flush=True)
    return bad


views = {"bench batch": tuple(torch.from_numpy(a[:4096].copy()).to(dev) for a in make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)),
         "sds view": tuple(torch.from_numpy(a).to(dev) for a in bench.sds_view(0))}
noise = torch.rand((4096, 64), generator=torch.Generator().manual_seed(5)).to(dev)
total_bad = 0
for vn, (ro, rd) in views.items():
    for prec in ("exact", "fast"):
        total_bad += soak(f"render {vn}, {prec}, lean", lambda: nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, inv_s, noise=noise, precision=prec), int(1500 * scale))
        total_bad += soak(f"render {vn}, {prec}, all per-sample outputs", lambda: nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, inv_s, noise=noise, precision=prec, extras=True,
                                                                                                         train_extras=True, debug_indices=True), int(300 * scale))
ro, rd = views["sds view"]
n2 = torch.rand((2, 4096, 64), generator=torch.Generator().manual_seed(6)).to(dev)


def pair():
    a, b = nsr_ops.render_rays_pair(f, ro, rd, n2, 64, 64, 1.6, inv_s)
    return {**{"a_" + k: v for k, v in a.items() if isinstance(v, torch.Tensor)}, **{"b_" + k: v for k, v in b.items() if isinstance(v, torch.Tensor)}}


total_bad += soak("pair launch (render_val + training forward), sds view", pair, int(300 * scale))
# backward on fixed saved tensors
out = nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, inv_s, noise=noise, extras=True, train_extras=True)
g = torch.Generator().manual_seed(9)
gi = torch.randn((4096, 3), generator=g).clamp(-1, 1).to(dev); gw = (torch.randn(4096, generator=g) * 3).to(dev); ge = torch.tensor(0.01, device=dev)


def backward():
    gt = torch.zeros_like(f.t["table"])
    a, b, c = nsr_ops.render_core_backward(f, out.opts, out, ro, rd, None, gi, gw, None, None, ge, gt)
    return {"g_table": gt, "g_sdf": a, "g_col": b, "g_invs": c}


total_bad += soak("render-core backward (binned scatter + MLP gradients), 4096 rays", backward, int(150 * scale))
# whole steps from identical state
from avatarcraft_amd.stylize import sds_step, flat_grad_view, SyntheticGuidance
tab = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
ro_h, rd_h = bench.sds_view(0)


def step():
    net = bench.make_net(p, tab, dev, True); net_gt = bench.make_net(p, tab, dev, False)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3, fused=True)
    flat = flat_grad_view(net.parameters())
    torch.manual_seed(11)
    for _ in range(2):
        sds_step(net, net_gt, ro, rd, (64, 64), opt, SyntheticGuidance(3), batch_size=4096, flat_grad=flat)
    return {k: v.detach() for k, v in net.named_parameters()}


total_bad += soak("two stylisation steps from identical state (parameters after Adam)", step, int(25 * scale))

# round 4: the occupancy-grid render in one launch, the per-sample field on packed samples, one posed frame (search + two render passes)
from avatarcraft_amd import raymarching
from avatarcraft_amd.synthetic import make_body
occ_net = bench.make_net(p, tab, dev, False, cuda_ray=True)
with torch.no_grad():
    occ_net.deviation_net.variance.fill_(float(np.log(512.0) / 10.0))
occ_net.update_extra_state(1.6)
vro, vrd = (torch.from_numpy(a).to(dev) for a in make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0))
occ_field = occ_net._field()
total_bad += soak("occupancy render, one launch, 65 536 rays",
                  lambda: nsr_ops.render_rays_occupancy(occ_field, vro, vrd, occ_net.density_grid, occ_net.mean_density, 1.6, 0.005, occ_net.forward_variance(), 1.0),
                  int(100 * scale))
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.6, occ_net.density_grid, occ_net.mean_density, 1, align=128, force_all_rays=True)
total_bad += soak("field on packed samples (ac_field_samples), %d samples" % xyzs.shape[0],
                  lambda: nsr_ops.field_samples(occ_field, xyzs, dirs, deltas, 1.6, 0.005, 512.0, 1.0, want_sdf=True, want_gradient=True), int(300 * scale))
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
wm = nsr_ops.WarpMesh(verts, faces, Ts, dev, 0.05, 0.05, True)
pro, prd = (torch.from_numpy(a).to(dev) for a in make_rays(256, 256, dist=1.8, f=443.405 / 2, yaw=0.3, pitch=-0.1))


def posed():
    o = nsr_ops.render_rays(f, pro, prd, 32, 32, 1.6, inv_s, warp=wm, skip_masked=True)
    return {k: o[k] for k in ("image", "weights_sum", "depth", "normal_map", "mask", "can_mid")}


total_bad += soak("posed frame (65 536 rays: two closest-face searches + two render passes)", posed, int(25 * scale))
# round 5: a field with view directions (the SH instantiation of the renderer: the per-ray bias is formed by a ray's first segment and handed on with the
# segment state), the SDF on a grid, marching cubes, the one-launch density-grid update
from avatarcraft_amd.instant_nsr import NeRFNetwork
torch.manual_seed(0)
vnet = NeRFNetwork(use_viewdirs=True)
vsd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net.1", "color_net.2", "deviation_net"))}
vsd["encoder.embeddings"] = torch.from_numpy(tab); vsd["encoder.offsets"] = torch.from_numpy(np.asarray(p["offsets"]))
vnet.load_state_dict(vsd, strict=False)
vnet = vnet.to(dev).eval()
with torch.no_grad():
    vf = vnet._field()
bro, brd = views["bench batch"]
total_bad += soak("render bench batch with view directions, exact, lean", lambda: nsr_ops.render_rays(vf, bro, brd, 64, 64, 1.6, inv_s, noise=noise), int(1500 * scale))
total_bad += soak("render sds view with view directions, exact, all per-sample outputs",
                  lambda: nsr_ops.render_rays(vf, ro, rd, 64, 64, 1.6, inv_s, noise=noise, extras=True, train_extras=True), int(300 * scale))
ax = torch.linspace(-1.6, 1.6, 256).to(dev)
vol = nsr_ops.field_sdf_grid(f, ax, ax, ax, 1.6, negate=True)
total_bad += soak("SDF on a 256^3 grid (ac_field_sdf_grid)", lambda: {"vol": nsr_ops.field_sdf_grid(f, ax, ax, ax, 1.6, negate=True)}, int(40 * scale))


def mesh():
    v, t = nsr_ops.marching_cubes(vol, 0.0, den=255.0, span=[3.2] * 3, lo=[-1.6] * 3)
    return {"v": v, "t": t}


total_bad += soak("marching cubes on that volume (%d vertices)" % mesh()["v"].shape[0], mesh, int(100 * scale))
ax129 = torch.linspace(-1.6, 1.6, 129).to(dev)
g0 = torch.rand((129, 129, 129), generator=torch.Generator().manual_seed(4)).to(dev) * 40.0


def grid_update():
    g = g0.clone()
    m = nsr_ops.density_grid_update(occ_field, ax129, g, 1.6, 512.0, 0.95)
    return {"grid": g, "mean": m}


total_bad += soak("density-grid update, one launch (129^3)", grid_update, int(200 * scale))
# the occupancy-grid training form in one launch (four phases, three grid barriers) on the SDS view, on a grid this field's own update produced
og = torch.zeros((129, 129, 129), device=dev)
for _ in range(3):
    og_mean = nsr_ops.density_grid_update(occ_field, ax129, og, 1.6, 512.0, 0.95)
og_mean = float(og_mean.item())
ctr = torch.zeros(2, dtype=torch.int32, device=dev)
sro, srd = views["sds view"]


def occ_train():
    ctr.zero_()
    o = nsr_ops.render_rays_occupancy_train(occ_field, sro, srd, og, og_mean, 1.6, 0.005, 512.0, 1.0, perturb=True, capacity=1 << 17, counter=ctr, bg=1.0)
    o["counter"] = ctr.clone()
    return o


total_bad += soak("occupancy training form, one launch (%d samples)" % int(occ_train()["counter"][0].item()), occ_train, int(2000 * scale))
print("hand-off timeouts on this stream:", nsr_ops.handoff_timeouts(dev), "| occupancy launches with a timed-out grid barrier:", nsr_ops.occupancy_launch_failures())
print("total: %d differing repeats; %.0f s" % (total_bad, time.time() - t0))
