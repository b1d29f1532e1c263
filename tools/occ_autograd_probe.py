"""run_cuda's training form UNDER AUTOGRAD: forward + backward of one 4096-ray batch, ms, and the launches it makes.
gpurun -- 'python tools/occ_autograd_probe.py'"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
dev = torch.device("cuda:0")
p, _f, table, ro, rd = B.make_inputs(dev, 0)
net = B.make_net(p, table, dev, True, cuda_ray=True)
with torch.no_grad():
    net.deviation_net.variance.fill_(float(np.log(512.0) / 10.0))
net.update_extra_state(1.6)
ro_t, rd_t = torch.from_numpy(ro[6 * 4096:7 * 4096].copy()).to(dev), torch.from_numpy(rd[6 * 4096:7 * 4096].copy()).to(dev)
kw = dict(num_steps=64, bound=1.6, upsample_steps=64, bg_color=torch.ones(1, 3, device=dev), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, perturb=True)
net.mean_count, net.local_step = 0, 0
with torch.no_grad():
    net.render(ro_t[None], rd_t[None], **kw)
net.mean_count = int(net.step_counter[0, 0].item())
print("samples per batch", net.mean_count)


def step():
    net.zero_grad(set_to_none=True)
    out = net.render(ro_t[None], rd_t[None], **kw)
    loss = out["rgb"].sum() + out["weight_sum"].sum() + 0.1 * out["gradient_error"]
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print("train form under autograd: forward + backward %.3f ms per 4096-ray batch" % ((time.perf_counter() - t0) / n * 1e3))
with torch.no_grad():
    for _ in range(3):
        net.render(ro_t[None], rd_t[None], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.render(ro_t[None], rd_t[None], **kw)
    torch.cuda.synchronize()
    print("no-grad one launch: %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
