"""soak test: N SDS steps with the synthetic guidance over the stylize outer loop -- finite parameters, stable memory, step time"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd.instant_nsr import NeRFNetwork
from avatarcraft_amd.stylize import stylize_epochs, SyntheticGuidance, flat_grad_view
from tests.common import load_golden, make_table
dev = "cuda:0"
p = load_golden("nsr_params.npz")
def make(train):
    torch.manual_seed(0); net = NeRFNetwork()
    sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net", "deviation_net"))}
    sd["encoder.embeddings"] = torch.from_numpy(make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"]))
    sd["encoder.offsets"] = torch.from_numpy(p["offsets"]); net.load_state_dict(sd); return net.to(dev).train(train)
net, net_gt = make(True), make(False)
opt = torch.optim.Adam(net.parameters(), lr=5e-3)
flat = flat_grad_view(net.parameters())
times, mem = [], []
def on_step(step, epoch, stats):
    torch.cuda.synchronize(); times.append(time.perf_counter()); mem.append(torch.cuda.memory_allocated() / 2**20)
n = int(os.environ.get("NCAP", 40))
steps = stylize_epochs(net, net_gt, opt, SyntheticGuidance(1), hw=(256, 256), n_cap=n, coarse_epochs=1, fine_epochs=1, subsample_scale=4, augment_cam=True,
                       stylize_head=True, coarse_head=0.2, fine_head=0.2, augment_bkg=True, augment_text=True, tgt_text="Hulk", device=dev, flat_grad=flat,
                       on_step=on_step)
dt = np.diff(np.array(times)) * 1e3
finite = all(bool(torch.isfinite(q).all()) for q in net.parameters())
print(f"steps {steps}  finite {finite}  coarse step ms median {np.median(dt[:n]):.2f}  fine (4x the rays) median {np.median(dt[n + 8:]):.2f}  mem MB first {mem[2]:.0f} last {mem[-1]:.0f} peak {torch.cuda.max_memory_allocated() / 2**20:.0f}")
