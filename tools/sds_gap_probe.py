"""tools/sds_gap_probe.py -- see sds_gap_probe.sh.  `run`: 2 warm-up + 6 timed stylisation steps (coarse: 64 x 64 rays; fine: 256 x 256 = 16 patches)
bracketed by marker kernels; `report`: per step, kernel-busy time, wall time and the idle gaps on the device, from the kernel trace."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(which):
    import torch
    import bench as B
    from avatarcraft_amd import stylize as ST
    from avatarcraft_amd.synthetic import make_rays
    dev = torch.device("cuda:0")
    p, _field, table, _ro, _rd = B.make_inputs(dev, 0)
    net, net_gt = B.make_net(p, table, dev, True), B.make_net(p, table, dev, False)
    opt = ST.Adam(net.parameters(), lr=5e-3, zero_grad_in_step=True)
    flat = ST.flat_grad_view(net.parameters())
    guide = ST.SyntheticGuidance(42)
    if which == "fine":
        ro, rd = make_rays(256, 256, dist=1.8, f=200.0, yaw=0.0, pitch=0.0); hw = (256, 256)
    else:
        ro, rd = B.sds_view(0); hw = (64, 64)
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    marker = torch.zeros(7, device=dev)
    for _ in range(2):
        ST.sds_step(net, net_gt, ro, rd, hw, opt, guide, batch_size=4096, flat_grad=flat)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    n = 6 if which != "fine" else 3
    for _ in range(n):
        torch.cumsum(marker, 0, out=marker)              # the step marker in the trace (no other cumsum in a step)
        ST.sds_step(net, net_gt, ro, rd, hw, opt, guide, batch_size=4096, flat_grad=flat)
    torch.cumsum(marker, 0, out=marker)
    torch.cuda.synchronize()
    print("host ms per step", (time.perf_counter() - t0) / n * 1e3)


def report(odir, which):
    f = glob.glob(os.path.join(odir, "kt", "**", "*kernel_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:48]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))
    marks = [i for i, e in enumerate(ev) if "adam_step_kernel" in e[2]]           # a step ends with the optimizer's launch
    print(open(os.path.join(odir, "run.log")).read().strip().splitlines()[-1])
    if len(marks) < 2:
        print("no markers found; kernels:", sorted({e[2] for e in ev})[:40]); return
    steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
    for k, (a, b) in enumerate(steps):
        seg = ev[a + 1:b + 1]
        busy = sum(e[1] - e[0] for e in seg)
        wall = seg[-1][1] - seg[0][0]
        gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1]), seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)), reverse=True)
        idle = sum(max(0, g[0]) for g in gaps)
        print(f"step {k}: {len(seg)} kernels, wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms; largest gaps (us): " +
              "; ".join(f"{g[0] / 1e3:.1f} {g[1]} -> {g[2]}" for g in gaps[:8]))
    a, b = steps[-1]
    seg = ev[a + 1:b + 1]
    print("-- last step, kernels in order (start offset us, duration us):")
    t0 = seg[0][0]
    agg = {}
    for s, e, n in seg:
        agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
    if which != "fine":
        for s, e, n in seg:
            print(f"   {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f}  {n}")
    else:
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"   {t:10.1f} us  x{c:4d}  {n}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "coarse")
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "coarse")
