import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from tests.test_gpu_model import golden_net, DEV
net, _ = golden_net(train=True)
rs = np.random.RandomState(3)
x = rs.uniform(-1.5, 1.5, (4096, 3)).astype(np.float32)
eps, bound = 0.005, 1.6
for case in ("centre", "grad"):
    go = torch.from_numpy(rs.normal(0, 1, (4096, 16)).astype(np.float32)).to(DEV) * (1.0 if case == "centre" else 0.0)
    gg = torch.from_numpy(rs.normal(0, 1, (4096, 3)).astype(np.float32)).to(DEV) * (0.0 if case == "centre" else 1.0)
    res = {}
    for mode in ("fused", "generic", "generic_noenc", "generic_encOnly"):
        net.zero_grad()
        xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
        if mode == "fused":
            net.fused_training = "core"
            o16, grad = net.forward_sdf_stencil(xt, bound, eps)
        else:
            net.fused_training = False
            def fsdf(xx):
                xe = xx.detach() if mode == "generic_noenc" else xx
                xi = xx.detach() if mode == "generic_encOnly" else xx
                h = net.encoder(xe, bound)
                h = torch.cat([xi, h], dim=-1)
                for l in range(net.num_layers):
                    h = net.sdf_net[l](h)
                    if l != net.num_layers - 1: h = net.activation(h)
                return h
            o16 = fsdf(xt)
            outs = []
            for k in range(3):
                e = torch.zeros(1, 3, device=DEV); e[0, k] = eps
                outs.append(0.5 * (fsdf((xt + e).clamp(-bound, bound))[:, :1] - fsdf((xt - e).clamp(-bound, bound))[:, :1]) / eps)
            grad = torch.cat(outs, -1)
        ((o16 * go).sum() + (grad * gg).sum()).backward()
        res[mode] = xt.grad.detach().cpu().numpy().astype(np.float64)
    f, g = res["fused"], res["generic"]
    print(case, "max|generic|", np.abs(g).max(), "err fused-generic", np.abs(f - g).max(), "noenc share max", np.abs(res["generic_noenc"]).max(), "enc share max", np.abs(res["generic_encOnly"]).max(),
          "sum of shares - generic", np.abs(res["generic_noenc"] + res["generic_encOnly"] - g).max())
    i = np.unravel_index(np.abs(f - g).argmax(), f.shape)
    print("  worst", i, f[i[0]], g[i[0]], res["generic_noenc"][i[0]], res["generic_encOnly"][i[0]])
    e = np.abs(f - g) / np.abs(g).max()
    print("  entries > 3e-4:", int((e > 3e-4).sum()), "of", e.size, " > 1e-2:", int((e > 1e-2).sum()), "rows", np.unique(np.argwhere(e > 3e-4)[:, 0])[:20])
