"""ac_field_sdf_grid at 512^3 under the experiment switches AC_GRID_ORDER / AC_GRID_ROUND (one process per setting: the library reads them once)."""
import json, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, bench
    from avatarcraft_amd import nsr_ops
    dev = torch.device("cuda", 0)
    p, field, table, ro, rd = bench.make_inputs(dev, 0)
    net = bench.make_net(p, table, dev, False)
    res = int(sys.argv[2])
    with torch.no_grad():
        f = net._field(); ax = net._grid_axis(1.6, res)
        vol = torch.empty((res,) * 3, device=dev)
        nsr_ops.field_sdf_grid(f, ax, ax, ax, 1.6, out=vol); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); nsr_ops.field_sdf_grid(f, ax, ax, ax, 1.6, out=vol); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(json.dumps({"order": os.environ.get("AC_GRID_ORDER"), "round": os.environ.get("AC_GRID_ROUND"), "res": res, "ms": sorted(ts)[1],
                          "checksum": float(vol.double().sum())}))
    sys.exit(0)
for res in (512, 256):
    for order in ("0", "1"):
        for rnd in ("2", "4"):
            env = dict(os.environ, AC_GRID_ORDER=order, AC_GRID_ROUND=rnd)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(res)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else ("FAILED " + r.stderr[-400:]))
