"""-m gpu: the drop-in boundary from plain C.  examples/render_c_abi.c includes include/avatarcraft_hip.h and the HIP runtime's C API only -- no Python, no torch --
uploads a field and rays from a file, calls ac_field_prepare + ac_render_rays and writes the image.  Its pixels must be the Python path's (ctypes -> the same
library) bit for bit, the CPU oracle's bit for bit, and the reference's golden render within the north-star tolerance."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from tests.common import load_golden, make_table
from tests.gpu_common import device_field, oracle_field, assert_bitwise

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_renders_the_golden_rays_through_the_c_abi(tmp_path):
    from avatarcraft_amd import _lib, nsr_ops
    from oracle import oracle as O
    exe = str(tmp_path / "render_c_abi")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "render_c_abi.c"), _lib.LIB_PATH, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                    f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    p = load_golden("nsr_params.npz")
    g = load_golden("run_eval_64_64.npz")
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    ro, rd = g["rays_o"].astype(np.float32), g["rays_d"].astype(np.float32)
    n = ro.shape[0]
    S = np.float32(np.log2(float(p["per_level_scale"])))
    lin_z = torch.linspace(0.0, 1.0, 64, dtype=torch.float32).numpy()
    lin_u = torch.linspace(0.5 / 16, 1.0 - 0.5 / 16, steps=16, dtype=torch.float32).numpy()
    blob = tmp_path / "in.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<4i", n, 64, 64, 16))
        f.write(np.asarray(p["offsets"], np.int32).tobytes())
        f.write(struct.pack("<3f", float(S), 1.6, float(p["inv_s"])))
        for a in (table, p["W1"], p["b1"], p["W2"], p["b2"], p["Wc1"], p["Wc2"], p["Wc3"], ro, rd, lin_z, lin_u):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(blob), str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert f"{n} rays x (64 + 64) samples rendered through the C ABI" in r.stdout
    raw = np.fromfile(out, np.float32)
    assert raw.size == n * 10
    image, wsum, depth, nmap, eik = raw[:3 * n].reshape(n, 3), raw[3 * n:4 * n], raw[4 * n:5 * n], raw[5 * n:8 * n].reshape(n, 3), raw[8 * n:].reshape(n, 2)
    # (a) the Python path: ctypes -> the same entry point
    f_dev, _ = device_field(p, device="cuda:0")
    o = nsr_ops.render_rays(f_dev, torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda(), 64, 64, 1.6, float(p["inv_s"]))
    for k, v in (("image", image), ("weights_sum", wsum), ("depth", depth), ("normal_map", nmap)):
        assert_bitwise(o[k], v, f"C program vs Python path: {k}")
    # (b) the CPU oracle, bit for bit; (c) the reference's own render of these rays
    ref = O.render_rays(oracle_field(p, table), ro, rd, 64, 64, 1.6, float(p["inv_s"]), bg=np.ones((n, 3), np.float32))
    assert_bitwise(image, ref["image"], "C program vs oracle: image")
    assert_bitwise(wsum, ref["weights_sum"], "C program vs oracle: weights_sum")
    white = g["image"] + (1.0 - g["weights_sum"])[:, None] * (1.0 - g["bg"])           # the golden was rendered over a random background: re-blend over white
    assert np.abs(image - white).max() <= 1e-3 and np.abs(wsum - g["weights_sum"]).max() <= 1e-3
    assert np.isfinite(eik).all() and float(eik[:, 1].sum()) > 0
