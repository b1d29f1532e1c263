"""Timing ablations of render_rays_kernel (NOT correct results): which resource bounds the kernel?
builds variants with -DAC_ABL_{SOFTPLUS,MFMA,GATHER} and times one 4096-ray launch of each."""
import ctypes, os, subprocess, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
csrc = os.path.join(ROOT, "avatarcraft_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("ac_capi.hip", "hashgrid.hip", "shencoder.hip", "raymarching.hip", "render_fused.hip", "hash_stencil.hip", "sdf_train.hip", "warp.hip")]
from tests.common import load_golden, make_rays
p = load_golden("nsr_params.npz")
ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
variants = {"xcd_remap": [], "no_remap": ["-DAC_NO_XCD_REMAP"]} if "--xcd" in sys.argv else {"cg_wpb8": ["-DAC_ABL_GATHER"], "cg_wpb4": ["-DAC_ABL_GATHER", "-DAC_WPB=4"], "cg_wpb2": ["-DAC_ABL_GATHER", "-DAC_WPB=2"],
            "nosp_nomfma_wpb8": ["-DAC_ABL_SOFTPLUS", "-DAC_ABL_MFMA"], "nosp_nomfma_wpb4": ["-DAC_ABL_SOFTPLUS", "-DAC_ABL_MFMA", "-DAC_WPB=4"], "nosp_nomfma_wpb2": ["-DAC_ABL_SOFTPLUS", "-DAC_ABL_MFMA", "-DAC_WPB=2"]} if "--wpb2" in sys.argv else {"baseline": [], "wpb4": ["-DAC_WPB=4"], "wpb2": ["-DAC_WPB=2"]} if "--wpb" in sys.argv else {"baseline": [], "no_softplus": ["-DAC_ABL_SOFTPLUS"], "no_mfma": ["-DAC_ABL_MFMA"], "cached_gather": ["-DAC_ABL_GATHER"],
            "no_softplus+no_mfma": ["-DAC_ABL_SOFTPLUS", "-DAC_ABL_MFMA"], "all_three": ["-DAC_ABL_SOFTPLUS", "-DAC_ABL_MFMA", "-DAC_ABL_GATHER"]}
procs = {}
for name, fl in variants.items():
    out = os.path.join(ROOT, "gpurun_out", f"libac_{name.replace('+','_')}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    procs[name] = (out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                          "-Wno-unused-result", "-o", out] + fl + srcs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
for name, (out, pr) in procs.items():
    pr.wait()
code = '''
import sys, os; sys.path.insert(0, %r)
import torch, numpy as np
from avatarcraft_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from avatarcraft_amd import nsr_ops
from tests.common import load_golden, make_rays
from tests.gpu_common import device_field
p = load_golden("nsr_params.npz"); f, _ = device_field(p)
ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
ro, rd = torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()
out = {}
for k in range(4): nsr_ops.render_rays(f, ro[k*4096:(k+1)*4096], rd[k*4096:(k+1)*4096], 64, 64, 1.6, float(p["inv_s"]), out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for k in range(32): nsr_ops.render_rays(f, ro[(k%%16)*4096:(k%%16+1)*4096], rd[(k%%16)*4096:(k%%16+1)*4096], 64, 64, 1.6, float(p["inv_s"]), out=out)
e.record(); torch.cuda.synchronize()
print("%%-22s %%.3f ms / launch" %% (sys.argv[2], s.elapsed_time(e) / 32))
''' % ROOT
for name, (out, pr) in procs.items():
    subprocess.run([sys.executable, "-c", code, out, name])
