"""The guidance leg's PyTorch-level settings, one at a time (fp32 throughout): ms of StableDiffusion.mannual_backward on the SD-1.5-sized
stand-in (avatarcraft_amd/sd_arch.py), phases by HIP events.  gpurun -- 'python tools/guidance_tune_probe.py'"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from avatarcraft_amd import sd_arch
from avatarcraft_amd.guidance import StableDiffusion, SDSGuidance

dev = torch.device("cuda:0")


def run(tag, **tune):
    torch.backends.cudnn.benchmark = False
    sd = StableDiffusion(dev, "1.5", components=sd_arch.components(device=dev))
    if tune:
        sd.tune(**tune)
    guide = SDSGuidance(sd, "Hulk, photorealistic style", 100.0)
    img = torch.rand(1, 3, 64, 64, device=dev)
    for _ in range(2):
        guide(img)
    torch.cuda.synchronize()
    sd.phase_marks = []
    n = 4
    t0 = time.perf_counter()
    for _ in range(n):
        guide(img)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    ph = {}
    mk = sd.phase_marks
    for (n0, e0), (n1, e1) in zip(mk[:-1], mk[1:]):
        if n1 != "start":
            ph[n1] = round(ph.get(n1, 0.0) + e0.elapsed_time(e1) / n, 2)
    print(tag, round(tot, 2), ph, flush=True)
    del sd, guide
    torch.cuda.empty_cache()


run("baseline (NCHW, no find mode)")
run("MIOpen find mode only", channels_last=None, miopen_find=True)
run("NHWC UNet only", channels_last="unet", miopen_find=False)
run("NHWC UNet + find mode", channels_last="unet", miopen_find=True)
run("NHWC VAE only", channels_last="vae", miopen_find=False)
run("NHWC both + find mode", channels_last="both", miopen_find=True)
