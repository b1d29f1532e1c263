#!/usr/bin/env python3
"""Build ablation / A-B variants of libavatarcraft_hip.so HERE (hipcc cross-compiles without a GPU) into tools/_bin/ (git-ignored, but it
travels to the GPU box with the snapshot), so that no GPU-minute is spent compiling:

    python tools/build_variants.py name1:"-DFLAG=1 -DOTHER" name2:"" ...

tools/run_variants.sh then times bench.py on each of them through AC_LIB_PATH."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avatarcraft_amd.build import SOURCES, FLAGS, CSRC, PER_FILE_FLAGS  # noqa: E402

OUT = os.path.join(ROOT, "tools", "_bin")


def build(spec):
    name, _, flags = spec.partition(":")
    flags, _, only = flags.partition("@")            # name:"flags@file.hip": the flags go to that source file only
    os.makedirs(OUT, exist_ok=True)
    objs = []
    all_flags = flags
    for src in SOURCES:
        flags = all_flags if (not only or src == only) else ""
        o = os.path.join(OUT, f"{name}_{src.replace('.hip', '.o')}")
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + PER_FILE_FLAGS.get(src, []) + flags.split() + ["-c", os.path.join(CSRC, src), "-o", o], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            return name, r.stdout[-2000:]
        objs.append(o)
    so = os.path.join(OUT, f"lib_{name}.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    for o in objs:
        os.remove(o)
    return name, (r.stdout[-2000:] if r.returncode else "ok")


if __name__ == "__main__":
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, msg in ex.map(build, sys.argv[1:]):
            print(name, msg)
