"""Build libavatarcraft_hip.so for gfx950 with hipcc (in-tree, no JIT-on-import).

    python -m avatarcraft_amd.build [--force]

hipcc cross-compiles without a GPU.  Flags that are part of the numerics contract:
  -ffp-contract=off   every fused multiply-add in the kernels is an explicit fma
  -fno-slp-vectorize  no compiler-formed packed-fp32 arithmetic (see FLAGS)
  (no -ffast-math; fp32 divide / sqrt stay correctly rounded, denormals are kept)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libavatarcraft_hip.so")
SOURCES = ["ac_capi.hip", "hashgrid.hip", "hash_stencil.hip", "shencoder.hip", "raymarching.hip", "render_fused.hip", "sdf_train.hip", "warp.hip", "step_glue.hip", "geometry.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wno-unused-result",
         # no compiler-formed packed-fp32 (v_pk_*_f32) code: the one run-to-run non-determinism ever observed in the renderer (round 2: the "face-value"
         # stencil variant; round 3: reproduced and bisected) appears exactly when the SLP vectorizer packs the xy weight
         # products and the two feature channels of that variant into v_pk_mul / v_pk_fma with crossed op_sel, and disappears with this flag;
         # measured cost of the flag on the shipped kernels: none (render 0.799 vs 0.796 ms; per kernel on one-batch posed frames at the end of round 3, vectoriser
         # on / off: warp search 4.86 / 4.74 ms, mesh near-far 0.33 / 0.17 ms, render passes equal -- profiles/r03_experiments.txt section 19)
         "-fno-slp-vectorize"]

# per-source additions: the instruction scheduler LLVM's AMDGPU back end uses for that file (scheduling only -- no arithmetic changes; every kernel stays bit-identical to
# the oracle, which the GPU tests check).  Measured same-box, profiles/r03_experiments.txt sections 20 - 21:
#   hash_stencil.hip (table-gradient scatter): max-memory-clause -- the queue fill is a long sequence of LDS record traffic between short arithmetic sections:
#       1.246 -> 1.188 ms per 4096-ray patch (the same strategy costs the render kernel 7 %, so it is not a global flag);
#   render_fused.hip: iterative-maxocc -- 0.763 -> 0.752 ms per 4096-ray launch of the instantiation without per-sample outputs (+0.6 % on the training one);
#   warp.hip: iterative-maxocc -- posed frame 8.21 -> 8.08 ms.
PER_FILE_FLAGS = {"hash_stencil.hip": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
                  "render_fused.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"],
                  "warp.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]}


# Library variants built next to the product (in-tree, so that they travel to the GPU box; git-ignored like every .so): the same objects except the listed
# sources, which are compiled with extra defines.  Loaded through AC_LIB_PATH by the tests that keep a compile-time alternative exercised.
#   rec12: the table-gradient scatter with full-fp32 12-byte queue records (-DAC_REC8=0) instead of the 8-byte records whose values are rounded to 16 / 17
#          mantissa bits (the precision reduction is a compile-time choice; the fp32 form must stay covered by the parity matrix)
VARIANTS = {"rec12": {"hash_stencil.hip": ["-DAC_REC8=0"]}}


def variant_path(name):
    return os.path.join(HERE, f"libavatarcraft_hip_{name}.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


_PROBED = {}


def _supported(hipcc, flags):
    """the per-file scheduler strategies are internal LLVM options (-mllvm ...): an older or newer ROCm may not know them.  They change instruction
    ORDER only (every kernel stays bit-identical), so a toolchain that rejects them builds without them, with a warning -- instead of not at all."""
    key = tuple(flags)
    if not flags:
        return True
    if key not in _PROBED:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void k(float *p) { p[0] = 1.0f; }\n")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-c", src, "-o", os.path.join(td, "probe.o")] + list(flags),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        _PROBED[key] = r.returncode == 0
        if not _PROBED[key]:
            print(f"avatarcraft_amd.build: this hipcc rejects {' '.join(flags)} (scheduling only): building without it", file=sys.stderr)
    return _PROBED[key]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "avatarcraft_hip.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        extra = PER_FILE_FLAGS.get(os.path.basename(s), [])
        cmd = [hipcc] + FLAGS + (extra if _supported(hipcc, extra) else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}")
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    for name, per_src in VARIANTS.items():
        vobjs, rebuilt = [], False
        for src in SOURCES:
            if src not in per_src:
                vobjs.append(os.path.join(bdir, src.replace(".hip", ".o")))
                continue
            s_ = os.path.join(CSRC, src)
            o = os.path.join(bdir, f"{name}_{src.replace('.hip', '.o')}")
            if force or _stale(o, [s_] + headers):
                extra = PER_FILE_FLAGS.get(src, [])
                cmd = [hipcc] + FLAGS + (extra if _supported(hipcc, extra) else []) + per_src[src] + ["-c", s_, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed for variant {name} of {src}:\n{r.stdout}")
                rebuilt = True
            vobjs.append(o)
        vso = variant_path(name)
        if force or rebuilt or _stale(vso, vobjs):
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", vso] + vobjs, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link of variant {name} failed:\n{r.stdout}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
