#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing (tools: `hipcc ... --cuda-device-only -S -o x.s file.hip`):
    python tools/isa_stats.py x.s [name-substring]
Counts per kernel: vector ALU, MFMA, LDS, VMEM, scalar, spill traffic -- the static counterpart of SQ_INSTS_*."""
import re
import sys
from collections import Counter


def stats(path, filt=""):
    cur = None
    out = {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); out[cur] = Counter(); continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None; continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";", "//")) or t[0].endswith(":"):
            continue
        op = t[0]
        c = out[cur]
        if op.startswith("v_mfma") or op.startswith("v_smfma"): c["mfma"] += 1
        elif op.startswith("v_"): c["valu"] += 1; c[op.split("_e")[0]] += 0
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")): c["vmem"] += 1
        elif op.startswith("scratch_"): c["scratch"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        if op.startswith(("v_readlane", "v_writelane")): c["lane_spill"] += 1
    for k, c in out.items():
        if filt in k and c["valu"]:
            print(f"{k[:70]:70s} valu {c['valu']:6d} mfma {c['mfma']:5d} lds {c['lds']:5d} vmem {c['vmem']:5d} scratch {c['scratch']:4d} salu {c['salu']:5d} waitcnt {c['waitcnt']:4d} lane r/w {c['lane_spill']:4d}")


if __name__ == "__main__":
    stats(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
