"""Copy the evidence of one `tools/prof_r02.sh <tag>` run from gpurun_out/ (scratch) into profiles/ (tracked) and rewrite
profiles/traffic.json from its PMC passes.

    python tools/collect_profiles.py r02a [--round r02]

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default `python bench.py --steps 20 --warmup 5`),
<round>_kernel_stats_by_workload.json (the same trace split by the bench's three workloads: the render kernel serves all of them),
<round>_pmc_summary.json (per-kernel means of every PMC group), <round>_bench.json (the JSON line printed under the profiler)
and traffic.json."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r05"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]


WHOLE_VIEW = 4          # bench.py's informational whole-view launches of the same kernel, between the main bench and the SDS steps
SIMDS, XCDS = 1024, 8   # MI355X: 256 CUs x 4 SIMDs in 8 XCDs (GRBM_GUI_ACTIVE is summed over the XCDs)


def busy_objects(c):
    """The two counter-derived resources that bind the gather kernels, from one kernel's per-launch counter means `c`:
    issue  = (SQ_INSTS_VALU x 4 + SQ_INSTS_MFMA x 32 clocks) / (1024 SIMDs x kernel clocks): the share of the SIMDs' issue time the vector and fp32-matrix
             instructions need (a 64-lane VALU instruction occupies its SIMD for 4 clocks, v_mfma_f32_16x16x4_f32 for 32; the few quarter-rate
             instructions -- v_mul_lo_u32, v_rcp_f32 -- make this a lower bound);
    gather = TA_BUSY_avr / kernel clocks: the per-CU texture-address path every table gather goes through.
    Kernel clocks = GRBM_GUI_ACTIVE / 8 XCDs (the counter is summed over the XCDs; it reproduces the kernel time at 2.4 GHz)."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "GRBM_GUI_ACTIVE")
    if any(k not in c or not c[k] for k in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE")):
        return None
    clk = c["GRBM_GUI_ACTIVE"] / XCDS
    valu, mfma = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_MFMA", 0.0)
    out = {"issue": {"bound": "valu+mfma issue", "busy_frac": round((valu * 4 + mfma * 32) / SIMDS / clk, 4), "valu_insts_per_launch": int(valu),
                     "mfma_insts_per_launch": int(mfma), "kernel_clocks": int(clk),
                     "formula": "(SQ_INSTS_VALU x 4 + SQ_INSTS_MFMA x 32) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}}
    if c.get("TA_BUSY_avr"):
        out["gather"] = {"bound": "L1 gather (texture addresser)", "busy_frac": round(c["TA_BUSY_avr"] / clk, 4), "formula": "TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs)"}
    if c.get("TCP_TCC_READ_REQ_sum"):
        out["l2_sector_bytes_per_launch"] = int(c["TCP_TCC_READ_REQ_sum"] * 64)          # what the L1s actually asked of L2: 64-byte sectors
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / SIMDS / clk, 4)
    return out


def grid_kernel_split(src):
    """field_sdf_grid_kernel serves two bench legs (the 512^3 mesh export and, since round 6, the 129^3 density grid): its dispatches come in the same order in
    every pass, the kernel trace's durations say which is which (> 5 ms = the export) -> {label: {counter: mean}}"""
    tr = glob.glob(src + "/kt/**/p_kernel_trace.csv", recursive=True)
    if not tr:
        return {}
    rows = sorted((r for r in csv.DictReader(open(tr[0])) if "field_sdf_grid_kernel" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    labels = ["mesh" if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) > 5e6 else "density" for r in rows]
    out = {"mesh": collections.defaultdict(list), "density": collections.defaultdict(list)}
    for f in glob.glob(src + "/g*/p_counter_collection.csv"):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if "field_sdf_grid_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for c, v in per.items():
            seq = [v[i] for i in sorted(v)]
            if len(seq) == len(labels):
                for lab, x in zip(labels, seq):
                    out[lab][c].append(x)
    return {lab: {c: sum(v) / len(v) for c, v in d.items()} for lab, d in out.items() if d}


def roofline_objects(summary, byc, grid_split=None):
    """-> the `binding` block of traffic.json: per kernel of a bench leg, busy_objects of its counters"""
    out = {}
    for lab, name in (("mesh", "field_sdf_grid_kernel"), ("density", "density_grid_kernel")):
        if grid_split and lab in grid_split:
            o = busy_objects(grid_split[lab])
            if o:
                out[name] = o
    main = {k: v.get("main bench (4096 rays)") for k, v in byc.items() if isinstance(v, dict)}
    o = busy_objects(main)
    if o:
        out["render_rays_kernel (main bench, 4096 rays)"] = o
    for name, key in (("field_sdf_grid_kernel", "field_sdf_grid_kernel"), ("density_grid_kernel", "density_grid_kernel"), ("warp_samples_accel_kernel", "warp_samples_accel_kernel"),
                      ("sdf_stencil_bwd_kernel", "sdf_stencil_bwd_kernel"), ("hash_stencil_bwd_binned_kernel", "hash_stencil_bwd_binned_kernel"),
                      ("bucket_accumulate_kernel", "bucket_accumulate_kernel"), ("color_bwd_kernel", "color_bwd_kernel")):
        ks = [k for k in summary if k.startswith(key)]
        if ks and name not in out:
            o = busy_objects(summary[ks[0]])
            if o:
                out[name] = o
    return out


def main():
    shutil.copy(glob.glob(src + "/kt/**/p_kernel_stats.csv", recursive=True)[0], f"{dst}/{rnd}_kernel_stats.csv")
    shutil.copy(src + "/summary.json", f"{dst}/{rnd}_pmc_summary.json")
    line = [l for l in open(src + "/kt_bench.json") if l.startswith("{")][-1]
    bench = json.loads(line)
    json.dump(bench, open(f"{dst}/{rnd}_bench.json", "w"), indent=1)
    W, K = bench["warmup"], bench["steps"]

    # --- kernel trace split by workload: dispatch order is main bench (W + K launches), then the SDS steps, then the posed frames
    rows = list(csv.DictReader(open(glob.glob(src + "/kt/**/p_kernel_trace.csv", recursive=True)[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # render_rays_kernel<MODE, FAST, EX, SH> (SH = a field with view directions, round 5; false here): <0, false, false> = exact, no per-sample outputs (the bench, render_val, the frozen net's render);
    # <0, false, true> = exact with per-sample outputs (the training forward); <0, true, *> = the fast arithmetic mode
    lean = [dur(r) for r in rows if "render_rays_kernel<0, false, false, false>" in r["Kernel_Name"]]
    fast = [dur(r) for r in rows if "render_rays_kernel<0, true, false, false>" in r["Kernel_Name"]]
    train = [dur(r) for r in rows if "render_rays_kernel<0, false, true, false>" in r["Kernel_Name"]]
    R = int(bench.get("repeat", 1))
    nm = W + R * K
    # bench.py (default precision exact): W + repeat x K exact launches, then the other mode (min(W, 2) + K fast launches), WHOLE_VIEW exact launches of
    # 65 536 rays, then the SDS steps (one pair launch = render_val + the training forward, and one lean render of the frozen net, each)
    split = {"render_rays_kernel<0, exact, lean> main bench, timed launches": lean[W:nm], "render_rays_kernel<0, exact, lean> main bench, warm-up": lean[:W],
             "render_rays_kernel<0, fast, lean> the other arithmetic mode, same launches (roofline.other_precision)": fast[min(W, 2):min(W, 2) + K],
             "render_rays_kernel<0, exact, lean> whole view (65 536 rays) in one launch": lean[nm:nm + WHOLE_VIEW],
             "render_rays_kernel<0, exact, lean> SDS step: the frozen net's render (training view: every ray hits the body)": lean[nm + WHOLE_VIEW:],
             "render_rays_kernel<0, exact, per-sample outputs> SDS step: render_val + the training forward in ONE launch (ac_render_rays_pair, 2 x 4096 rays)": train}
    byw = {k: dict(calls=len(v), avg_us=sum(v) / len(v), min_us=min(v), max_us=max(v)) for k, v in split.items() if v}
    byw["bench_line"] = dict(kernel_ms_hip_events=bench["roofline"]["kernel_ms"], ms_per_step=bench["ms_per_step"])
    json.dump(byw, open(f"{dst}/{rnd}_kernel_stats_by_workload.json", "w"), indent=1)

    # --- PMC: per-dispatch values; the first (warmup + steps) dispatches of the render kernel are the main bench
    def per_dispatch(counter):
        out = collections.defaultdict(dict)
        for f in glob.glob(src + "/g*/p_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    out[short(r["Kernel_Name"])][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        return {k: [v[i] for i in sorted(v)] for k, v in out.items()}
    fetch, write = per_dispatch("FETCH_SIZE"), per_dispatch("WRITE_SIZE")
    log = open(glob.glob(src + "/g1.log")[0]).read()
    pb = json.loads([l for l in log.splitlines() if l.startswith("{")][-1])
    n_main = pb["warmup"] + pb["steps"]
    rk = [k for k in fetch if k.startswith("render_rays_kernel<0, false, false")][0]
    rkt = [k for k in fetch if k.startswith("render_rays_kernel<0, false, true")]
    main_f = fetch[rk][:n_main]
    sds_f, sds_w = fetch[rk][n_main + WHOLE_VIEW:], write[rk][n_main + WHOLE_VIEW:]
    trn_f, trn_w = (fetch[rkt[0]], write[rkt[0]]) if rkt else ([0.0], [0.0])
    KB = 1024
    mean = lambda v: sum(v) / len(v)
    parts = {"render_rays_kernel (lean: the frozen net's render)": (mean(sds_f) + mean(sds_w)) * KB,
             "render_rays_kernel (pair launch: render_val + training forward)": (mean(trn_f) + mean(trn_w)) * KB}
    for k in fetch:
        if any(s in k for s in ("hash_stencil_bwd", "bucket_acc", "sdf_stencil_bwd", "color_bwd", "composite_bwd", "core_mid", "core_normals")):
            parts[k] = (mean(fetch[k]) + mean(write[k])) * KB
    # every counter of the render kernel split by workload (summary.json averages over all dispatches of a kernel name, and the fast kernel serves
    # the main bench, the whole-view launches and the renders of the SDS steps)
    byc = {}
    for f in glob.glob(src + "/g*/p_counter_collection.csv"):
        vals = collections.defaultdict(dict)
        valt = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) == rk:
                vals[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
            elif rkt and short(r["Kernel_Name"]) == rkt[0]:
                valt[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for c, v in vals.items():
            seq = [v[i] for i in sorted(v)]
            byc[c] = {"main bench (4096 rays)": mean(seq[:n_main]), "whole view (65 536 rays)": mean(seq[n_main:n_main + WHOLE_VIEW]),
                      "SDS step: the frozen net's render (4096 rays, training view, no per-sample outputs)": mean(seq[n_main + WHOLE_VIEW:])}
            if c in valt:
                byc[c]["SDS step: pair launch (render_val + training forward, 2 x 4096 rays, per-sample outputs + stencil features of the second copy kept)"] = mean(list(valt[c].values()))
    json.dump({"kernel": rk, "per_launch_mean_by_workload": byc}, open(f"{dst}/{rnd}_pmc_render_by_workload.json", "w"), indent=1)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    # matrix-pipe occupancy of the render kernel: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel clocks); clocks from the trace's mean duration x the
    # sustained shader clock (2.1 GHz, tools/phase_profile.py)
    mfma_busy = {}
    for prec, key, durs in (("exact", rk, lean[W:nm]), ("fast", "render_rays_kernel<0, true, false", fast[min(W, 2):min(W, 2) + K])):
        for f in glob.glob(src + "/g*/p_counter_collection.csv"):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" and short(r["Kernel_Name"]).startswith(key)]
            if v and durs:
                n = n_main if prec == "exact" else len(v)
                mfma_busy[prec] = round(mean(v[:n]) / 1024.0 / (mean(durs) * 1e-6 * 2.1e9), 4)
    traffic = {
        "binding": roofline_objects(json.load(open(src + "/summary.json")), byc, grid_kernel_split(src)),
        "binding_source": f"profiles/{rnd}_pmc_summary.json, profiles/{rnd}_pmc_render_by_workload.json (field_sdf_grid_kernel split by workload: mesh export 512^3 | density grid 129^3, "
                          "whose `density_grid_kernel` entry is that kernel in its density mode)",
        "render_rays_kernel_hbm_bytes_per_launch": int(mean(main_f) * KB),
        "sds_step_hbm_bytes_per_step": int(sum(parts.values())),
        "sds_step_by_kernel": {k: int(v) for k, v in parts.items()},
        "commit": head, "profile": f"tools/prof_{rnd}.sh {tag} -> profiles/{rnd}_pmc_summary.json",
        "command": f"bench.py --steps {pb['steps']} --warmup {pb['warmup']} --sds-steps 2 --posed-frames 1 --repeat 1 under rocprofv3 --pmc (one pass per counter group)",
        "render_rays_kernel_mfma_busy_frac": mfma_busy,
        "fetch_size_note": ("FETCH_SIZE = TCC_EA0_RDREQ x 64 B on gfx950; calibrated for THIS access pattern (8-byte gathers, one 64-byte sector per L2 miss: "
                            "profiles/r01_fetch_calibration.txt) -> factor 1.0 used; the guide's x2 correction applies to wide coalesced streaming reads "
                            "(128-byte requests tallied at 64 B), which this kernel does not issue -- with x2 the figure would still be below the algorithmic bytes"),
        "_note": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) on `bench.py --steps {pb['steps']} --warmup {pb['warmup']} --sds-steps 2 "
                  f"--posed-frames 1`. render: mean FETCH_SIZE of the {n_main} main-bench dispatches of {rk} (KB x 1024; WRITE_SIZE "
                  f"{mean(write[rk][:n_main]):.0f} KB). sds_step: FETCH_SIZE + WRITE_SIZE of the HIP kernels of one step (three renders of the training "
                  "view + the kernels of ac_render_core_backward); torch's Adam / loss kernels (~0.4 GB) not included. Calibration as in round 1 "
                  "(profiles/r01_fetch_calibration.txt): for 8-byte gathers FETCH_SIZE is 64 B per L2 miss, no correction factor; memory-side counter, "
                  "Infinity-Cache hits included (L2-miss traffic, an upper bound of DRAM traffic).")}
    json.dump(traffic, open(f"{dst}/traffic.json", "w"), indent=1)
    # the bench line of the trace pass read the PREVIOUS profile's traffic.json; the copy kept next to this profile carries this profile's counters
    bench["roofline"]["traffic"] = traffic["render_rays_kernel_hbm_bytes_per_launch"]
    if isinstance(bench.get("sds_step"), dict) and "roofline" in bench["sds_step"]:
        bench["sds_step"]["roofline"]["traffic"] = traffic["sds_step_hbm_bytes_per_step"]
    bench["_traffic_fields"] = f"from the PMC passes of this profile ({tag}), see traffic.json; everything else as printed by bench.py under rocprofv3 --kernel-trace"
    json.dump(bench, open(f"{dst}/{rnd}_bench.json", "w"), indent=1)
    print(json.dumps(byw, indent=1))
    print(json.dumps({k: v for k, v in traffic.items() if k != "_note"}, indent=1))


if tag == "--refresh-binding":
    # python tools/collect_profiles.py --refresh-binding --round r05: rebuild traffic.json's `binding` block from the committed summaries of that round
    tj = json.load(open(f"{dst}/traffic.json"))
    tj["binding"] = roofline_objects(json.load(open(f"{dst}/{rnd}_pmc_summary.json")), json.load(open(f"{dst}/{rnd}_pmc_render_by_workload.json"))["per_launch_mean_by_workload"])
    tj["binding_source"] = f"profiles/{rnd}_pmc_summary.json, profiles/{rnd}_pmc_render_by_workload.json"
    json.dump(tj, open(f"{dst}/traffic.json", "w"), indent=1)
    print(json.dumps(tj["binding"], indent=1))
else:
    main()
