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
rnd = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]


WHOLE_VIEW = 4          # bench.py's informational whole-view launches of the same kernel, between the main bench and the SDS steps


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
    fast = [dur(r) for r in rows if "render_rays_kernel<0, true>" in r["Kernel_Name"]]
    exact = [dur(r) for r in rows if "render_rays_kernel<0, false>" in r["Kernel_Name"]]
    # bench.py (default precision fast): W + K fast launches, then the other mode (min(W, 2) + K exact launches), WHOLE_VIEW fast launches of 65 536
    # rays, then the SDS steps (three fast renders each)
    split = {"render_rays_kernel<0, fast> main bench, timed launches": fast[W:W + K], "render_rays_kernel<0, fast> main bench, warm-up": fast[:W],
             "render_rays_kernel<0, exact> the other arithmetic mode, same launches (roofline.other_precision)": exact[min(W, 2):min(W, 2) + K],
             "render_rays_kernel<0, fast> whole view (65 536 rays) in one launch": fast[W + K:W + K + WHOLE_VIEW],
             "render_rays_kernel<0, fast> SDS step (training view: every ray hits the body)": fast[W + K + WHOLE_VIEW:]}
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
    rk = [k for k in fetch if k.startswith("render_rays_kernel<0")][0]
    main_f = fetch[rk][:n_main]
    sds_f, sds_w = fetch[rk][n_main + WHOLE_VIEW:], write[rk][n_main + WHOLE_VIEW:]
    KB = 1024
    mean = lambda v: sum(v) / len(v)
    step = 3 * (mean(sds_f) + mean(sds_w))
    parts = {"3 x render_rays_kernel": step * KB}
    for k in fetch:
        if any(s in k for s in ("hash_stencil_bwd", "bucket_acc", "sdf_stencil_bwd", "color_bwd", "composite_bwd", "core_mid", "core_normals")):
            parts[k] = (mean(fetch[k]) + mean(write[k])) * KB
    # every counter of the render kernel split by workload (summary.json averages over all dispatches of a kernel name, and the fast kernel serves
    # the main bench, the whole-view launches and the renders of the SDS steps)
    byc = {}
    for f in glob.glob(src + "/g*/p_counter_collection.csv"):
        vals = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) == rk:
                vals[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for c, v in vals.items():
            seq = [v[i] for i in sorted(v)]
            byc[c] = {"main bench (4096 rays)": mean(seq[:n_main]), "whole view (65 536 rays)": mean(seq[n_main:n_main + WHOLE_VIEW]),
                      "SDS renders (4096 rays, training view)": mean(seq[n_main + WHOLE_VIEW:])}
    json.dump({"kernel": rk, "per_launch_mean_by_workload": byc}, open(f"{dst}/{rnd}_pmc_render_by_workload.json", "w"), indent=1)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    traffic = {
        "render_rays_kernel_hbm_bytes_per_launch": int(mean(main_f) * KB),
        "sds_step_hbm_bytes_per_step": int(sum(parts.values())),
        "sds_step_by_kernel": {k: int(v) for k, v in parts.items()},
        "commit": head, "profile": f"tools/prof_r02.sh {tag} -> profiles/{rnd}_pmc_summary.json",
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


main()
