# on the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of bench.py's SDS step on prebuilt library variants (tools/build_variants.py)
#   KERNELS="binned|bucket" bash tools/prof_variants.sh head flush1 ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pvar; mkdir -p $O
BENCH_ARGS=${BENCH_ARGS:---steps 2 --warmup 1 --repeat 1 --no-cpu-baseline --sds-steps 6 --posed-frames 0 --no-occupancy --sd-arch-steps 0}
KERNELS=${KERNELS:-binned|bucket_acc|sdf_stencil_bwd|color_bwd|render_rays}
for n in "$@"; do
  lib=$R/tools/_bin/lib_$n.so; [ "$n" = "head" ] && lib=$R/avatarcraft_amd/libavatarcraft_hip.so
  rm -rf $O/kt_$n
  AC_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o p -- python $R/bench.py $BENCH_ARGS > $O/kt_$n.log 2>&1
  python - "$n" "$O/kt_$n" "$KERNELS" "$O/kt_$n.log" <<'PY'
import csv, re, sys, glob, json
n, d, pat, log = sys.argv[1:5]
f = glob.glob(d + "/**/p_kernel_stats.csv", recursive=True)
out = []
if f:
    for r in csv.DictReader(open(f[0])):
        if re.search(pat, r['Name']):
            nm = re.sub(r"\(anonymous namespace\)::|void |ac::", "", r['Name'])[:34]
            out.append("%s %.3f x%s" % (nm, float(r['TotalDurationNs']) / 1e6 / int(r['Calls']), r['Calls']))
sds = ""
try:
    line = [l for l in open(log) if l.startswith("{")][-1]; s = json.loads(line)["sds_step"]; sds = "sds %.3f %s" % (s["ms_per_step"], s["phase_ms"])
except Exception as e:
    sds = "no line: %r" % e
print("%-10s %s\n           %s" % (n, " | ".join(sorted(out)), sds))
PY
done 2>&1 | tee -a $O/summary.txt
