# calibration of FETCH_SIZE on the renderer's own access pattern: 8-byte gathers from tables of 64 MB .. 16 KB (tools/gather_bench.hip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_gather; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o p -- $R/tools/_bin/gather_bench > $O/f.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/t -o p -- $R/tools/_bin/gather_bench > $O/t.log 2>&1
python - <<PY
import csv, collections
def load(f):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return d
a = load("$O/f/p_counter_collection.csv"); b = load("$O/t/p_counter_collection.csv")
lanes = 2048 * 256 * 128 * 8
print("per kernel variant (dispatches in launch order: tables of 64 MB, 32 MB, 2 MB, 16 KB; 4 repetitions each): FETCH_SIZE KB, bytes per lane-load, L1->L2 requests per lane-load, L2 hit rate")
for k in a:
    if 'gather' not in k: continue
    fs = a[k]['FETCH_SIZE']; rq = b[k]['TCP_TCC_READ_REQ_sum']; h = b[k]['TCC_HIT_sum']; m = b[k]['TCC_MISS_sum']
    for t in range(4):
        sl = slice(4 * t, 4 * t + 4)
        f_ = sum(fs[sl]) / 4; r_ = sum(rq[sl]) / 4; h_ = sum(h[sl]) / 4; m_ = sum(m[sl]) / 4
        print("%-22s table %d  FETCH %10.0f KB  %6.1f B/lane  req/lane %.3f  L2 hit %.3f  miss/lane %.3f  FETCH B per L2 miss %.1f" % (k[:22], t, f_, f_ * 1024 / lanes, r_ / lanes, h_ / max(h_ + m_, 1), m_ / lanes, f_ * 1024 / max(m_, 1)))
PY
