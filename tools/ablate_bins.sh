# is bucket_accumulate_kernel bound by its LDS float atomics?  (ablated build gives wrong results; timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_bins; rm -rf $O; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
for v in base nolds; do
  fl=""; case $v in nolds) fl="-DAC_ABL_NOLDSATOMIC";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$v.so $SRC > /dev/null 2>&1 &
done
wait
for v in base nolds; do
  AC_LIB_PATH=$O/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps 3 --posed-frames 0 > $O/$v.log 2>&1
  echo "== $v"; python - <<PY
import csv
for r in csv.DictReader(open("$O/$v/p_kernel_stats.csv")):
    if 'bucket_accumulate' in r['Name'] or 'binned' in r['Name']: print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e6, "ms avg")
PY
done
