cd $GRAFT_REPO_ROOT; O=gpurun_out/j18; mkdir -p $O
echo "== count"; AC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/lib_count.so COUNT=1 timeout 300 python tools/bench_warp.py 2>&1 | grep candidate | tee $O/count.txt
echo "== profile"; timeout 300 python tools/warp_profile.py 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl" | tee $O/warp_profile.txt
