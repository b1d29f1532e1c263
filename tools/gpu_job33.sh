cd $GRAFT_REPO_ROOT; O=gpurun_out/j33; mkdir -p $O
for n in head noxcd head noxcd; do
  lib=$GRAFT_REPO_ROOT/tools/_bin/lib_$n.so; [ "$n" = "head" ] && lib=$GRAFT_REPO_ROOT/avatarcraft_amd/libavatarcraft_hip.so
  echo "== $n"; AC_LIB_PATH=$lib timeout 300 python tools/bench_hashgrid.py 2>&1 | tail -5 | tee -a $O/bench_hashgrid_$n.txt
done
