cd $GRAFT_REPO_ROOT
for n in head occnf; do
lib=tools/_bin/lib_$n.so; [ "$n" = "head" ] && lib=avatarcraft_amd/libavatarcraft_hip.so
echo "$n"; AC_LIB_PATH=$GRAFT_REPO_ROOT/$lib AC_OCC_GLOG=4 timeout 300 python bench.py --steps 4 --sds-steps 0 --posed-frames 0 --no-cpu-baseline --sd-arch-steps 0 --repeat 1 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read())['occupancy_render']
print({k:(round(v['ms_per_view'],3) if isinstance(v,dict) and 'ms_per_view' in v else v) for k,v in r.items() if k.startswith('eval_') or k=='samples_evaluated_per_view'})"
done
