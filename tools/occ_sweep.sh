cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "run_cuda" 2>&1 | tail -2
for g in 3 4 5 6; do
echo "AC_OCC_GLOG=$g"; AC_OCC_GLOG=$g timeout 300 python bench.py --steps 4 --sds-steps 0 --posed-frames 0 --no-cpu-baseline --sd-arch-steps 0 --repeat 1 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read())['occupancy_render']
print({k:(round(v['ms_per_view'],3) if isinstance(v,dict) and 'ms_per_view' in v else v) for k,v in r.items() if k.startswith('eval_one') or k=='samples_evaluated_per_view'})"
done
