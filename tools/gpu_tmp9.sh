cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_occ8; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "run_cuda or occupancy" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -3; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
for nl in 4 3 5; do AC_OCC_NLOG=$nl python tools/occ_train_probe.py 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print($nl, {k:(round(v['gpu_ms'],3) if isinstance(v,dict) and 'gpu_ms' in v else v) for k,v in r.items() if k.startswith('eval') or k=='view_samples'})"; done
