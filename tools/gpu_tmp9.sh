cd $GRAFT_REPO_ROOT
for v in "" phnofield; do if [ -n "$v" ]; then export AC_LIB_PATH=$PWD/tools/_bin/lib_$v.so; fi; python tools/occ_train_probe.py 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$v', {k:(round(v['gpu_ms'],3) if isinstance(v,dict) and 'gpu_ms' in v else v) for k,v in r.items() if k.startswith('eval') or k=='view_samples'})"; done
