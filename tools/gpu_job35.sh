cd $GRAFT_REPO_ROOT; O=gpurun_out/j35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_render.py -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -4 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
for v in 0 1 0 1; do
  echo "== AC_NO_FEAT7=$v"; AC_NO_FEAT7=$v timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --sds-steps 12 --posed-frames 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['sds_step']
print('render %.4f other %.4f  sds %.4f  phases %s' % (d['roofline']['kernel_ms'], d['roofline']['other_precision']['kernel_ms'], s['ms_per_step'], s['phase_ms']))"
done
