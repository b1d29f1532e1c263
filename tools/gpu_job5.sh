cd $GRAFT_REPO_ROOT; O=gpurun_out/j5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -8 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
for prec in fast exact; do for i in 1 2; do timeout 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --sds-steps 8 --posed-frames 2 --precision $prec 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['sds_step']; p = d['posed_frame']
print('$prec', 'render %.4f ms frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']), 'sds', s.get('ms_per_step'), s.get('phase_ms'), 'posed', p.get('ms_per_frame'))
"; done; done
cat gpurun_out/fast_vs_exact.json
