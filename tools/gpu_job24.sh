cd $GRAFT_REPO_ROOT; O=gpurun_out/j24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -5 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --posed-frames 1 --sds-steps 16 2>/dev/null | tail -1 > $O/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/j24/bench.json'))
print('render', d['value'], d['ms_per_step'], d['roofline']['frac'], 'sds', d['sds_step']['ms_per_step'], d['sds_step']['phase_ms'], 'posed', d['posed_frame']['ms_per_frame'])
PY
