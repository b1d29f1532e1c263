cd $GRAFT_REPO_ROOT; O=gpurun_out/j22; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -14 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/j22/bench.json'))
print('render', d['value'], d['ms_per_step'], d['roofline']['frac'], 'sds', d['sds_step']['ms_per_step'], d['sds_step']['phase_ms'], 'posed', d['posed_frame']['ms_per_frame'])
PY
