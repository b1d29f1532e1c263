cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-t3}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "sds_step or stylize or fine_view or harness or posed_render_matches" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
python tools/posed_flip_diag.py 2>&1 | grep -v Warning | tail -20 | tee $O/posed_flip_diag.txt
python - <<'PY' 2>&1 | tail -5
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
p, field, table, ro, rd = bench.make_inputs(dev, 0)
r = bench.time_sds_fine_view(dev, p, table)
print(json.dumps({k: r[k] for k in ("ms_per_view", "phase_ms", "patch_by_patch")})); print(r["roofline"]["frac"])
PY
