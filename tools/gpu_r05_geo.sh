cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-geo}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "geometry or density_grid or host" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-400 | head -40
timeout 600 python - > $O/geo.json 2> $O/geo.err <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
p, field, table, ro, rd = bench.make_inputs(dev, 0)
print(json.dumps(bench.time_geometry(dev, p, table)))
PY
echo "geo rc $?"; tail -c 600 $O/geo.err; python -c "
import json; r=json.loads(open('$O/geo.json').read().strip().splitlines()[-1])
m=r['mesh_export_512']; print({k:m[k] for k in m if k not in ('note','roofline')}); print(m['roofline'])
d=r['density_grid_update']; print({k:d[k] for k in d if k not in ('note',)})"
