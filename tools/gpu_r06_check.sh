# the whole -m gpu suite + smoke + the default bench line in one GPU job:   gpurun -- 'bash tools/gpu_r06_check.sh <tag> [pytest -k expression]'
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_${1:-check}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs ${2:+-k "$2"} > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/bench.err
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value",r["value"],"ms/step",r["ms_per_step"],"frac",r["roofline"]["frac"],"kern",r["roofline"]["kernel_ms"])
s=r.get("sds_step",{}); print("sds",s.get("ms_per_step"),s.get("phase_ms"),s.get("error"))
f=r.get("sds_view_fine",{}); print("fine",f.get("ms_per_view"),f.get("phase_ms"),(f.get("patch_by_patch") or {}).get("ms_per_view"),(f.get("roofline") or {}).get("frac"),f.get("error"))
p=r.get("posed_frame",{}); print("posed",p.get("ms_per_frame"),p.get("error"))
m=r.get("mesh_export_512",{}); print("mesh",{k:m.get(k) for k in ("ms","sdf_grid_ms","marching_cubes_ms","error")}); print("dg",(r.get("density_grid_update") or {}).get("ms"))
print("vd",r.get("viewdirs")); print("occ",json.dumps(r.get("occupancy_render"))[:300])
PY
