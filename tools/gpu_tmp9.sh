cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_occ9; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "run_cuda or occupancy or viewdirs or raymarch or march or pins" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -3; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
timeout 600 python tools/soak.py 0.2 2>&1 | grep "occupancy\|total\|hand-off"
