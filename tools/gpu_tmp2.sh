cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-geo}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "geometry or density_grid" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-400 | head -40
python tools/grid_order_probe.py 2>&1 | tee $O/grid_order.txt
