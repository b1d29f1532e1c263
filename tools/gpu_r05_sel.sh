# usage (through gpurun): K="multi or stylize" bash tools/gpu_r05_sel.sh <tag>      -- the -m gpu tests selected by -k "$K" (all if K is empty)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-sel}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs ${K:+-k "$K"} > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -40
