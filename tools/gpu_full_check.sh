cd $GRAFT_REPO_ROOT; O=gpurun_out/full_check; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log | head -2; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/prof_r03.sh ${1:-r03x} 2>&1 | tail -22 | cut -c1-200
