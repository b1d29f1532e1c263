cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_occ3; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "one_launch_equals or training_form" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -3; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
for gl in 1 2 3; do AC_OCC_TRAIN_GLOG=$gl python tools/occ_train_probe.py 2>/dev/null | tail -1; done
