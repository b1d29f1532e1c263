cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/j15_pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/j15_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/prof_r02.sh r02a 2>&1 | tail -40
