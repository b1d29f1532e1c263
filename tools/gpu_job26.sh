cd $GRAFT_REPO_ROOT; O=gpurun_out/j26; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_render.py -m gpu -q --maxfail=15 -p no:cacheprovider --durations=6 -k "render_core_operator or full_batch or without_autograd" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -12 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
cat gpurun_out/render_core_parity_4096.json
