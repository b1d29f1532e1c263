cd $GRAFT_REPO_ROOT
timeout 300 python tools/ray_order_probe.py 2>&1 | tail -6
