cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x -k "render or model or smoke or field" 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
