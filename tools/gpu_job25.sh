cd $GRAFT_REPO_ROOT
bash tools/prof_r02.sh r02b 2>&1 | tail -24
