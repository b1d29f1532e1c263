cd $GRAFT_REPO_ROOT; O=gpurun_out/j37; mkdir -p $O
timeout 300 python tools/fill_profile.py 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee $O/fill_profile.txt
