# where the SDS step's wall time goes BETWEEN kernels: rocprofv3 kernel trace of N steps, then per step the kernel-busy time and the idle gaps.
#   gpurun -- 'bash tools/sds_gap_probe.sh [fine]'     -> gpurun_out/sds_gaps[_fine].txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=${1:-coarse}; O=$R/gpurun_out/sds_gaps_$W; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $R/tools/sds_gap_probe.py run $W > $O/run.log 2>&1
python $R/tools/sds_gap_probe.py report $O $W | tee $R/gpurun_out/sds_gaps_$W.txt
