cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r01 -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o r01 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o r01 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -o r01 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -o r01 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tc -o r01 -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/pmc_tc.log 2>&1
find $O -name "*.csv" | head -40; tail -2 $O/*.log | cut -c1-300
