cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in 15 8; do
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/pmc_rowa_a_$t -o p -- python3 $R/scripts/one_shape.py $t conv128 > $R/gpurun_out/pmc_rowa_a_$t.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $R/gpurun_out/pmc_rowa_b_$t -o p -- python3 $R/scripts/one_shape.py $t conv128 > $R/gpurun_out/pmc_rowa_b_$t.log 2>&1 || exit 1
done
for d in $R/gpurun_out/pmc_rowa_*_*/; do echo "== $d"; f=$(find $d -name "*counter_collection.csv" | head -1); python3 $R/scripts/pmc_sum.py $f; done
