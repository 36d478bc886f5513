# Final-build records of round 3: kernel-trace summaries (headline, 4-NFE, 2048^2) and the two PMC traffic passes of the
# headline step (run from the repo root on the GPU box: bash scripts/micro/collect_r03_final.sh)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-roofline --no-extras"
# kernel traces on ONE stream (LBM_NO_STREAM_SPLIT=1: a launch's duration is then its own; with the two-stream split
# concurrent kernels stretch each other and the durations sum to more than the step)
export LBM_NO_STREAM_SPLIT=1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/f3_trace -o out -- python3 $R/bench.py --steps 4 --warmup 1 $B > $R/gpurun_out/f3_trace.log 2>&1 || exit 1
python3 $R/scripts/rocpd_summary.py $(find $R/gpurun_out/f3_trace -name "*.db" | head -1) 6 > $R/gpurun_out/r03_bench_b16_1024_kernel_summary.txt || exit 1
rm -rf $R/gpurun_out/f3_trace
unset LBM_NO_STREAM_SPLIT
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/f3_fetch -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 $B > $R/gpurun_out/f3_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/f3_write -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 $B > $R/gpurun_out/f3_write.log 2>&1 || exit 1
python3 $R/scripts/pmc_traffic.py $(find $R/gpurun_out/f3_fetch -name "*counter_collection.csv" | head -1) $(find $R/gpurun_out/f3_write -name "*counter_collection.csv" | head -1) 3 > $R/gpurun_out/r03_pmc_igemm_step_traffic.json || exit 1
rm -rf $R/gpurun_out/f3_fetch $R/gpurun_out/f3_write
export LBM_NO_STREAM_SPLIT=1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/f3_nfe4 -o out -- python3 $R/bench.py --steps 2 --warmup 1 --nfe 4 --batch 8 $B > $R/gpurun_out/f3_nfe4.log 2>&1 || exit 1
python3 $R/scripts/rocpd_summary.py $(find $R/gpurun_out/f3_nfe4 -name "*.db" | head -1) 4 > $R/gpurun_out/r03_nfe4_kernel_summary.txt
rm -rf $R/gpurun_out/f3_nfe4
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/f3_2048 -o out -- python3 $R/bench.py --steps 1 --warmup 1 --res 2048 --batch 8 $B > $R/gpurun_out/f3_2048.log 2>&1 || exit 1
python3 $R/scripts/rocpd_summary.py $(find $R/gpurun_out/f3_2048 -name "*.db" | head -1) 3 > $R/gpurun_out/r03_res2048_kernel_summary.txt
rm -rf $R/gpurun_out/f3_2048
head -25 $R/gpurun_out/r03_bench_b16_1024_kernel_summary.txt; cat $R/gpurun_out/r03_pmc_igemm_step_traffic.json
