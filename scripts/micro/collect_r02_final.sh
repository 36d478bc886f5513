# Final-build records: kernel trace summary and the two PMC traffic passes of the headline step (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_trace -o out -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/final_trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/final_fetch -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/final_fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/final_write -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/final_write.log 2>&1 || exit 1
find $R/gpurun_out/final_trace $R/gpurun_out/final_fetch $R/gpurun_out/final_write -type f | head -20
