# Final-build kernel traces of the 4-NFE and 2048^2 configurations (run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_nfe4 -o out -- python3 $R/bench.py --steps 2 --warmup 1 --nfe 4 --batch 8 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/final_nfe4.log 2>&1 || exit 1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_2048 -o out -- python3 $R/bench.py --steps 1 --warmup 1 --res 2048 --batch 8 --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/final_2048.log 2>&1 || exit 1
