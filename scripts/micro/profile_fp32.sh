# Kernel-trace summary of the fp32-mode 512x512 batch-1 step (run from the repo root on the GPU box): profile_fp32.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-fp32}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o out -- python3 $R/scripts/bench_fp32.py --steps 3 > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
DB=$(find $R/gpurun_out/prof_$TAG -name "*.db" | head -1)
python3 $R/scripts/rocpd_summary.py $DB 4 > $R/gpurun_out/${TAG}_kernel_summary.txt
rm -rf $R/gpurun_out/prof_$TAG
head -40 $R/gpurun_out/${TAG}_kernel_summary.txt
