# Kernel-trace summary of the batch-8 training step (run from the repo root on the GPU box): profile_train.sh <tag> [resolution]
R=$GRAFT_REPO_ROOT
TAG=${1:-train}
RES=${2:-512}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o out -- python3 $R/scripts/bench_train.py --steps 4 --warmup 1 --res $RES > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
DB=$(find $R/gpurun_out/prof_$TAG -name "*.db" | head -1)
python3 $R/scripts/rocpd_summary.py $DB 5 > $R/gpurun_out/${TAG}_kernel_summary.txt
rm -rf $R/gpurun_out/prof_$TAG
head -50 $R/gpurun_out/${TAG}_kernel_summary.txt
