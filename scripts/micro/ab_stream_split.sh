timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/b_split1.log 2>&1 &&
LBM_NO_STREAM_SPLIT=1 timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/b_nosplit.log 2>&1 &&
timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/b_split2.log 2>&1 &&
timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --nfe 4 --batch 8 > gpurun_out/b_split_nfe4.log 2>&1 &&
LBM_NO_STREAM_SPLIT=1 timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --nfe 4 --batch 8 > gpurun_out/b_nosplit_nfe4.log 2>&1 &&
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --res 2048 --batch 8 > gpurun_out/b_split_2048.log 2>&1 &&
LBM_NO_STREAM_SPLIT=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --res 2048 --batch 8 > gpurun_out/b_nosplit_2048.log 2>&1
for f in b_split1 b_nosplit b_split2 b_split_nfe4 b_nosplit_nfe4 b_split_2048 b_nosplit_2048; do echo $f; tail -1 gpurun_out/$f.log | cut -c1-180; done
