cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 > gpurun_out/bench_full.log 2>&1; echo "bench exit $?"; tail -2 gpurun_out/bench_full.log
timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --force-comm > gpurun_out/bench_comm.log 2>&1; echo "comm exit $?"; tail -1 gpurun_out/bench_comm.log | cut -c1-600
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1; echo "torchrun exit $?"; tail -1 gpurun_out/bench_torchrun1.log | cut -c1-300
timeout -k 10 400 python3 bench.py --solver als --steps 2 --warmup 1 > gpurun_out/bench_als.log 2>&1; echo "als exit $?"; tail -2 gpurun_out/bench_als.log
