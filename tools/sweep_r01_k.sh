cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests -m gpu -q --timeout 200 > gpurun_out/pytest_gpu15.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu15.log
for a in "--rows 6040 --cols 3706 --nnz 1000000 --k 40" "--rows 6040 --cols 3706 --nnz 1000000 --k 40 --graph -1" "--rows 943 --cols 1682 --nnz 100000 --k 10" "--rows 943 --cols 1682 --nnz 100000 --k 10 --graph -1" "--rows 71567 --cols 10681 --nnz 10000054 --k 40" "" "--graph -1"; do timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-event-pass $a 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(\"$a\", \"|\", j[\"ms_per_step\"], \"ms\", \"%.3g nnz/s\"%j[\"value\"], j[\"test_rmse_after\"])"; done
