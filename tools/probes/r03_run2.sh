# round 3, run 2: the group test that failed on score bit-equality, launcher mode without torch (1 rank), group host cost
set -o pipefail
mkdir -p gpurun_out/r03/group_host
run() { name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/r03/$name.json 2> gpurun_out/r03/$name.err; rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "group or exception_barrier" > gpurun_out/r03/gputests2.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/r03/gputests2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
# one rank under the real launcher, the whole N > 1 code path (file rendezvous, ncclCommInitRank, RCCL plumbing), no torch in the worker
WDBX_BENCH_FORCE_GROUP=1 run group_host/launcher_1rank_1250000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --rows 1250000 --steps 400 --warmup 40
run group_host/index_1250000 python bench.py --rows 1250000 --steps 400 --warmup 40 --no-cpu-baseline --no-other-configs
run group_host/group_1250000 python bench.py --mode group --rows 1250000 --steps 400 --warmup 40
run group_host/group_8shards_one_gpu_10m python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 100 --warmup 10
grep -h -o '"value": [0-9.]*\|"host_enqueue_p50": [0-9.]*\|"p50": [0-9.]*' gpurun_out/r03/group_host/*.json
