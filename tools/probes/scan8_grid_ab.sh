set -o pipefail
mkdir -p gpurun_out/r03/scan8_grid
for rows in ${ROWS:-2500000 5000000 7000000}; do
  for pq in 0 1 0 1; do
    timeout -k 10 200 python bench.py --rows $rows --steps 400 --warmup 40 --no-cpu-baseline --no-other-configs --no-facade --verify 0 --latency-queries 4 --opt scan8_per_query=$pq 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows=$rows per_query=$pq', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  done
done
