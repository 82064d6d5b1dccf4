# round 3 records, part 5: batch sizes on the i8 tiles (final code), 768-dim cosine, the async facade under concurrency,
# single queries across dimensions (selection scan vs fp32 scan), facade latency at 10 k and 1 M rows
set -o pipefail
O=gpurun_out/r03
mkdir -p $O/batch_sizes
for nq in 16 64 128 1024; do
  timeout -k 10 200 python tools/bench_batch.py 10000000 384 $nq 10 20 > $O/batch_sizes/bench_batch_$nq.json 2> $O/batch_sizes/b$nq.err || exit 1
done
timeout -k 10 200 python tools/bench_batch.py 10000000 768 256 10 10 > $O/batch_sizes/bench_batch_768_cos_256.json 2> $O/batch_sizes/b768.err || exit 1
grep -h -o '"ms_per_batch": [0-9.]*' $O/batch_sizes/bench_batch_16.json $O/batch_sizes/bench_batch_64.json $O/batch_sizes/bench_batch_128.json $O/batch_sizes/bench_batch_1024.json $O/batch_sizes/bench_batch_768_cos_256.json
timeout -k 10 300 python tools/bench_async.py > $O/bench_async_facade.json 2> $O/bench_async_facade.err; echo "async rc=$?"; tail -1 $O/bench_async_facade.json | cut -c1-600
timeout -k 10 400 python tools/bench_dims_shadow.py > $O/bench_dims_shadow.txt 2> $O/bench_dims_shadow.err; echo "dims rc=$?"
timeout -k 10 300 python tools/bench_facade.py > $O/facade_latency.json 2> $O/facade_latency.err; echo "facade rc=$?"; cat $O/facade_latency.json
