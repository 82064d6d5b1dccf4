import sys, numpy as np
sys.path.insert(0, "wdbx-py_amd"); sys.path.insert(0, "oracle")
from wdbx_amd import _native
import wdbx_oracle as O
for (n, d, nq, k, metric) in [(200_000, 384, 256, 10, 0), (100_003, 100, 40, 25, 0), (70_001, 100, 70, 50, 1), (300_000, 128, 513, 5, 0), (150_000, 384, 128, 10, 1)]:
    with _native.NativeIndex(d, metric=metric, capacity_rows=n) as ix:
        ix.fill_synthetic(O.SEED_CORPUS, 0, n, normalize=True)
        dq = ix.device_queries_synthetic(O.SEED_QUERY, 0, nq, normalize=True)
        d_idx, d_score = ix.alloc(nq * k * 8), ix.alloc(nq * k * 4)
        ix.search_batch_device(dq, nq, k, d_idx, d_score)
        st0 = ix.batch_status(nq)
        r = (d_idx.download(np.int64, (nq, k)), d_score.download(np.float32, (nq, k)))
        ix.set_option("gemm_bf16", 1)
        ix.search_batch_device(dq, nq, k, d_idx, d_score)
        st = ix.batch_status(nq)
        g = (d_idx.download(np.int64, (nq, k)), d_score.download(np.float32, (nq, k)))
        print(n, d, nq, k, metric, "ids_equal", np.array_equal(g[0], r[0]), "rows_differing", int(np.sum(np.any(g[0] != r[0], axis=1))),
              "maxdiff", float(np.max(np.abs(g[1] - r[1]))), "cand fp32", float(st0["counts"].mean()), "bf16", float(st["counts"].mean()),
              "overflow", st["overflowed"], flush=True)
