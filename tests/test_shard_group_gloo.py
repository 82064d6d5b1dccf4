"""N > 1 path on CPU: world_size-2 (and 3) process groups over gloo.  Every rank owns a
contiguous row range; the group search must equal the oracle's single-shard answer on every
rank.  The per-rank scan is the oracle here (no GPU in this container): what is under test is
the product's partitioning, exchange and merge."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, d, k, metric, out_dir):
    for p in (ROOT / "wdbx-py_amd", ROOT / "oracle"):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import wdbx_oracle as O
    from wdbx_amd.shard_group import ShardGroup, shard_row_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = O.synth_rows(O.SEED_CORPUS, 0, total, d)
    if metric == 0:
        rows = O.normalize_rows_fast(rows)
    rows[total // 3] = rows[total - 5]  # a tie that straddles shards
    begin, end = shard_row_range(total, world, rank)
    mine = rows[begin:end]

    def local_search(queries, kk):
        idx = np.full((len(queries), kk), -1, np.int64)
        sc = np.zeros((len(queries), kk), np.float32)
        for i, q in enumerate(queries):
            li, ls = O.flat_search(mine, q, kk, metric, normalize_query=False)
            idx[i, : len(li)] = li
            sc[i, : len(ls)] = ls
        return idx, sc

    def gloo_exchange(payload: bytes):
        """The test's own transport (the product has none but RCCL): an all-gather of equal-length byte strings over gloo."""
        import torch

        mine_t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        parts = [torch.empty_like(mine_t) for _ in range(world)]
        dist.all_gather(parts, mine_t)
        return [bytes(t.numpy().tobytes()) for t in parts]

    group = ShardGroup(rank, world, begin, metric, local_search=local_search, exchange=gloo_exchange)
    queries = O.synth_rows(O.SEED_QUERY, 0, 5, d)
    if metric == 0:
        queries = O.normalize_rows_fast(queries)
    queries[4] = rows[total - 5]
    g_idx, g_score = group.search(queries, k)
    for i, q in enumerate(queries):
        oi, os_ = O.flat_search(rows, q, k, metric, normalize_query=False)
        assert g_idx[i, : len(oi)].tolist() == oi.tolist(), (rank, i)
        assert g_score[i, : len(os_)].tolist() == os_.tolist(), (rank, i)
        assert np.all(g_idx[i, len(oi):] == -1)
    np.save(os.path.join(out_dir, f"idx_{rank}.npy"), g_idx)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,k,metric", [(2, 5000, 10, 0), (2, 7, 10, 0), (3, 1001, 25, 1)])
def test_shard_group_over_gloo_equals_single_shard(tmp_path, world, total, k, metric):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, 24, k, metric, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"idx_{r}.npy") for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])  # identical on every rank
