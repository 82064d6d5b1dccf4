"""bench.py's launcher-side plumbing that needs no GPU: the torch-free rendezvous of the ranks one launcher started
(the RCCL unique id travels through a directory named after the launcher process), and the command-line routing
(`--gpus N` as typed = the in-process shard group; RANK / WORLD_SIZE in the environment = one rank per GPU)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import bench
r = bench.FileRendezvous(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]))
uid = r.share_unique_id(lambda: bytes([7]) * bench._native.UNIQUE_ID_BYTES if int(os.environ["RANK"]) == 0 else None)
print(r.dir.name, uid.hex())
"""


def test_ranks_of_one_launcher_share_the_unique_id_without_torch(tmp_path):
    env = dict(os.environ, WORLD_SIZE="3", MASTER_PORT="29555", TMPDIR=str(tmp_path))
    procs = []
    for rank in (2, 1, 0):  # rank 0 last: the others must wait for its file
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD.format(root=str(ROOT))], env=dict(env, RANK=str(rank)),
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert len({o[0] for o in outs}) == 1, outs          # one directory: same launcher (this test process), same port
    assert {o[1] for o in outs} == {"07" * 128}          # everybody holds rank 0's id
    assert str(os.getpid()) in outs[0][0]                # named after the common parent


def test_bench_routes_gpus_n_to_the_in_process_group_and_ranks_to_the_launcher_path(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench

    called = {}
    monkeypatch.setattr(bench, "main_group", lambda args: called.setdefault("group", args.gpus))
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"])
    bench.main()
    assert called == {"group": 8}                         # `python bench.py --gpus 8` as typed starts, and starts the group
    called.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--mode", "group"])
    bench.main()
    assert called == {"group": 1}
    src = (ROOT / "bench.py").read_text()
    assert "import torch" not in src.split("def main_group")[1].split("def main():")[0]  # no torch in the group driver
