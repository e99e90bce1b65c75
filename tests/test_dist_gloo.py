"""Two-rank rehearsal (gloo, CPU) of the multi-GPU layer: row sharding of the keyframe descriptor
database with one all-gather of the per-shard top-2 lists, and the frame-chunk sharding of offline
batches.  The per-shard search is the CPU oracle here (no GPU in this test); the exchange and the
merge are the product code of openvslam_amd.dist."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def oracle_search(db, first_row, queries):
    import ctypes  # noqa: F401
    import oracle
    from openvslam_amd import _abi as A
    out = np.zeros(len(queries), dtype=A.KNN2_DTYPE)
    oracle.check(oracle.lib().ovo_knn2_search(A.ptr(db, A.u8p), len(db), first_row, A.ptr(queries, A.u8p), len(queries), A.ptr(out, A.knn2p)))
    return out


def make_data(seed=5):
    rng = np.random.default_rng(seed)
    db = rng.integers(0, 256, (3001, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (257, 32), dtype=np.uint8)
    # duplicates across the shard boundary: equal distances must resolve to the lower global row
    db[2000] = db[100]
    db[2999] = db[1]
    q[0] = db[100]
    q[1] = db[1]
    return db, q


def worker(rank, world, port, result_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from openvslam_amd import dist as ovd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        db, q = make_data()
        first, count = ovd.shard_rows(len(db), world, rank)
        shard = np.ascontiguousarray(db[first:first + count])
        merged = ovd.sharded_search(lambda qq: oracle_search(shard, first, qq), q, dist=dist)
        np.save(os.path.join(result_dir, "knn_%d.npy" % rank), merged)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_top2_equals_serial_scan(tmp_path, world):
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    db, q = make_data()
    want = oracle_search(db, 0, q)
    assert want["row"][0, 0] == 100 and want["row"][0, 1] == 2000 and want["dist"][0].tolist() == [0, 0]
    for r in range(world):
        got = np.load(tmp_path / ("knn_%d.npy" % r))
        assert got.tobytes() == want.tobytes(), r


def test_shard_partitions_cover_everything_once():
    from openvslam_amd import dist as ovd
    for n in (0, 1, 7, 256, 1000003):
        for world in (1, 2, 3, 8):
            parts = [ovd.shard_rows(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_bench_gpus_flag_starts_that_many_ranks():
    """`bench.py --gpus N` without a torch.distributed environment starts N rank processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (before any GPU call: --dry-run needs no GPU)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    ranks = [json.loads(line) for line in p.stderr.splitlines() if line.startswith("{")]
    assert sorted(r["rank"] for r in ranks) == [0, 1, 2]
    assert all(r["world"] == 3 and r["local_rank"] == r["rank"] for r in ranks)
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    # rank 0's line carries the roofline, the CPU baseline, the parity gate and the host-fed legs at every
    # world size, not only at one GPU
    legs = {r["rank"]: r["legs"] for r in ranks}
    assert set(legs[0]) >= {"roofline", "cpu_baseline", "parity", "pcie_inclusive", "latency_b1"} and legs[1] == legs[2] == []
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "knn", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=120)
    knn = [json.loads(line) for line in p.stderr.splitlines() if line.startswith("{")]
    assert p.returncode == 0 and set([r for r in knn if r["rank"] == 0][0]["legs"]) >= {"roofline", "cpu_baseline", "parity"}
    # a contradiction between --gpus and the environment is an error, not a silent single-rank run
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
    # a failing rank fails the run
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "nonsense"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
