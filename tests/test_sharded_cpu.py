"""world_size-2 gloo tests (CPU) of the N>1 path: block partition of independent pairs + ONE
all-gather of the 96-byte result records (riv-slam_amd/sharded.py).  The compute engine is replaced by
the CPU oracle here (tests may do that; the product engine is the HIP batch and needs a GPU)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_block_partition(pkg):
    sh = importlib.import_module("riv-slam_amd.sharded")
    assert sh.block_partition(256, 8) == [(32 * r, 32 * r + 32) for r in range(8)]
    assert sh.block_partition(5, 2) == [(0, 3), (3, 5)]
    assert sh.block_partition(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert sh.block_partition(0, 2) == [(0, 0), (0, 0)]
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            parts = sh.block_partition(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))


def _oracle_records(indices, n_pts):
    """registers pairs `indices` with the CPU oracle and packs apdgicp_result-shaped records"""
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref as R
    scene = importlib.import_module("riv-slam_amd.scene")
    reg = importlib.import_module("riv-slam_amd.registration")
    out = np.zeros(len(indices), dtype=reg.RESULT_DTYPE)
    o = R.RefAPDGICP(R.default_params(max_correspondence_distance=2.5), num_threads=1)
    for k, p in enumerate(indices):
        s, t, _, g = scene.make_pair(n_pts, n_pts, scene.pair_seed(8, p), "odometry")
        o.setInputSource(s)
        o.setInputTarget(t)
        T = o.align(g)
        out[k]["T"] = T.T.reshape(-1)
        out[k]["converged"], out[k]["iterations"] = int(o.converged), o.nr_iterations
        out[k]["n_linearize"], out[k]["n_compute_error"] = o.n_linearize, o.n_compute_error
    return torch.from_numpy(out.view(np.uint8).reshape(len(indices), 96).copy())


def _worker(rank, world, port, n_pairs, n_pts, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    sh = importlib.import_module("riv-slam_amd.sharded")

    class OracleEngine:
        def align_block(self, indices):
            return _oracle_records(indices, n_pts)

    al = sh.ShardedBatchAligner(OracleEngine())
    assert al.my_block(n_pairs) == sh.block_partition(n_pairs, world)[rank]
    got = al.align(n_pairs)
    q.put((rank, got.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", (4, 5))
def test_two_ranks_gather_all_results_in_order(pkg, n_pairs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_pts = 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, n_pts, q)) for r in range(2)]
    [p.start() for p in procs]
    outs = dict(q.get(timeout=300) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = _oracle_records(list(range(n_pairs)), n_pts).numpy().tobytes()
    assert outs[0] == want and outs[1] == want          # every rank holds all records, global pair order
    reg = importlib.import_module("riv-slam_amd.registration")
    recs = np.frombuffer(outs[0], dtype=reg.RESULT_DTYPE)
    assert len(recs) == n_pairs and recs["n_linearize"].min() >= 1


def test_bench_spawns_its_own_ranks_before_touching_torch():
    """`python bench.py --gpus N` without a launcher (how the driver starts the scaling run): N child ranks with the
    torch.distributed environment, started by a parent that has not imported torch; only rank 0 reaches stdout."""
    import json
    import subprocess
    env = dict(os.environ, APDGICP_BENCH_SPAWN_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    assert lines == [{"rank": 0, "world": 4, "local_rank": 0, "master": "127.0.0.1", "torch_imported": False}]
    # under a launcher (WORLD_SIZE set) the same file is a plain rank and spawns nothing
    env2 = dict(env, RANK="1", WORLD_SIZE="4", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout)["rank"] == 1
