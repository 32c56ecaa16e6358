"""The N>1 path on real hardware with ONE rank: ShardedBatchAligner over the product engine (BatchAPDGICP) and the
`nccl` backend (= RCCL).  The process group lives in a child process so that pytest's own process stays clean."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import importlib, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["APD_ROOT"])
    reg = importlib.import_module("riv-slam_amd.registration")
    scene = importlib.import_module("riv-slam_amd.scene")
    sharded = importlib.import_module("riv-slam_amd.sharded")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    P = 5
    clouds, guesses = [], []
    for p in range(P):
        s, t, _, g = scene.make_pair(2048, 2048 + 64 * p, scene.pair_seed(9, p), "odometry" if p % 2 else "loop")
        clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
        guesses.append(g)
    pairs = [(2 * i, 2 * i + 1) for i in range(P)]
    params = reg.default_params(max_correspondence_distance=2.0, transformation_epsilon=0.01, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(params, device=0)
    pairs_arr = b.make_pairs(pairs, guesses)

    class Engine:
        def align_block(self, idx):
            assert idx == list(range(P))
            b.set_clouds(0, clouds)
            return b.align_device(pairs_arr)

    al = sharded.ShardedBatchAligner(Engine())
    assert al.world == 1 and al.rank == 0
    out = al.align(P)
    torch.cuda.synchronize()
    got = out.cpu().numpy().tobytes()
    ref = reg.BatchAPDGICP(params, device=0)
    ref.set_clouds(0, clouds)
    want = ref.align(pairs_arr).tobytes()
    assert got == want, "all-gathered records differ from the synchronous batch"
    # the pipelined form bench.py uses: enqueue / collect / gather(wait=True)
    tk = b.align_enqueue(pairs_arr)
    loc = b.align_collect(tk, device=True)
    g2 = al.gather(loc, P, wait=True)
    assert g2.cpu().numpy().tobytes() == want
    print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()), "world", dist.get_world_size(), "OK")
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.gpu
def test_sharded_aligner_over_the_hip_batch_and_rccl_world_size_1(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, APD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "AssertionError" not in r.stderr and any(k in r.stderr for k in ("NCCL", "TCPStore", "Timeout", "timed out", "Address already in use")):
        # a rendezvous / communicator start-up problem of the box (seen once in ~60 runs), not a result: one more try.
        # A wrong result (AssertionError in the child) is never retried.
        with socket.socket() as s2:
            s2.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s2.getsockname()[1])
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "world 1 OK" in r.stdout


@pytest.mark.gpu
def test_bench_force_dist_prints_world_size_and_rccl_version():
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "2", "--repeats", "3", "--pairs-per-gpu", "8",
                        "--points", "2048", "--no-cpu-baseline", "--no-diagnostics"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["world_size"] == 1 and d["rccl_version"] and d["n_gpus"] == 1 and d["timing"]["repeats"] == 3
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["bound"] == "hbm"
