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


CHILD2 = textwrap.dedent("""
    # one of TWO ranks that share GPU 0: the product engine (HIP batch) on every rank, the records gathered with gloo (two
    # ranks on one device cannot form an RCCL communicator; the partition, the engine and the gather call are the real ones)
    import importlib, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["APD_ROOT"])
    reg = importlib.import_module("riv-slam_amd.registration")
    scene = importlib.import_module("riv-slam_amd.scene")
    sharded = importlib.import_module("riv-slam_amd.sharded")
    rank, world, P = int(os.environ["RANK"]), 2, int(os.environ["APD_PAIRS"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    params = reg.default_params(max_correspondence_distance=2.0, transformation_epsilon=0.01, azimuth_variance_deg=1.0)
    b = reg.BatchAPDGICP(params, device=0)

    class Engine:
        def align_block(self, idx):
            clouds, guesses = [], []
            for p in idx:
                s, t, _, g = scene.make_pair(1500 + 100 * p, 1800, scene.pair_seed(12, p), "loop" if p % 2 else "odometry")
                clouds += [s, t]
                guesses.append(g)
            if not idx:
                return torch.zeros((0, 96), dtype=torch.uint8)
            b.set_clouds(0, clouds)
            return torch.from_numpy(b.align([(2 * i, 2 * i + 1) for i in range(len(idx))], guesses).view(np.uint8).reshape(len(idx), 96).copy())

    al = sharded.ShardedBatchAligner(Engine())
    assert al.world == 2 and al.my_block(P) == sharded.block_partition(P, 2)[rank]
    out = al.align(P)
    if rank == 1:     # the rank that did NOT register the first block holds everything too
        open(os.environ["APD_OUT"], "wb").write(out.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.gpu
@pytest.mark.parametrize("n_pairs", (6, 5))
def test_two_ranks_share_one_gpu_with_the_product_engine(tmp_path, scene, n_pairs):
    """World size 2 on the hardware there is: two processes, each with its own HIP batch handle on GPU 0, block partition,
    one all-gather of the 96-byte records (gloo: RCCL refuses two ranks on one device).  Every record must equal the one a
    single handle computes for the same pair."""
    import importlib
    import numpy as np
    reg = importlib.import_module("riv-slam_amd.registration")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script, out_path = tmp_path / "child2.py", tmp_path / "gathered.bin"
    script.write_text(CHILD2)
    procs = []
    for r in range(2):
        env = dict(os.environ, APD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE="2", APD_PAIRS=str(n_pairs),
                   APD_OUT=str(out_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-3000:] for o in outs)
    got = np.frombuffer(open(out_path, "rb").read(), dtype=reg.RESULT_DTYPE)
    assert len(got) == n_pairs
    params = reg.default_params(max_correspondence_distance=2.0, transformation_epsilon=0.01, azimuth_variance_deg=1.0)
    single = reg.BatchAPDGICP(params)
    clouds, guesses = [], []
    for p in range(n_pairs):
        s_, t_, _, g = scene.make_pair(1500 + 100 * p, 1800, scene.pair_seed(12, p), "loop" if p % 2 else "odometry")
        clouds += [s_, t_]
        guesses.append(g)
    single.set_clouds(0, clouds)
    want = single.align([(2 * i, 2 * i + 1) for i in range(n_pairs)], guesses)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("optimizer", ("gn", "lm"))
def test_bench_timed_loop_with_two_ranks_on_one_gpu(tmp_path, scene, optimizer):
    """bench.py's OWN N > 1 loop -- rank spawn, steps kept in flight, the gather of a step waited for one round later,
    barrier + synchronise around the timed region, all_reduce(MAX) of the repetition times -- with world size 2 on the one
    GPU there is (`--ranks-share-gpu --dist-backend gloo`; the 8-GPU run goes through exactly this code with RCCL).  One JSON
    line, world_size 2, and the gathered records of both ranks' blocks equal those of a single handle."""
    import importlib
    import json
    import numpy as np
    reg = importlib.import_module("riv-slam_amd.registration")
    sys.path.insert(0, ROOT)
    import bench
    P, n = 4, 2048
    dump = tmp_path / "records.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ranks-share-gpu", "--dist-backend", "gloo", "--steps", "5",
                        "--warmup", "2", "--repeats", "2", "--pairs-per-gpu", str(P), "--points", str(n), "--optimizer", optimizer,
                        "--kind", "loop" if optimizer == "lm" else "odometry", "--no-cpu-baseline", "--no-diagnostics", "--dump-records", str(dump)],
                       capture_output=True, text=True, timeout=900, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["world_size"] == 2 and d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["ranks_share_gpu"] is True
    assert d["config"]["pairs_per_gpu"] == P and d["timing"]["repeats"] == 2 and d["value"] > 0
    assert abs(d["value"] - 2 * P / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]   # whole job: both ranks' pairs per step
    got = np.frombuffer(np.load(dump).tobytes(), dtype=reg.RESULT_DTYPE)
    assert len(got) == 2 * P
    # the same 2 P pairs (bench.py seeds pair p of the JOB with pair_seed(2, p)) on one handle
    kind = "loop" if optimizer == "lm" else "odometry"
    single = reg.BatchAPDGICP(bench.bench_params(reg, optimizer))
    clouds, guesses = [], []
    for p in range(2 * P):
        s_, t_, _, g = scene.make_pair(n, n, scene.pair_seed(2, p), kind)
        clouds += [s_, t_]
        guesses.append(np.eye(4, dtype=np.float32) if kind == "loop" else g)
    single.set_clouds(0, clouds)
    want = single.align([(2 * i, 2 * i + 1) for i in range(2 * P)], guesses)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
@pytest.mark.timeout(1500)
@pytest.mark.parametrize("optimizer", ("lm", "gn"))
def test_c4_whole_256_pairs_over_8_ranks(tmp_path, scene, optimizer):
    """BASELINE configs[3] as a whole: 256 pairwise registrations of 8 192-point clouds, block-partitioned 32 per rank over EIGHT
    ranks (loop_detector.cpp:404-423 is the workload), through bench.py's own N > 1 launch path -- rank spawn, pooled / pipelined
    steps, one all-gather of 96-byte records per step, barrier, all_reduce(MAX).  The box has one GPU, so the eight ranks share
    it and gather with gloo (RCCL refuses several ranks on one device); partition, engine and gather call are the ones the 8-GPU
    run takes.  The 256 gathered records must equal, byte for byte, those of ONE handle registering all 256 pairs, and 16 of
    them (two per rank's block) are checked against the CPU oracle: counts exact, pose inside the north-star tolerance."""
    import importlib
    import json
    import numpy as np
    import ref as R
    reg = importlib.import_module("riv-slam_amd.registration")
    sys.path.insert(0, ROOT)
    import bench
    W, P, n = 8, 32, 8192
    kind = "loop" if optimizer == "lm" else "odometry"
    dump = tmp_path / "records.npy"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(W), "--ranks-share-gpu", "--dist-backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--repeats", "1", "--pairs-per-gpu", str(P), "--points", str(n), "--optimizer", optimizer, "--kind", kind,
                        "--no-cpu-baseline", "--no-diagnostics", "--dump-records", str(dump)],
                       capture_output=True, text=True, timeout=1200, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["world_size"] == W and d["n_gpus"] == W and d["config"]["pairs_per_gpu"] == P and d["scaling"] == "weak"
    assert abs(d["value"] - W * P / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    # what the first real SCALE run will be read by: every rank's own time, the imbalance, the gather's latency, the efficiency field
    mr = d["multi_rank"]
    assert mr["ranks"] == W and len(mr["ms_per_step_per_rank"]) == W and 0 < mr["ms_per_step_min"] <= mr["ms_per_step_max"]
    assert mr["ms_per_step_max"] <= d["ms_per_step"] * 1.0001 + 1e-6 and mr["imbalance"] >= 0.0
    assert mr["gather_ms"]["median"] > 0 and mr["weak_scaling_efficiency"]["value"] is None     # (ranks share the GPU: not a scaling point, said so)
    assert "concurrency_note" in d["roofline"] and "kernel_time_check" not in d["roofline"] and "variant:" in d["library_build_flags"]
    got = np.frombuffer(np.load(dump).tobytes(), dtype=reg.RESULT_DTYPE)
    assert len(got) == W * P
    params = bench.bench_params(reg, optimizer)
    single = reg.BatchAPDGICP(params)
    clouds, guesses = [], []
    for p in range(W * P):
        s_, t_, _, g = scene.make_pair(n, n, scene.pair_seed(2, p), kind)
        clouds += [s_, t_]
        guesses.append(np.eye(4, dtype=np.float32) if kind == "loop" else g)
    single.set_clouds(0, clouds)
    want = single.align([(2 * i, 2 * i + 1) for i in range(W * P)], guesses)
    assert got.tobytes() == want.tobytes(), int((got["T"] != want["T"]).any(axis=1).sum())
    rp = R.default_params(**{f: getattr(params, f) for f, _ in reg.Params._fields_})
    worst = [0.0, 0.0]
    for p in [q for rank in range(W) for q in (rank * P + 3, rank * P + P - 1)]:
        o = R.RefAPDGICP(rp)
        o.setInputSource(clouds[2 * p])
        o.setInputTarget(clouds[2 * p + 1])
        To = o.align(guesses[p])
        rec = got[p]
        assert [int(rec["converged"]), int(rec["iterations"]), int(rec["n_linearize"]), int(rec["n_compute_error"])] == \
            [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error], p
        te, re_ = scene.pose_error(To, rec["T"].reshape(4, 4).T)
        worst = [max(worst[0], te), max(worst[1], re_)]
        assert te <= 1e-3 and re_ <= 1e-4, (p, te, re_)
    print(f"C4 whole ({optimizer}): 256 records byte-equal to one handle; 16 oracle checks, worst pose difference {worst[0]:.2e} m / {worst[1]:.2e} rad; "
          f"{d['value']:.0f} registrations/s with 8 ranks sharing the GPU")
