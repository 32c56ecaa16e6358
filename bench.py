#!/usr/bin/env python3
"""bench.py -- APD-GICP registrations/s on MI355X (BASELINE.json metric).

A "step" registers, on every GPU, one batch of P independent 8k x 8k synthetic radar scan pairs
(BASELINE configs[1] = "single scan-to-scan APD-GICP, 8k-pt source/target, 20 GN iters"; P = 32 pairs
per GPU is the per-GPU shard of configs[3], 256 pairs over 8 GPUs -> weak scaling).  The raw clouds
are resident in HBM before the timed region; a step packs them, computes BOTH clouds' k-NN covariances
(nothing cached), runs 20 Gauss-Newton iterations per pair on the device and, for N > 1, all-gathers
the 96-byte result records with RCCL.  Synthetic data, seeded (riv-slam_amd/scene.py).

  python bench.py --gpus 1 --steps 20 --warmup 10     (the defaults; a step takes 2 ms, the clocks need a few steps to settle)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PTS = 8192
GN_ITERS = 20
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def bench_params(reg):
    # configs[1]: GN, 20 iterations, never early-exit; gate / APD variances as shipped in the launch file
    return reg.default_params(optimizer=reg.OPT_GN, max_iterations=GN_ITERS, transformation_epsilon=1e-300, rotation_epsilon=1e-300,
                              max_correspondence_distance=2.0, azimuth_variance_deg=1.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs-per-gpu", type=int, default=32)
    ap.add_argument("--points", type=int, default=N_PTS)
    ap.add_argument("--handles", type=int, default=0, help="batch handles = steps kept in flight (0: 3, or 4 with a process group; 1: one "
                                                            "handle with three pair groups)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the untimed executed-flops / brute-force legs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even for one rank (self-test of the N>1 path)")
    args = ap.parse_args()

    # Three or four batch handles (one stream each) plus RCCL's stream are busy at once; the HIP runtime multiplexes the streams
    # of a process onto 4 hardware queues by default, and a fifth busy stream costs ~5 %.  Must be in the
    # environment before the runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    ge.build()
    reg = importlib.import_module("riv-slam_amd.registration")
    scene = importlib.import_module("riv-slam_amd.scene")
    sharded = importlib.import_module("riv-slam_amd.sharded")

    P, n = args.pairs_per_gpu, args.points
    total_pairs = P * world
    my_b, my_e = sharded.block_partition(total_pairs, world)[rank]
    assert my_e - my_b == P

    # ---- synthetic inputs, generated on the host, resident in HBM before timing
    d_clouds, h_pairs, guesses = [], [], []
    for p in range(my_b, my_e):
        s, t, _, g = scene.make_pair(n, n, scene.pair_seed(2, p), "odometry")
        h_pairs.append((s, t, g))
        d_clouds += [torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()]
        guesses.append(g)
    pair_idx = [(2 * i, 2 * i + 1) for i in range(P)]

    params = bench_params(reg)
    # Consecutive steps are independent batches, so several of them are kept in flight: step s runs on batch handle s % H
    # (H = --handles, 3 by default), each handle with ONE pair group = one HIP stream.  A step alone leaves the GPU
    # underfed (32 pairs: three groups of latency-bound tick kernels); with three steps at different phases one handle's
    # covariance kernels fill the gaps of the others' ticks.  Every handle registers its own copy of the step's clouds.
    H = args.handles if args.handles > 0 else (4 if use_dist else 3)   # r01, one rank: 1.32 ms per step with 3; with RCCL's stream 1.38 / 1.35 for 3 / 4
    batches = []
    for _ in range(H):
        bh = reg.BatchAPDGICP(params, device=local_rank)
        bh.set_profiling(os.environ.get("APDGICP_BENCH_NOPROF", "0") != "1")
        if H > 1:
            bh.set_pair_groups(1)
        batches.append(bh)
    batch = batches[0]
    pairs_arr = batch.make_pairs(pair_idx, guesses)
    clouds_arg = batch.pack_clouds(d_clouds)   # the pointer array a C caller would hold; the clouds themselves are re-registered every step

    class Engine:  # this rank's block through the C ABI, synchronous form (ShardedBatchAligner.align); the timed loop uses gather() only
        def align_block(self, _indices):
            batch.set_clouds(0, clouds_arg)
            return batch.align_device(pairs_arr)

    aligner = sharded.ShardedBatchAligner(Engine())
    nn_acc = [0.0, 0, 0]

    # A step = set this rank's 64 fresh clouds (packed, sorted, covariances recomputed) + register its 32 pairs + (N > 1)
    # all-gather the records.  enqueue returns without waiting (Gauss-Newton: the run length is known); a step is collected
    # -- waited for, gathered -- just before its handle is needed again, H steps later.
    def enqueue_step(bh):
        bh.set_clouds(0, clouds_arg)
        return bh.align_enqueue(pairs_arr)

    def collect_step(bh, ticket):
        local = bh.align_collect(ticket, device=True)        # zero-copy view of that step's records on the device
        out = aligner.gather(local, total_pairs, wait=True)  # (the host has about a millisecond of slack per step)
        ms, k, pr = bh.last_nn_profile()
        nn_acc[0] += ms
        nn_acc[1] += k
        nn_acc[2] += pr
        return out

    def run_steps(count):
        tickets, out = [None] * H, None
        for s in range(count):
            h = s % H
            if tickets[h] is not None:
                out = collect_step(batches[h], tickets[h])
            tickets[h] = enqueue_step(batches[h])
        for s in range(count, count + H):   # the steps still in flight, oldest first
            h = s % H
            if tickets[h] is not None:
                out = collect_step(batches[h], tickets[h])
                tickets[h] = None
        return out

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    sync_all()
    nn_acc[:] = [0.0, 0, 0]
    t0 = time.perf_counter()
    gathered = run_steps(args.steps)   # every step enqueued AND collected (and gathered) inside the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    nn_ms, nn_launches, nn_pairs = nn_acc
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_pairs * args.steps / elapsed
    ticks, nn_S, nn_T = batch.last_ticks()

    out = None
    if rank == 0:
        recs = sharded.records_from_bytes(gathered)
        assert len(recs) == total_pairs and int(recs["n_linearize"].min()) == GN_ITERS
        # ---- roofline of the nearest-neighbour kernel: ALGORITHMIC fp32 flops = 8*N*M per pair per launch
        avg_nn_ms = nn_ms / max(1, nn_launches)
        flops_per_launch = 8.0 * n * n * nn_pairs / max(1, nn_launches)   # a launch covers one pair group (P / 2 pairs)
        achieved_tf = flops_per_launch / (avg_nn_ms * 1e-3) / 1e12 if avg_nn_ms > 0 else 0.0
        # whole-registration algorithmic bytes, SURVEY 8d: B_reg = 40(N+M) + L(108N + 16M)
        b_reg = 40.0 * (2 * n) + GN_ITERS * (108.0 * n + 16.0 * n)
        hbm_gbs = b_reg * P / (ms_per_step * 1e-3) / 1e9
        pmc = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_nn_latest.json")))
        except Exception:
            pass
        out = {
            "metric": "APD-GICP registrations/s (8k-pt scan pairs, GN-20, covariances recomputed)",
            "value": round(value, 2), "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 nearest-neighbour + f64 covariance/Mahalanobis/Hessian", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1] (8k x 8k scan pair, 20 GN iterations) x {P} independent pairs per GPU per step "
                                   f"(= per-GPU shard of configs[3])", "points": n, "pairs_per_gpu": P, "gn_iterations": GN_ITERS,
                       "nn_mode": os.environ.get("APDGICP_NN_MODE", "pruned"), "nn_sources_per_lane": nn_S, "nn_target_splits": nn_T,
                       "ticks": ticks, "steps_in_flight": H, "batch_handles": H},
            "ms_per_gn_iter_batched": round(ms_per_step / GN_ITERS, 4),
            "roofline": {"kernel": "k_nn_pruned (exact fp32 nearest neighbour: Z-curve sorted clouds, bounding-box pruning, LDS-staged "
                                   "target groups)" if os.environ.get("APDGICP_NN_MODE", "pruned") != "brute" else
                                   "k_nn_partial (brute-force fp32 nearest neighbour, LDS-tiled)",
                         "bound": "mfma", "achieved": round(achieved_tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4),
                         "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
                         "avg_launch_ms": round(avg_nn_ms, 4), "launches": nn_launches,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "note": "achieved = ALGORITHMIC flops (8 per source x target point pair) / measured launch time; the pruned "
                                 "kernel returns the brute-force result bit for bit while evaluating only a few % of the pairs, so the "
                                 "algorithmic rate can exceed the 157.3 TF fp32 peak (vector == fp32-input MFMA peak); see "
                                 "executed_* and roofline_bruteforce for the rates the hardware actually sustains"},
            "roofline_hbm": {"bound": "hbm", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(hbm_gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_registration": b_reg},
        }

        if world == 1 and not args.no_diagnostics:
            # ---- outside the timed region: what the pruned kernel really executes, and the brute-force kernel on the same data
            def one_step(env):
                old = {k_: os.environ.get(k_) for k_ in env}
                os.environ.update(env)
                try:
                    bb = reg.BatchAPDGICP(params, device=local_rank)
                finally:
                    for k_, v_ in old.items():
                        if v_ is None:
                            os.environ.pop(k_, None)
                        else:
                            os.environ[k_] = v_
                bb.set_profiling(True)
                for _ in range(2):
                    bb.set_clouds(0, d_clouds)
                    bb.align_async(pairs_arr)
                    bb.synchronize()
                return bb
            if os.environ.get("APDGICP_NN_MODE", "pruned") != "brute":
                bs = one_step({"APDGICP_STATS": "1"})
                st = bs.debug_stats()      # counters of the second step only would need a reset; use per-launch averages
                pairs_per_launch = nn_pairs / max(1, nn_launches)
                waves_per_launch = (n / 64.0) * pairs_per_launch
                chunks_scanned = float(st[2]) / max(1.0, float(st[3])) * waves_per_launch   # 16-target chunk scans per launch
                executed = chunks_scanned * 16 * 64 * 8.0          # x 64 lanes (queries) x 8 flop
                out["roofline"]["executed_flops_per_launch"] = executed
                out["roofline"]["executed_fraction_of_algorithmic"] = round(executed / flops_per_launch, 5)
                out["roofline"]["executed_TFLOPs"] = round(executed / (avg_nn_ms * 1e-3) / 1e12, 2)
                del bs
            bf = one_step({"APDGICP_NN_MODE": "brute", "APDGICP_KNN_MODE": "brute", "APDGICP_STREAMS": "1", "APDGICP_PROFILE_STRIDE": "1"})
            ms_b, k_b, pr_b = bf.last_nn_profile()
            tf_b = 8.0 * n * n * pr_b / max(1e-9, ms_b * 1e-3) / 1e12
            _, sb, tb = bf.last_ticks()
            out["roofline_bruteforce"] = {"kernel": f"k_nn_partial<{sb}> (every pair evaluated, LDS-tiled, T={tb} target splits)", "bound": "mfma",
                                          "achieved": round(tf_b, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": round(tf_b / FP32_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms_b / max(1, k_b), 4),
                                          "note": "same results bit for bit; exact non-fused arithmetic (no FMA) caps this formulation at "
                                                  "~0.4 of the FMA-based spec peak (profiles/r01_ubench_valu.txt)"}
            del bf

        if world == 1:
            # ---- single-pair latency (configs[1] exactly): one handle, one registration at a time
            s, t, g = h_pairs[0]
            one = reg.FastAPDGICP(params, device=local_rank)
            ds, dt = d_clouds[0], d_clouds[1]
            for _ in range(3):
                one.setInputSource(ds), one.setInputTarget(dt), one.align(g)
            reps = 10
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                one.setInputSource(ds), one.setInputTarget(dt), one.align(g)
            single_ms = (time.perf_counter() - t1) / reps * 1e3
            # ms per GN iteration (BASELINE metric, second component): the same registration with both clouds' covariances
            # cached (pointer-equality tokens), i.e. 20 x (search + Mahalanobis + H/b + step) on device-resident data
            for _ in range(2):
                one.setInputSource(ds, token=11), one.setInputTarget(dt, token=12), one.align(g)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                one.setInputSource(ds, token=11), one.setInputTarget(dt, token=12), one.align(g)
            iter_ms = (time.perf_counter() - t1) / reps * 1e3 / GN_ITERS
            out["single_pair"] = {"ms_per_registration": round(single_ms, 3), "registrations_per_s": round(1e3 / single_ms, 1),
                                  "ms_per_gn_iteration": round(iter_ms, 4)}

            if not args.no_cpu_baseline:
                # ---- CPU baseline: the oracle's OpenMP restatement ("port") on the same pairs, bounded sample
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import ref as R
                kw = dict(optimizer=1, max_iterations=GN_ITERS, transformation_epsilon=1e-300, rotation_epsilon=1e-300,
                          max_correspondence_distance=2.0, azimuth_variance_deg=1.0)
                o = R.RefAPDGICP(R.default_params(**kw), num_threads=0)
                done, worst_t, worst_r = 0, 0.0, 0.0
                tc = time.perf_counter()
                while done < P and (time.perf_counter() - tc) < args.cpu_seconds:
                    s, t, g = h_pairs[done]
                    o.setInputSource(s), o.setInputTarget(t)
                    To = o.align(g)
                    te, re_ = scene.pose_error(To, reg.result_matrix(recs[done]))
                    worst_t, worst_r = max(worst_t, te), max(worst_r, re_)
                    done += 1
                cpu_elapsed = time.perf_counter() - tc
                out["cpu_baseline"] = {"value": round(done / cpu_elapsed, 3), "unit": "registrations/s", "cores": o.num_threads,
                                       "kind": "port", "sample": f"{done} of the {P} timed pairs (same clouds, GN-20, kd-tree + OpenMP "
                                                                 f"restatement of the reference; not the reference binary)"}
                out["parity"] = {"pairs_checked": done, "max_t_err_m": worst_t, "max_r_err_rad": worst_r, "tolerance": "1e-3 m / 1e-4 rad"}
                assert worst_t <= 1e-3 and worst_r <= 1e-4, (worst_t, worst_r)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
