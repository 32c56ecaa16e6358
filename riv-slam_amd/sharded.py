"""Loop-closure candidate batches sharded over the GPUs of one node.

Independent (source, target) pairs -- the candidates of LoopDetector::detect
(radar_graph_slam/src/radar_graph_slam/loop_detector.cpp:222-236, and the one-target-many-sources loop
at :404-423) -- are block-partitioned over ranks: pair p belongs to rank p // ceil(P / world).  Each
rank registers its block on its own GPU with `BatchAPDGICP`; there is no collective on the data path.
The only exchange is ONE all-gather of the fixed-size result records (96 B per pair) so that every rank
(in particular the one that owns the pose graph) sees all results: `torch.distributed` with backend
"nccl" is RCCL over xGMI; the payload is a few KB, i.e. latency-bound, so a single call per batch.
"""
from __future__ import annotations

import numpy as np

RESULT_BYTES = 96


def block_partition(n_pairs: int, world: int):
    """[(begin, end)] per rank; contiguous blocks of ceil(P/world) pairs (last ranks may be short/empty)."""
    per = (n_pairs + world - 1) // world if n_pairs > 0 else 0
    return [(min(r * per, n_pairs), min((r + 1) * per, n_pairs)) for r in range(world)]


class ShardedBatchAligner:
    """engine: object with `align_block(pair_indices) -> torch.uint8 tensor [len, 96]` living on the
    device the process group communicates from (CUDA tensor for nccl/RCCL, CPU tensor for gloo)."""

    def __init__(self, engine, group=None):
        import torch.distributed as dist
        self.engine = engine
        self.group = group
        self.dist = dist
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def my_block(self, n_pairs: int):
        return block_partition(n_pairs, self.world)[self.rank]

    def align(self, n_pairs: int):
        """Registers this rank's block and all-gathers the records.  Returns a uint8 tensor
        [n_pairs, 96] in global pair order on every rank (valid until the next call: the buffer is reused)."""
        b, e = block_partition(n_pairs, self.world)[self.rank]
        return self.gather(self.engine.align_block(list(range(b, e))), n_pairs)

    def gather(self, local, n_pairs: int, wait: bool = False):
        """The exchange step alone: `local` = this rank's records of a batch of n_pairs ([len(my_block), 96] uint8, on the
        device the group communicates from).  For callers that keep two batches in flight (BatchAPDGICP.align_enqueue /
        align_collect) and gather batch s while batch s+1 runs; wait=True returns only when the collective has finished
        reading `local` (the engine reuses that buffer two batches later)."""
        import torch
        out = self._gather(local, n_pairs)
        if wait and self.dist.is_initialized() and getattr(local, "is_cuda", False):
            torch.cuda.current_stream(local.device).synchronize()
        return out

    def _gather(self, local, n_pairs: int):
        import torch
        parts = block_partition(n_pairs, self.world)
        b, e = parts[self.rank]
        per = parts[0][1] - parts[0][0]
        if self.world == 1 and not self.dist.is_initialized():
            return local
        # fixed-size contribution per rank -> one all_gather_into_tensor; the buffers are allocated once per batch shape
        key = (per, self.world, str(local.device))
        if getattr(self, "_key", None) != key:
            self._key = key
            self._out = torch.empty((self.world * per, RESULT_BYTES), dtype=torch.uint8, device=local.device)
            self._pad = torch.zeros((per, RESULT_BYTES), dtype=torch.uint8, device=local.device)
        even = all(pe - pb == per for pb, pe in parts)
        if even and local.is_contiguous():  # every rank holds a full block: no staging copy, no trimming
            self.dist.all_gather_into_tensor(self._out, local, group=self.group)
            return self._out
        if e > b:
            self._pad[: e - b] = local
        self.dist.all_gather_into_tensor(self._out, self._pad, group=self.group)
        return torch.cat([self._out[r * per: r * per + (pe - pb)] for r, (pb, pe) in enumerate(parts)], dim=0)


def records_from_bytes(t) -> np.ndarray:
    """uint8 tensor [n, 96] -> structured numpy array (registration.RESULT_DTYPE layout)."""
    from importlib import import_module
    reg = import_module("riv-slam_amd.registration")
    a = t.detach().cpu().numpy()
    return np.frombuffer(a.tobytes(), dtype=reg.RESULT_DTYPE)
