"""Candidate-batch loop-closure verification.

Mirrors LoopDetector::matching (radar_graph_slam/src/radar_graph_slam/loop_detector.cpp:387-441, disabled with
`#if 0` in the snapshot) and the live single-candidate check of performScanContextLoopClosure (:222-236):
the new keyframe is the TARGET, every candidate keyframe is a SOURCE; all candidates are registered in one
batched call, the fitness score of every result is evaluated in one more call, and the best converged
candidate wins if its score is below `fitness_score_thresh`.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Loop:
    candidate: int            # index into the candidate list
    relative_pose: np.ndarray  # [4,4] float32, candidate -> new keyframe (getFinalTransformation, :422)
    fitness_score: float


def verify_candidates(batch, target_cloud, candidate_clouds, guesses=None, fitness_score_max_range=float(np.finfo(np.float64).max),
                      fitness_score_thresh=0.5):
    """batch: registration.BatchAPDGICP.  Returns (Loop | None, scores, results).

    Selection rule of loop_detector.cpp:415-423: skip a candidate that did not converge or whose score is
    worse than the best so far; reject the loop when best_score > fitness_score_thresh (:431)."""
    reg = __import__("importlib").import_module("riv-slam_amd.registration")
    batch.clear()
    if not len(candidate_clouds):
        return None, np.zeros(0), None
    try:    # all clouds in ONE call (host clouds: packed by the library's host threads, one copy; device clouds: one pack launch)
        batch.set_clouds(0, [target_cloud, *candidate_clouds])
        tgt, srcs = 0, list(range(1, len(candidate_clouds) + 1))
    except ValueError:   # mixed strides / memory spaces: one by one
        batch.clear()
        tgt = batch.add_cloud(target_cloud)
        srcs = [batch.add_cloud(c) for c in candidate_clouds]
    pairs = batch.make_pairs([(s, tgt) for s in srcs], guesses)
    results = batch.align(pairs)
    scores, _ = batch.fitness(pairs, None, fitness_score_max_range)
    best_score, best = np.finfo(np.float64).max, -1
    for i in range(len(srcs)):
        if not results[i]["converged"] or scores[i] > best_score:
            continue
        best_score, best = scores[i], i
    if best < 0 or best_score > fitness_score_thresh:
        return None, scores, results
    return Loop(best, reg.result_matrix(results[best]), float(best_score)), scores, results
