"""radar_graph_slam::InformationMatrixCalculator with its nearest-neighbour pass on the device (SURVEY.md 8f-1's twin).

Mirrors radar_graph_slam/src/radar_graph_slam/information_matrix_calculator.cpp:29-86 and `weight()` of
include/radar_graph_slam/information_matrix_calculator.hpp:40-43: the fitness score -- the mean squared 1-NN distance of
cloud2, moved by the relative pose, to cloud1 (PCL getFitnessScore semantics) -- comes from `FastAPDGICP.getFitnessScore`
instead of a kd-tree built on the CPU after every align; the scalar mapping to the 6x6 edge information is restated.
C++ form: riv-slam_amd/cpp/information_matrix_hip.hpp.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class InformationMatrixParams:   # defaults of the constructor (information_matrix_calculator.cpp:14-25)
    use_const_inf_matrix: bool = False
    const_stddev_x: float = 0.5
    const_stddev_q: float = 0.1
    var_gain_a: float = 20.0
    min_stddev_x: float = 0.1
    max_stddev_x: float = 5.0
    min_stddev_q: float = 0.05
    max_stddev_q: float = 0.2
    fitness_score_thresh: float = 0.5


def weight(a: float, max_x: float, min_y: float, max_y: float, x: float) -> float:
    """hpp:40-43"""
    y = (1.0 - math.exp(-a * x)) / (1.0 - math.exp(-a * max_x))
    return min_y + (max_y - min_y) * y


def information_from_fitness(p: InformationMatrixParams, fitness_score: float) -> np.ndarray:
    """the scalar part of calc_information_matrix (:29-52); w_x / w_q are floats there"""
    inf = np.zeros((6, 6))
    if p.use_const_inf_matrix:
        dx, dq = 1.0 / p.const_stddev_x, 1.0 / p.const_stddev_q
    else:
        w_x = np.float32(1.0e-8 * weight(p.var_gain_a, p.fitness_score_thresh, p.min_stddev_x ** 2, p.max_stddev_x ** 2, fitness_score))
        w_q = np.float32(1.0e-8 * weight(p.var_gain_a, p.fitness_score_thresh, p.min_stddev_q ** 2, p.max_stddev_q ** 2, fitness_score))
        dx, dq = 1.0 / float(w_x), 1.0 / float(w_q)
    inf[0, 0] = inf[1, 1] = inf[2, 2] = dx
    inf[3, 3] = inf[4, 4] = inf[5, 5] = dq
    return inf


class InformationMatrixCalculator:
    def __init__(self, params: InformationMatrixParams | None = None, device: int = 0):
        from importlib import import_module
        reg = import_module("riv-slam_amd.registration")
        self.params = params or InformationMatrixParams()
        self._g = reg.FastAPDGICP(device=device)

    def calc_fitness_score(self, cloud1, cloud2, relpose, max_range: float = float(np.finfo(np.float64).max), token1: int = 0, token2: int = 0) -> float:
        """:55-86 -- cloud1 is the target, cloud2 (transformed by relpose.cast<float>()) the source"""
        self._g.setInputTarget(cloud1, token=token1)
        self._g.setInputSource(cloud2, token=token2)
        return self._g.getFitnessScore(max_range, T=np.asarray(relpose, dtype=np.float64).astype(np.float32))

    def calc_information_matrix(self, cloud1, cloud2, relpose, token1: int = 0, token2: int = 0) -> np.ndarray:
        fs = 0.0 if self.params.use_const_inf_matrix else self.calc_fitness_score(cloud1, cloud2, relpose, token1=token1, token2=token2)
        return information_from_fitness(self.params, fs)
