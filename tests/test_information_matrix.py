"""InformationMatrixCalculator mirror (information_matrix_calculator.cpp:29-86): scalar mapping on CPU, the device fitness
pass against a numpy nearest-neighbour evaluation on the GPU."""
import importlib
import math

import numpy as np
import pytest


def test_weight_and_information_mapping(pkg):
    im = importlib.import_module("riv-slam_amd.information_matrix")
    p = im.InformationMatrixParams()
    # hand evaluation of hpp:40-43 and cpp:39-50
    fs = 0.2
    y = (1.0 - math.exp(-20.0 * fs)) / (1.0 - math.exp(-20.0 * 0.5))
    w_x = np.float32(1e-8 * (0.1 ** 2 + (5.0 ** 2 - 0.1 ** 2) * y))
    w_q = np.float32(1e-8 * (0.05 ** 2 + (0.2 ** 2 - 0.05 ** 2) * y))
    inf = im.information_from_fitness(p, fs)
    assert inf[0, 0] == 1.0 / float(w_x) and inf[5, 5] == 1.0 / float(w_q) and np.count_nonzero(inf) == 6
    assert im.weight(20.0, 0.5, 1.0, 3.0, 0.0) == 1.0 and abs(im.weight(20.0, 0.5, 1.0, 3.0, 0.5) - 3.0) < 1e-12
    const = im.information_from_fitness(im.InformationMatrixParams(use_const_inf_matrix=True), 123.0)
    assert const[0, 0] == 2.0 and const[3, 3] == 10.0
    # monotone: a worse fit gives a weaker edge
    assert im.information_from_fitness(p, 0.4)[0, 0] < im.information_from_fitness(p, 0.1)[0, 0]


@pytest.mark.gpu
def test_fitness_pass_on_device_matches_numpy(scene, pkg):
    import apdgicp_np as O
    im = importlib.import_module("riv-slam_amd.information_matrix")
    cloud2, cloud1, T_true, _ = scene.make_pair(3000, 3500, scene.pair_seed(41, 0), "odometry")
    calc = im.InformationMatrixCalculator()
    pt = O.transform_points_f32(T_true.astype(np.float32).astype(np.float64), cloud2)
    _, sq = O.nn1(pt, cloud1)
    for max_range in (float(np.finfo(np.float64).max), 1.0):
        sel = sq.astype(np.float64) <= max_range
        want = sq[sel].astype(np.float64).mean()
        got = calc.calc_fitness_score(cloud1, cloud2, T_true, max_range, token1=1, token2=2)
        assert abs(got - want) < 1e-9 * want
    inf = calc.calc_information_matrix(cloud1, cloud2, T_true, token1=1, token2=2)
    assert np.array_equal(inf, im.information_from_fitness(calc.params, float(sq.astype(np.float64).mean())))
