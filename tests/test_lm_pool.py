"""GPU tests (-m gpu) of the pooled Levenberg-Marquardt batches (Engine::pool_*, k_pool_poll): LM is the reference's only
reachable optimiser (lsq_registration_impl.hpp:17) and loop-closure candidates are aligned from the identity
(loop_detector.cpp:222-225), so their run lengths differ widely inside one batch.  The pool keeps several batches in flight
on ONE handle and ONE host thread; every record must be byte-identical to the host-polled loop (APDGICP_LM_POOL=0) and agree
with the CPU oracle."""
import importlib
import os

import numpy as np
import pytest

import ref as R

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("APDGICP_LM_POOL", "1") == "0" or os.environ.get("APDGICP_NN_MODE") == "brute",
                                 reason="tools/knob_matrix.sh row without the pair pool: these tests compare the pool WITH the host-polled loop")]

LM = dict(max_correspondence_distance=2.0, transformation_epsilon=0.1, azimuth_variance_deg=1.0)
T_TOL, R_TOL = 1e-3, 1e-4


@pytest.fixture(scope="module")
def reg():
    import __graft_entry__ as g
    g.build()
    return importlib.import_module("riv-slam_amd.registration")


def polled_align(b, pairs, guesses):
    """the same batch through run_align's host-polled loop (the round-2 path, APDGICP_LM_POOL=0): the cross-check"""
    os.environ["APDGICP_LM_POOL"] = "0"
    try:
        return b.align(pairs, guesses).copy()
    finally:
        os.environ.pop("APDGICP_LM_POOL", None)


def loop_batches(scene, n_batches, n_pairs, n_pts, seed0):
    data = []
    for s in range(n_batches):
        clouds, guesses = [], []
        for p in range(n_pairs):
            a, b_, _, _ = scene.make_pair(n_pts + 13 * p, n_pts, scene.pair_seed(seed0 + s, p), "loop")
            clouds += [a, b_]
            guesses.append(np.eye(4, dtype=np.float32))   # loop_detector.cpp:225
        data.append((clouds, guesses))
    return data


def test_pool_records_equal_the_host_polled_loop_and_the_oracle(reg, scene):
    n_pairs = 10
    (clouds, guesses), = loop_batches(scene, 1, n_pairs, 1500, 300)
    pair_idx = [(2 * i, 2 * i + 1) for i in range(n_pairs)]
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    b.set_clouds(0, clouds)
    got = b.align(pair_idx, guesses)
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    ref_b.set_clouds(0, clouds)
    want = polled_align(ref_b, pair_idx, guesses)
    assert got.tobytes() == want.tobytes()
    assert len(set(int(x) for x in got["n_linearize"])) > 2      # the run lengths really differ inside the batch
    for p in range(n_pairs):
        o = R.RefAPDGICP(R.default_params(**LM))
        o.setInputSource(clouds[2 * p]), o.setInputTarget(clouds[2 * p + 1])
        To = o.align(guesses[p])
        te, re_ = scene.pose_error(To, reg.result_matrix(got[p]))
        assert te <= T_TOL and re_ <= R_TOL, (p, te, re_)
        assert [got[p]["converged"], got[p]["iterations"], got[p]["n_linearize"], got[p]["n_compute_error"]] == \
            [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error], p


def test_four_batches_in_flight_on_one_handle(reg, scene):
    """Batches of different sizes in disjoint cloud slots, enqueued back to back, collected out of order, in host and device
    form; afterwards the slots are reused (set_clouds waits for the batch that still reads them)."""
    sizes = (7, 12, 3, 9, 12, 5)
    data = [loop_batches(scene, 1, n, 900 + 100 * (k % 3), 320 + k)[0] for k, n in enumerate(sizes)]
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    want = []
    for (clouds, guesses), n in zip(data, sizes):
        ref_b.set_clouds(0, clouds)
        want.append(polled_align(ref_b, [(2 * i, 2 * i + 1) for i in range(n)], guesses))
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    base, tickets = 0, []
    for k in range(4):
        clouds, guesses = data[k]
        b.set_clouds(base, clouds)
        tickets.append(b.align_enqueue([(base + 2 * i, base + 2 * i + 1) for i in range(sizes[k])], guesses))
        base += len(clouds)
    for k in (2, 0, 3, 1):
        if k == 3:
            assert b.align_collect(tickets[k], device=True).cpu().numpy().tobytes() == want[k].tobytes()
        assert b.align_collect(tickets[k]).tobytes() == want[k].tobytes(), k
    # slots 0.. are free again; batch 4 goes there while batch 5 reuses them at once: set_clouds waits for batch 4
    b.set_clouds(0, data[4][0])
    t4 = b.align_enqueue([(2 * i, 2 * i + 1) for i in range(sizes[4])], data[4][1])
    b.set_clouds(0, data[5][0])
    t5 = b.align_enqueue([(2 * i, 2 * i + 1) for i in range(sizes[5])], data[5][1])
    assert b.align_collect(t5).tobytes() == want[5].tobytes()
    assert b.align_collect(t4).tobytes() == want[4].tobytes()
    assert b.align([(2 * i, 2 * i + 1) for i in range(sizes[5])], data[5][1]).tobytes() == want[5].tobytes()


@pytest.mark.parametrize("lanes", (8, 24))
def test_more_batches_than_lanes_and_a_growing_pool(reg, scene, monkeypatch, lanes):
    """lanes + 2 enqueues without a collect: the oldest batches give their lanes up (their tickets are void), the newest
    `lanes` stay collectable; then a batch with more pairs and larger clouds than the pool was laid out for."""
    if lanes != 24:
        monkeypatch.setenv("APDGICP_POOL_LANES", str(lanes))   # (read when the pool is laid out; twenty-four by default)
    else:
        monkeypatch.delenv("APDGICP_POOL_LANES", raising=False)
    data = loop_batches(scene, lanes + 2, 4, 800, 340)
    pair_idx = [(2 * i, 2 * i + 1) for i in range(4)]
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    want = []
    for clouds, guesses in data:
        ref_b.set_clouds(0, clouds)
        want.append(polled_align(ref_b, pair_idx, guesses))
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    tickets = []
    for s, (clouds, guesses) in enumerate(data):
        b.set_clouds(8 * s, clouds)
        tickets.append(b.align_enqueue([(8 * s + 2 * i, 8 * s + 2 * i + 1) for i in range(4)], guesses))
    order = [lanes + 1, 4, 7, 2, 3, 5, 6, 8] + [s for s in range(9, lanes + 1)]
    assert sorted(order) == list(range(2, lanes + 2))
    for s in order:
        assert b.align_collect(tickets[s]).tobytes() == want[s].tobytes(), s
    for s in (0, 1):
        with pytest.raises(Exception, match="ticket"):
            b.align_collect(tickets[s])
    (clouds, guesses), = loop_batches(scene, 1, 9, 2600, 350)
    b.set_clouds(0, clouds)
    got = b.align([(2 * i, 2 * i + 1) for i in range(9)], guesses)
    ref_b.set_clouds(0, clouds)
    assert got.tobytes() == polled_align(ref_b, [(2 * i, 2 * i + 1) for i in range(9)], guesses).tobytes()
    # the pool has laid itself out anew for that batch (larger segments): the DEVICE records of a ticket from before are a copy
    # of its host records now, not a pointer into the old layout
    newest = lanes + 1
    dev = b.align_collect(tickets[newest], device=True)
    assert bytes(dev.cpu().numpy().tobytes()) == want[newest].tobytes()
    assert b.align_collect(tickets[newest]).tobytes() == want[newest].tobytes()


def test_a_failing_batch_fails_alone_and_the_handle_stays_usable(reg, scene):
    data = loop_batches(scene, 3, 5, 1000, 360)
    pair_idx = lambda base: [(base + 2 * i, base + 2 * i + 1) for i in range(5)]  # noqa: E731
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    b.set_clouds(0, data[0][0])
    want0 = b.align(pair_idx(0), data[0][1]).copy()
    bad = [c.copy() for c in data[1][0]]
    bad[4][17] = np.nan
    b.set_clouds(10, bad)
    t_bad = b.align_enqueue(pair_idx(10), data[1][1])
    with pytest.raises(Exception, match="non-finite"):
        b.align_collect(t_bad)
    b.set_clouds(10, data[1][0])                      # the same slots, healthy clouds
    t1 = b.align_enqueue(pair_idx(10), data[1][1])
    t0 = b.align_enqueue(pair_idx(0), data[0][1])   # clouds cached since the first align
    assert b.align_collect(t0).tobytes() == want0.tobytes()
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    ref_b.set_clouds(0, data[1][0])
    assert b.align_collect(t1).tobytes() == polled_align(ref_b, pair_idx(0), data[1][1]).tobytes()
    # parameters change between batches: the pool drains, the next batch runs with the new ones; GN leaves the pool
    b.set_params(reg.default_params(optimizer=1, max_iterations=3, transformation_epsilon=1e-300, rotation_epsilon=1e-300, max_correspondence_distance=2.0))
    gn = b.align(pair_idx(0), data[0][1])
    assert int(gn["n_linearize"].min()) == 3
    b.set_params(reg.default_params(**LM))
    assert b.align(pair_idx(0), data[0][1]).tobytes() == want0.tobytes()


def test_c4_shard_at_full_size(reg, scene):
    """BASELINE configs[3], one GPU's shard as SURVEY 8d specifies it: 32 loop-closure candidates at 8192 points, identity
    guess, LM with the launch parameters.  Flags and iteration counts exact against the oracle, poses inside north_star's
    tolerance; run with two batches in flight (the same pairs in two slot ranges)."""
    P, n = 32, 8192
    clouds, guesses = [], []
    for p in range(P):
        s, t, _, _ = scene.make_pair(n, n, scene.pair_seed(4, p), "loop")
        clouds += [s, t]
        guesses.append(np.eye(4, dtype=np.float32))
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    b.set_clouds(0, clouds)
    t0 = b.align_enqueue([(2 * i, 2 * i + 1) for i in range(P)], guesses)
    b.set_clouds(2 * P, clouds)
    t1 = b.align_enqueue([(2 * P + 2 * i, 2 * P + 2 * i + 1) for i in range(P)], guesses)
    r0, r1 = b.align_collect(t0), b.align_collect(t1)
    assert r0.tobytes() == r1.tobytes()
    assert int(r0["n_linearize"].max()) >= 20 and int(r0["n_linearize"].min()) <= 5
    for p in range(P):
        o = R.RefAPDGICP(R.default_params(**LM))
        o.setInputSource(clouds[2 * p]), o.setInputTarget(clouds[2 * p + 1])
        To = o.align(guesses[p])
        te, re_ = scene.pose_error(To, reg.result_matrix(r0[p]))
        assert te <= T_TOL and re_ <= R_TOL, (p, te, re_)
        assert [r0[p]["converged"], r0[p]["iterations"], r0[p]["n_linearize"], r0[p]["n_compute_error"]] == \
            [int(o.converged), o.nr_iterations, o.n_linearize, o.n_compute_error], p


def test_pool_list_longer_than_one_poll_round(reg, scene):
    """k_pool_poll compacts the list of running pairs 256 entries per round: two batches of 300 and 270 small pairs in flight
    (a list of up to 570 entries, several rounds per poll, finished pairs leaving from every part of it) must give the records of
    the host-polled loop, and a third batch admitted while the first two still run must too."""
    kw = dict(max_correspondence_distance=2.5, transformation_epsilon=1e-3, azimuth_variance_deg=1.0)
    base_clouds, base_guesses = [], []
    for p in range(12):
        s, t, _, g = scene.make_pair(600 + 37 * p, 700, scene.pair_seed(370, p), "loop" if p % 3 else "odometry")
        base_clouds += [s, t]
        base_guesses.append(g if p % 3 == 0 else np.eye(4, dtype=np.float32))
    sizes = (300, 270, 40)
    b = reg.BatchAPDGICP(reg.default_params(**kw))
    b.set_clouds(0, base_clouds)      # 24 clouds shared by every pair of every batch: nothing is replaced while batches fly
    batches = []
    for k, n in enumerate(sizes):
        idx = [(7 * i + 3 * k) % 12 for i in range(n)]
        batches.append(([(2 * j, 2 * j + 1) for j in idx], [base_guesses[j] for j in idx]))
    tickets = [b.align_enqueue(pairs, guesses) for pairs, guesses in batches]
    got = [b.align_collect(t) for t in (tickets[1], tickets[0], tickets[2])]
    ref_b = reg.BatchAPDGICP(reg.default_params(**kw))
    ref_b.set_clouds(0, base_clouds)
    for g, k in zip(got, (1, 0, 2)):
        want = polled_align(ref_b, *batches[k])
        assert len(g) == sizes[k] and g.tobytes() == want.tobytes(), k
    assert len(set(int(x) for x in got[1]["n_linearize"])) > 3


def test_is_pooled_reports_the_number_of_batches_a_handle_keeps_in_flight(reg, monkeypatch):
    """include/apdgicp_hip.h: > 0 = the pool's lanes (24 unless APDGICP_POOL_LANES says otherwise, at most 32), 0 = no pool
    (Gauss-Newton: two record buffers per handle)."""
    monkeypatch.delenv("APDGICP_POOL_LANES", raising=False)
    lm = reg.BatchAPDGICP(reg.default_params(**LM))
    assert lm.L.apdgicp_batch_is_pooled(lm.b) == 24
    gn = reg.BatchAPDGICP(reg.default_params(optimizer=1, **LM))
    assert gn.L.apdgicp_batch_is_pooled(gn.b) == 0
    for env, want in (("8", 8), ("64", 32), ("0", 1)):
        monkeypatch.setenv("APDGICP_POOL_LANES", env)
        assert lm.L.apdgicp_batch_is_pooled(lm.b) == want


@pytest.mark.parametrize("lanes,seed", ((24, 1), (5, 2), (2, 3)))
def test_random_schedule_on_the_two_lists(reg, scene, monkeypatch, lanes, seed):
    """A randomised stream of batches -- 1 ... 20 pairs, clouds of 400 ... 2600 points (the pool lays itself out anew when a batch
    outgrows it), a random number of batches in flight, collected in random order, odometry and loop pairs mixed (2 ... 40
    iterations) -- on the two independent pair lists of one handle: every record byte-identical to the host-polled loop."""
    if lanes != 24:
        monkeypatch.setenv("APDGICP_POOL_LANES", str(lanes))
    else:
        monkeypatch.delenv("APDGICP_POOL_LANES", raising=False)
    rng = np.random.default_rng(1000 + seed)
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    slot_base, in_flight, done = 0, [], 0
    for step in range(36):
        n_pairs = int(rng.integers(1, 21))
        n_pts = int(rng.choice([400, 900, 1500, 2600]))
        clouds, guesses = [], []
        for p in range(n_pairs):
            kind = "loop" if rng.random() < 0.6 else "odometry"
            a, c_, _, g = scene.make_pair(n_pts + int(rng.integers(0, 50)), n_pts, scene.pair_seed(500 + seed, 100 * step + p), kind)
            clouds += [a, c_]
            guesses.append(np.eye(4, dtype=np.float32) if kind == "loop" else g)
        ref_b.set_clouds(0, clouds)
        want = polled_align(ref_b, [(2 * i, 2 * i + 1) for i in range(n_pairs)], guesses)
        # every batch in flight owns its own range of cloud slots
        base = slot_base
        slot_base = (slot_base + 2 * n_pairs) % 4000
        if base + 2 * n_pairs > 4000:
            base, slot_base = 0, 2 * n_pairs
        if any(not (base + 2 * n_pairs <= ob or ob + 2 * on <= base) for _, _, ob, on in in_flight):   # would overlap a batch in flight: drain first
            for t, w, _, _ in in_flight:
                assert b.align_collect(t).tobytes() == w.tobytes()
                done += 1
            in_flight = []
        b.set_clouds(base, clouds)
        in_flight.append((b.align_enqueue([(base + 2 * i, base + 2 * i + 1) for i in range(n_pairs)], guesses), want, base, n_pairs))
        while len(in_flight) > min(lanes - 1, int(rng.integers(1, 12))):   # (one lane stays free for the next enqueue: a ticket whose lane is reused is void)
            t, w, _, _ = in_flight.pop(int(rng.integers(0, len(in_flight))))
            assert b.align_collect(t).tobytes() == w.tobytes(), step
            done += 1
    for t, w, _, _ in in_flight:
        assert b.align_collect(t).tobytes() == w.tobytes()
        done += 1
    assert done == 36


def test_pooled_search_launches_are_timed(reg, scene, monkeypatch):
    """bench.py's roofline for the LM line: with profiling on, the pool brackets one search launch of every APDGICP_PROFILE_STRIDE-th
    chunk with events; last_nn_profile() hands out what has been harvested since the previous call (milliseconds, launches, pairs
    really on the list) and resets it; profiling changes no record."""
    monkeypatch.setenv("APDGICP_PROFILE_STRIDE", "2")
    data = loop_batches(scene, 6, 8, 2048, 390)
    pair_idx = lambda base: [(base + 2 * i, base + 2 * i + 1) for i in range(8)]  # noqa: E731
    ref_b = reg.BatchAPDGICP(reg.default_params(**LM))
    b = reg.BatchAPDGICP(reg.default_params(**LM))
    b.set_profiling(True)
    tickets, want = [], []
    for s, (clouds, guesses) in enumerate(data):
        ref_b.set_clouds(0, clouds)
        want.append(polled_align(ref_b, pair_idx(0), guesses))
        b.set_clouds(16 * s, clouds)
        tickets.append(b.align_enqueue(pair_idx(16 * s), guesses))
    ms = launches = pairs = 0
    for t, w in zip(tickets, want):
        assert b.align_collect(t).tobytes() == w.tobytes()
        m_, l_, p_ = b.last_nn_profile()
        ms, launches, pairs = ms + m_, launches + l_, pairs + p_
    assert launches >= 2 and 0.0 < ms / launches < 5.0
    assert launches <= pairs <= launches * 6 * 8          # at least one pair per timed launch, never more than are in flight
    assert b.last_nn_profile() == (0.0, 0, 0)             # reading resets
    assert "k_nn" in b.last_nn_kernel()
