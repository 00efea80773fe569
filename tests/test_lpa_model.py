"""The lanes-per-agent rule (csrc/pmaf_lpa_model.hpp, exported as pmaf_pick_lanes_per_agent / pmaf_estimate_rollout_us:
pure functions, no device) held to the measurements it was built from AND to rows it was not: profiles/r6_lpa_grid.txt
(the fitted grid), r6_lpa_band.txt (the 1 025 ... 2 048-agent band, incl. several populations per handle),
r6_lpa_heldout.txt (used to correct the first version of the table) and r6_lpa_heldout2.txt (never used for fitting).
Every file is tools/lpaband.py's output on one MI355X: kernel us per 200-step launch for each mapping (band and grid re-measured with
the round's final library: the wave per agent's two-per-SIMD rows run its priority-slicing loop).
The scheduling decision has no counterpart in the reference (one std::thread per agent, B/src/cf_manager.cpp:118-123)."""
import ctypes
import os
import re

import pytest

import conftest

ROOT = conftest.ROOT
PROF = os.path.join(ROOT, "profiles")
ROW = re.compile(r"M\s+(\d+) N\s+(\d+) P (\d+) H (\d+) \| (.*)")


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert hip_lib.pmaf_estimate_rollout_us.restype is ctypes.c_double     # (bound by planner.py's table)
    return hip_lib


def rows(name):
    out = []
    for line in open(os.path.join(PROF, name)):
        m = ROW.match(line)
        if not m:
            continue
        M, N, P, H = (int(m.group(i)) for i in (1, 2, 3, 4))   # "M <field obstacles> N <agents> P <populations> H <steps>"
        d = {}
        for part in m.group(5).split("|"):
            q = re.match(r"\s*lpa (\d+)( \(auto\))?[^:]*: (\d+) us", part)
            if q and not q.group(2):
                d[int(q.group(1))] = int(q.group(3))
        if d:
            out.append((N, P, M, H, d))
    return out


# (file, rows at least, largest accepted regret of the chosen mapping: 5 % on what the table was fitted / corrected on, 8 % on the
# never-fitted set -- its worst rows: 34 obstacles x 11 264 agents 7.6 %, 34 x 6 656 6.4 %, 6 x 2 200 6.3 %)
FILES = [("r6_lpa_grid.txt", 100, 0.05), ("r6_lpa_band.txt", 30, 0.05), ("r6_lpa_heldout.txt", 90, 0.05), ("r6_lpa_heldout2.txt", 100, 0.08)]


@pytest.mark.parametrize("name,min_rows,max_regret", FILES)
def test_the_chosen_mapping_is_within_a_few_percent_of_the_best_measured_one(lib, name, min_rows, max_regret):
    rs = rows(name)
    assert len(rs) >= min_rows
    worst = (0.0, ())
    for N, P, M, H, d in rs:
        pick = lib.pmaf_pick_lanes_per_agent(N, P, M, 0)
        if pick not in d:
            continue   # (a row that did not time every mapping)
        regret = d[pick] / min(d.values()) - 1.0
        if regret > worst[0]:
            worst = (regret, (N, P, M, pick, d))
        est = lib.pmaf_estimate_rollout_us(pick, N, P, M, H, 0)
        if M <= 128:   # (the four-slot kernel's estimate is an extrapolation: it is the only mapping offered there)
            assert abs(est / d[pick] - 1.0) < 0.15, (N, P, M, pick, est, d)
    assert worst[0] <= max_regret, worst


def round5_rule(N, P, M):
    """rounds 1-5: narrow the mapping until the launch has <= 2048 waves, then widen again while a lane would hold more than
    two obstacle slots (pmaf_host.cpp before round 6)"""
    lpa = 64
    while lpa > 1 and ((N * lpa + 63) // 64) * P > 2048:
        lpa //= 2
    while lpa < 64 and (M + lpa - 1) // lpa > 2:
        lpa *= 2
    return lpa


def test_round5_rule_would_lose_a_fifth_in_the_three_regions(lib):
    """what the table is for: over the measured grid the old wave-count rule is up to ~30 % behind the best mapping in three
    regions it had never been measured in, the table's choice nowhere more than 5 %"""
    worst_old, worst_new, losers = 0.0, 0.0, []
    for N, P, M, H, d in rows("r6_lpa_grid.txt"):
        old, new = round5_rule(N, P, M), lib.pmaf_pick_lanes_per_agent(N, P, M, 0)
        if old not in d or new not in d:
            continue
        best = min(d.values())
        worst_old, worst_new = max(worst_old, d[old] / best - 1), max(worst_new, d[new] / best - 1)
        if d[old] / best - 1 > 0.15:
            losers.append((M, N, old, new))
    assert worst_old > 0.25 and worst_new <= 0.05, (worst_old, worst_new)
    region = lambda f: [x for x in losers if f(*x)]
    assert region(lambda M, N, o, n: M <= 16 and 2304 <= N <= 4096 and (o, n) == (32, 16))           # few obstacles: 16 lanes, not 32
    assert region(lambda M, N, o, n: 33 <= M <= 60 and 2304 <= N <= 3072 and (o, n) == (32, 64))      # the one-slot wave per agent, third round
    g = {(N, M): d for N, P, M, H, d in rows("r6_lpa_grid.txt")}
    d = g[(2048, 64)]                                                                                 # 61..64 obstacles: 32 lanes x 2 slots
    assert round5_rule(2048, 1, 64) == 64 and lib.pmaf_pick_lanes_per_agent(2048, 1, 64, 0) == 32 and d[64] / d[32] > 1.10


def test_baseline_configurations_keep_their_mappings(lib):
    for N, P, M, want in ((16, 1, 9, 64), (64, 1, 32, 64), (256, 1, 128, 64), (256, 2, 32, 64), (10, 1, 9, 64),
                          (1024, 8, 32, 16), (1024, 4, 32, 32), (1024, 2, 32, 64), (1024, 1, 32, 64)):
        assert lib.pmaf_pick_lanes_per_agent(N, P, M, 0) == want, (N, P, M)


def test_estimates_are_sane(lib):
    e = lib.pmaf_estimate_rollout_us
    assert lib.pmaf_pick_lanes_per_agent(0, 1, 9, 0) == 0 and lib.pmaf_pick_lanes_per_agent(64, 0, 9, 0) == 0
    assert e(32, 64, 1, 65, 200, 0) < 0 and e(16, 64, 1, 33, 200, 0) < 0 and e(8, 64, 1, 17, 200, 0) < 0   # > 2 slots per lane: not offered
    assert e(48, 64, 1, 9, 200, 0) < 0                                                                      # not a tuned mapping
    assert e(64, 64, 1, 300, 200, 0) > 0                                                                    # the wave per agent always is
    for lpa, M in ((64, 32), (64, 100), (32, 32), (32, 60), (16, 9), (16, 32), (8, 8), (8, 16)):
        prev = 0.0
        for N in (64, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
            t = e(lpa, N, 1, M, 200, 0)
            assert t >= prev > -1, (lpa, M, N, t, prev)          # more agents never cost less
            prev = t
        assert abs(e(lpa, 4096, 1, M, 400, 0) / e(lpa, 4096, 1, M, 200, 0) - 2.0) < 1e-9      # linear in the horizon
        assert e(lpa, 2048, 2, M, 200, 0) == e(lpa, 4096, 1, M, 200, 0)                        # populations are more waves
    # half the SIMDs: the same launch is twice as many waves per SIMD
    assert e(64, 1024, 1, 32, 200, 512) == e(64, 2048, 1, 32, 200, 1024)
    # many obstacles never narrow the mapping (a narrower one would need > 2 slots per lane)
    for N in (64, 4096, 16384):
        assert lib.pmaf_pick_lanes_per_agent(N, 1, 200, 0) == 64
