"""The lanes-per-agent rule (csrc/pmaf_lpa_model.hpp, exported as pmaf_pick_lanes_per_agent / pmaf_estimate_rollout_us:
pure functions, no device) held to the measurements it was built from AND to rows it was not: profiles/r6_lpa_grid.txt
(the fitted grid), r6_lpa_band.txt (the 1 025 ... 2 048-agent band, incl. several populations per handle),
r6_lpa_heldout.txt (used to correct the first version of the table) and r6_lpa_heldout2.txt (never used for fitting).
Every file is tools/lpaband.py's output on one MI355X: kernel us per 200-step launch for each mapping.
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


def test_round5_rule_would_lose_a_fifth_in_the_three_regions(lib):
    """what the table is for: rows where "narrow until <= 2048 waves, at most two slots per lane" picked a mapping 19 ... 31 %
    slower than the best one (the r5 choice is the `(auto)` column of the grid, which was measured with the r5 library)"""
    g = {(N, M): d for N, P, M, H, d in rows("r6_lpa_grid.txt")}
    for (N, M), r5_pick, r6_pick in (((2304, 9), 32, 16), ((4096, 16), 32, 16), ((2304, 48), 32, 64), ((3072, 60), 32, 64),
                                     ((2048, 64), 64, 32)):
        d = g[(N, M)]
        assert lib.pmaf_pick_lanes_per_agent(N, 1, M, 0) == r6_pick
        assert d[r5_pick] / d[r6_pick] > 1.10, ((N, M), d)


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
