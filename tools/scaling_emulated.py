"""profiles/rN_scaling_emulated.json: BASELINE C5's 1/2/4/8-GPU strong-scaling curve as ONE GPU predicts it.
Inputs (tools/evidence.sh scaling): the bench line of the driver's command (its `scaling_c5.prediction` = bench.py's
emulate_c5_scaling: each rank's share of the 8 scenes in a handle of its own, tick at N GPUs = the slowest rank's), the
C5 line with a ONE-rank RCCL communicator (what enqueueing + running ncclAllGather of the winner records costs on the
exchange stream), the C5 line of two ranks sharing this GPU over the host transport.
A fourth line, BASELINE C2 with the one-rank RCCL communicator, gives the collective's cost on a GPU with free CUs.
usage: python tools/scaling_emulated.py <bench line> <c5 rccl 1 rank line> <c5 2 ranks host line> [<c2 rccl 1 rank line>]   -> JSON on stdout"""
import json
import sys


def load(p):
    try:
        return json.loads(open(p).read().strip().split("\n")[-1])
    except (OSError, ValueError, IndexError):
        return None


d, r1, h2 = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
c2r = load(sys.argv[4]) if len(sys.argv) > 4 else None
pred = d["scaling_c5"]["prediction"]
ag = {"on_the_tick_critical_path": False,
      "why": "pmaf_tick packs the winner records and enqueues the all-gather on high-priority side streams beside the NEXT rollout "
             "(two slots: a tick never waits for the previous tick's collective); it would matter only if it took longer than a tick",
      "record_bytes_per_population": (8 + 3 * 201) * 8}
if c2r and c2r.get("allgather_us"):
    a = c2r["allgather_us"]
    ag["rccl_one_rank_free_cus"] = {"median_us": a["median"], "p99_us": a["p99"], "n": a["n"], "collective_world": a["collective_world"],
                                    "note": "BASELINE C2 (64 waves: the chip is almost empty), one population's record: what enqueueing + running "
                                            "ncclAllGather costs on the exchange stream when its kernel gets a CU at once; one rank: no wire time"}
if r1 and r1.get("allgather_us"):
    a = r1["allgather_us"]
    ag["rccl_one_rank_under_c5x8"] = {"median_us": a["median"], "p99_us": a["p99"], "n": a["n"], "collective_world": a["collective_world"],
                                      "ms_per_tick_with_it": r1["ms_per_step"], "ms_per_tick_without": d["scaling_c5"]["measured"]["ms_per_tick"],
                                      "note": "8 populations' records beside the 8-scene rollout: the group kernel's 2 x 252 VGPRs fill every SIMD's "
                                              "register file, so the collective's kernel waits for the rollout's first waves to retire -- the figure is that "
                                              "WAIT (about one rollout), not a transfer; the tick does not wait for it (compare the two ms_per_tick)"}
if h2 and h2.get("allgather_us"):
    a = h2["allgather_us"]
    ag["two_ranks_one_gpu_host_transport"] = {"median_us": a["median"], "p99_us": a["p99"], "ms_per_tick": h2["ms_per_step"],
                                              "note": "two processes sharing this GPU, gloo all-gather on the host: the plumbing, not xGMI"}
ag["not_measured_here"] = ("ncclAllGather across GPUs over xGMI (no multi-GPU box): 8 x 4.9 KB records; a small-message ring all-gather "
                           "on 8 GPUs is tens of microseconds, well inside the >= 235 us rollout it overlaps")
# the pessimistic reading: were the all-gather serialised behind every tick, at the one-rank RCCL figure per hop
hop = (ag.get("rccl_one_rank_free_cus") or {}).get("median_us")
out = {"source": "one MI355X; %s" % pred["method"], "predicted_by_n": pred["predicted_by_n"], "chain_floor_us": pred["chain_floor_us"],
       "reason": pred["reason"], "allgather": ag,
       "measured_one_gpu_record_of_the_same_run": d["scaling_c5"]["measured"]}
if hop:
    out["if_the_allgather_were_serialised"] = {
        n: {"ms_per_tick": row["ms_per_tick"] + (int(n) - 1) * hop * 1e-3 if int(n) > 1 else row["ms_per_tick"]}
        for n, row in pred["predicted_by_n"].items()}
    t1 = out["if_the_allgather_were_serialised"]["1"]["ms_per_tick"]
    for n, row in out["if_the_allgather_were_serialised"].items():
        row["efficiency_vs_1gpu"] = t1 / row["ms_per_tick"] / int(n)
    out["if_the_allgather_were_serialised"]["assumption"] = "(N - 1) ring steps, each at the one-rank ncclAllGather time on a GPU with free CUs (launch + kernel, no wire time), ON the critical path -- what the side streams avoid"
print(json.dumps(out, indent=1))
