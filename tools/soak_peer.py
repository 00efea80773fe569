"""Soak run of the peer mailboxes (BASELINE config 4, one arm per PROCESS, two ranks sharing GPU 0: the set-points travel
through hipIpc-mapped inboxes only -- tests/test_peer_gpu.py's two-process layout, its worker reused) for n_ticks ticks
instead of 200, every tick's set-point of both arms against two coupled CPU oracles (TEST INFRASTRUCTURE) bit for bit.
The in-kernel polling of a header another process's kernel stores is the one hand-off of this build that no collective
or stream orders: a stale or torn header once in 10^5 ticks would show here.
usage: python tools/soak_peer.py [n_ticks = 30000]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.multiprocessing as mp
    import __graft_entry__ as g
    import test_peer_gpu as T
    pm = g.load_package()
    from oracle import orc
    orc.set_exp_mode(1)
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=T._ipc_worker, args=(r, world, PORT, ticks, False, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=1500)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(120)
    t_gpu = time.time() - t0
    arms = pm.scenes.dual_arm_scenes(64, 150, 24)
    ref, gap, oras = T._coupled_oracles(orc, pm.shard, arms, ticks)
    bad = sum(int(not np.array_equal(got[r][0][t], ref[t, r])) for r in range(world) for t in range(ticks))
    moving = int((np.abs(np.diff(ref[:, 0], axis=0)).max(axis=1) > 0).sum())
    print("peer-mailbox soak: %d ticks x 2 ranks (the arms move during %d of them), %d set-points differ from the coupled oracles; "
          "header wait median %.2f / %.2f us, p99 %.1f / %.1f us; exit codes %s; %.0f s on the GPU side"
          % (ticks, moving, bad, got[0][2], got[1][2], got[0][3], got[1][3], [p.exitcode for p in procs], t_gpu))
    sys.exit(1 if bad or any(p.exitcode for p in procs) else 0)


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); PORT = s.getsockname()[1]; s.close()
    main()
