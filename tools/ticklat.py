"""Set-point latency of idle-stream ticks on the library's own clock (pmaf_get_tick_times_us), with / without the
rollout's event timing, with / without obstacles handed over, with / without the winner path; then the CLOSED-LOOP tick:
pmaf_set_real_position (the measured position, B/src/panda_bimanual_control.cpp:333-335) + pmaf_tick timed together
around the two calls, open loop beside it on the same clock (round 5: the position rides in pinned memory into the
manager kernel -- no stream sync, no copy command).
usage: python tools/ticklat.py [C2] [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pm = g.load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
sc = pm.scenes.config_scene(cfg)
for prof in (False, True):
    for with_obs in (False, True):
        for wp in (False, True):
            h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
            h.set_initial_position(sc["start"])
            h.set_profiling(prof)
            if wp: h.enable_winner_path()
            for k in range(n + 50):
                if k == 50: h.tick_times_us()
                if k % 128 == 0: h.set_initial_position(sc["start"])
                h.stop()
                h.tick(sc["obstacles"] if with_obs else None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
                if wp: h.winner_path_wait()
            h.stop()
            enq, sp = h.tick_times_us()
            w = h.winner_path_times_us() if wp else np.zeros(1)
            print("%s events %-5s obstacles %-5s winner path %-5s | enqueue median %.2f p99 %.2f | set-point median %.2f p99 %.2f | path median %.2f p99 %.2f us"
                  % (cfg, prof, with_obs, wp, np.median(enq), np.percentile(enq, 99), np.median(sp), np.percentile(sp, 99),
                     np.median(w), np.percentile(w, 99)), flush=True)
            h.close()

# ---- closed loop: set_real_position + tick, wall clock around the calls (the interpreter's share is in both rows)
import time
for closed in (False, True):
    h = pm.PmafPlanner(sc, device=0, mgr_init_pos=sc["start"])
    h.set_initial_position(sc["start"])
    meas = np.asarray(sc["start"], dtype=np.float64).copy()
    wall = np.zeros(n)
    for k in range(n + 50):
        if k == 50: h.tick_times_us()
        if k % 128 == 0:
            h.set_initial_position(sc["start"]); meas = np.asarray(sc["start"], dtype=np.float64).copy()
        h.stop()
        t0 = time.perf_counter()
        if closed: h.set_real_position(meas)
        h.tick(None, sc["dt"], sc["cost_gains"], sc["ws_limits"])
        if k >= 50: wall[k - 50] = (time.perf_counter() - t0) * 1e6
        sp_ = np.asarray(h.real_state()[0]); meas = sp_ - 0.3 * (sp_ - meas)
    h.stop()
    enq, sp = h.tick_times_us()
    print("%s %-11s | around the calls median %.2f p99 %.2f us | pmaf_tick alone (library clock) set-point median %.2f p99 %.2f us"
          % (cfg, "closed loop" if closed else "open loop", np.median(wall), np.percentile(wall, 99), np.median(sp), np.percentile(sp, 99)), flush=True)
    h.close()

# ---- a NEW obstacle list on every tick (moving obstacles streamed at the tick rate, as the shipped dyn tasks do): the
# manager kernel reads the list out of mapped pinned host memory (PCIe) in front of the real step
for cfgname, dyn_sc in (("C2 moving", pm.scenes.config_scene("C2", scene_id=3, dynamic=True)), ("C3 moving", pm.scenes.config_scene("C3", dynamic=True))):
    h = pm.PmafPlanner(dyn_sc, device=0, mgr_init_pos=dyn_sc["start"])
    h.set_initial_position(dyn_sc["start"])
    o = dyn_sc["obstacles"].copy()
    for k in range(n + 50):
        if k == 50: h.tick_times_us()
        if k % 128 == 0:
            h.set_initial_position(dyn_sc["start"]); o = dyn_sc["obstacles"].copy()
        h.stop()
        h.tick(o, dyn_sc["dt"], dyn_sc["cost_gains"], dyn_sc["ws_limits"])
        o = pm.scenes.advance_live_obstacles(o)
    h.stop()
    enq, sp = h.tick_times_us()
    print("%s, a new list of %d obstacles on every tick | enqueue median %.2f p99 %.2f | set-point median %.2f p99 %.2f us"
          % (cfgname, o.shape[0], np.median(enq), np.percentile(enq, 99), np.median(sp), np.percentile(sp, 99)), flush=True)
    h.close()
