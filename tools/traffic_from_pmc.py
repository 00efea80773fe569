#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes
(tools/gpu_prof.sh): HBM bytes per launch of the dominant kernel, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B
request: doubled; counters are in KiB). One entry per (config, kernel):
The x2 for FETCH_SIZE and x1 for WRITE_SIZE were checked on known byte counts in
this kernel's own access patterns (8-byte loads / same-address 8-byte stores):
tools/calib_traffic.hip, profiles/r2_traffic_calibration.txt.
usage: traffic_from_pmc.py C2=r2_c2 C3=r2_c3 C5x8=r2_c5"""
import json
import re
import sys

res = {}
for arg in (sys.argv[1:] or ["C2=r1_c2"]):
    cfg, tag = arg.split("=")
    out = {}
    for name, f in (("FETCH_SIZE", "profiles/%s_pmc3.txt" % tag), ("WRITE_SIZE", "profiles/%s_pmc4.txt" % tag)):
        for line in open(f):
            m = re.match(r"(?:void )?(k_rollout[^|(]*)\(.*\| %s \| (\d+) \| ([0-9.]+) \|" % name, line)
            if m:
                out.setdefault(m.group(1).strip(), {})[name] = float(m.group(3))
    # VALU-active share of a wave's cycles (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both in units of 4 cycles)
    act = {}
    for name, f in (("SQ_WAVE_CYCLES", "profiles/%s_pmc1.txt" % tag), ("SQ_WAVES", "profiles/%s_pmc1.txt" % tag),
                    ("SQ_INSTS_VALU", "profiles/%s_pmc1.txt" % tag), ("SQ_ACTIVE_INST_VALU", "profiles/%s_pmc2.txt" % tag)):
        for line in open(f):
            m = re.match(r"(?:void )?(k_rollout[^|(]*)\(.*\| %s \| (\d+) \| ([0-9.]+) \|" % name, line)
            if m:
                act.setdefault(m.group(1).strip(), {})[name] = float(m.group(3))
    for k, v in out.items():
        fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        res["%s:%s" % (cfg, k)] = {
            "fetch_kib_raw": fetch, "write_kib_raw": write,
            "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
            "valu_active_frac_of_wave_cycles": (act.get(k, {}).get("SQ_ACTIVE_INST_VALU", 0.0) /
                                                max(act.get(k, {}).get("SQ_WAVE_CYCLES", 0.0), 1.0)),
            "valu_insts_per_launch": act.get(k, {}).get("SQ_INSTS_VALU"),
            "waves_per_launch": act.get(k, {}).get("SQ_WAVES"),
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --steps 50 --warmup 5` "
                      "(profiles/%s_pmc3.txt, _pmc4.txt); FETCH_SIZE x 2, WRITE_SIZE x 1 as calibrated on known byte counts in this "
                      "kernel's 8-byte access patterns (profiles/r2_traffic_calibration.txt)" % tag}
json.dump(res, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
