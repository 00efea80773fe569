#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE PMC passes
(tools/gpu_prof.sh): HBM bytes per launch of the dominant kernel, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B
request: doubled; counters are in KiB). One entry per (config, kernel):
usage: traffic_from_pmc.py C2=r1_c2 C3=r1_c3 C5x8=r1_c5"""
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
    for k, v in out.items():
        fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        res["%s:%s" % (cfg, k)] = {
            "fetch_kib_raw": fetch, "write_kib_raw": write,
            "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --steps 50 --warmup 5` "
                      "(profiles/%s_pmc3.txt, _pmc4.txt); FETCH_SIZE doubled per the gfx950 note" % tag}
json.dump(res, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
