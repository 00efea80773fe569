"""One-box A/B of library builds: the product build and every tools/dbg/ab/<name>/libpmaf_hip.so (variant builds:
`PMAF_OUT=../../tools/dbg/ab/<name> PMAF_EXTRA_KFLAGS=-D... bash predictive-multi-agent-framework_amd/csrc/build.sh`),
interleaved over rounds (boxes differ by +-2 %, runs on one box by +-0.1-0.3 %), timed with tools/policytime.py.
usage: python tools/ab.py [--rounds 3] [--policies strict] C2 C3 C5 ...   -> table of median kernel / tick us per build"""
import glob, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, pol = 3, "strict"
if "--rounds" in args: k = args.index("--rounds"); rounds = int(args[k + 1]); del args[k:k + 2]
if "--policies" in args: k = args.index("--policies"); pol = args[k + 1]; del args[k:k + 2]
cfgs = args or ["C2"]
libs = [("product", os.path.join(ROOT, "predictive-multi-agent-framework_amd", "lib", "libpmaf_hip.so"))]
libs += [(os.path.basename(os.path.dirname(p)), p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "dbg", "ab", "*", "libpmaf_hip.so")))]
res = {}
for r in range(rounds):
    for name, lib in libs:
        tmp = "/tmp/ab_%s_%d.json" % (name, r)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "policytime.py")] + cfgs + ["--rounds", "1", "--policies", pol, "--out", tmp],
                       env=dict(os.environ, PMAF_LIB_PATH=lib), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        for c, d in json.load(open(tmp))["summary"].items():
            for p, v in d.items():
                res.setdefault((c, p), {}).setdefault(name, []).append((v["kernel_us"], v["tick_us"]))
for (c, p), d in sorted(res.items()):
    base = np.median([x[0] for x in d["product"]])
    for name, v in d.items():
        ku = np.median([x[0] for x in v]); tu = np.median([x[1] for x in v])
        print("%-3s %-10s %-28s kernel %8.1f us (%+5.2f %%)  tick %8.1f us   runs %s" %
              (c, p, name, ku, (ku / base - 1) * 100, tu, " ".join("%.1f" % x[0] for x in v)), flush=True)
