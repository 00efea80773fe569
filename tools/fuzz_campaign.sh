#!/bin/bash
# randomised parity campaigns with fresh seeds against the oracle (bit-exact; tools/fuzz_parity.py: random scenes through
# every kernel family, tools/fuzz_api.py: random C-ABI call sequences) -- prints the three summary lines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
SEED=${1:-8880001}
echo "# round 5 (final kernels: glibc-compatible exp; fuzz_api also draws closed loop, prediction_freq_multiple 2 / 3 and a goal change with the best agent carried over): randomised parity campaigns, seeds from $SEED"
for args in "fuzz_parity.py 8000 $SEED" "fuzz_api.py 6000 $((SEED+2))"; do
  echo "== tools/$args"; python tools/$args 2>&1 | tail -1
done
