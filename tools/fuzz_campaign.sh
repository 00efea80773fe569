#!/bin/bash
# randomised parity campaigns with fresh seeds against the oracle (bit-exact; tools/fuzz_parity.py: random scenes through
# every kernel family, tools/fuzz_api.py: random C-ABI call sequences) -- prints the three summary lines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
SEED=${1:-7770001}
echo "# round 4, final kernels (k_rollout_mw drawn with PMAF_MW in {default, 0, 3, 4}, group kernel's STATIC body, manager selection, resident obstacle lists): randomised parity campaigns, seeds from $SEED"
for args in "fuzz_parity.py 12000 $SEED" "fuzz_parity.py 6000 $((SEED+1))" "fuzz_api.py 3000 $((SEED+2))"; do
  echo "== tools/$args"; python tools/$args 2>&1 | tail -1
done
