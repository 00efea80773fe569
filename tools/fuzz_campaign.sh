#!/bin/bash
# randomised parity campaigns with fresh seeds against the oracle (bit-exact; tools/fuzz_parity.py: random scenes through
# every kernel family, tools/fuzz_api.py: random C-ABI call sequences) -- prints the three summary lines
cd ${GRAFT_REPO_ROOT:-$(pwd)}
SEED=${1:-8880001}
echo "# randomised parity campaigns against the oracle, tolerance 0 (fuzz_api also draws closed loop, prediction_freq_multiple 2 / 3 and a goal change with the best agent carried over; the MANY campaign draws 900 ... 4 600 agents per population: the mappings of the measured table, the priority-slicing loop), seeds from $SEED"
for args in "fuzz_parity.py 8000 $SEED" "fuzz_api.py 6000 $((SEED+2))"; do
  echo "== tools/$args"; python tools/$args 2>&1 | tail -1
done
echo "== PMAF_FUZZ_MANY=1 tools/fuzz_parity.py 1500 $((SEED+5))"; PMAF_FUZZ_MANY=1 python tools/fuzz_parity.py 1500 $((SEED+5)) 2>&1 | tail -1
