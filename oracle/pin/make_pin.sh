#!/bin/bash
# make_pin.sh -- ONE command that pins the CPU oracle of this repo to the REFERENCE's own arithmetic.
#
#   bash oracle/pin/make_pin.sh [/path/to/predictive-multi-agent-framework]      (default: /root/reference)
#
# What it does: compiles the reference's planner core -- src/bimanual_planning_ros/src/cf_agent.cpp and cf_manager.cpp,
# UNMODIFIED, from where they lie -- together with the build-owned driver oracle/pin/pin_harness.cpp against the
# machine's REAL Eigen3 and REAL dqrobotics headers, runs every scenario of oracle/pin/scenarios/, and writes
# tests/golden/ref_<scenario>.json (every double a hex literal). `python -m pytest tests/test_reference_pin.py` then
# holds oracle/pmaf_oracle.c to those files bit for bit and reports which 3-vector dot-product association
# (include/pmaf.h, pmaf_eval_order) the reference build evaluated -- i.e. which libpmaf_hip.so variant replaces it.
#
# What it will NOT do: build against stand-ins. Eigen3 is the arithmetic being pinned; a look-alike header pins nothing
# (tests/golden/survey_probe.json is such a record and is treated as a smoke value only). If Eigen3 or dqrobotics
# (the reference's headers include <dqrobotics/DQ.h> and the V-REP interface header) is missing, the script says what is
# missing and exits 77 (= skipped) without writing anything. The build container of this repo has neither, so the
# fixtures can only be produced on a machine that can build the reference (Ubuntu 18.04 / 20.04 + ROS + the dqrobotics
# PPA, reference README.md:29-33): run it there and commit tests/golden/ref_*.json.
#
# Environment: CXX (default g++), PIN_CXXFLAGS (default "-O2": the numbers do not depend on the optimisation level
# without -ffast-math, but DO depend on -DEIGEN_DONT_VECTORIZE and on -march flags that enable FMA contraction -- build
# the pin the way the node is built; the harness records Eigen's version, EIGEN_VECTORIZE, the compiler and glibc in
# every file), PIN_SCENARIOS (default: all).
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
REF=${1:-/root/reference}
B=$REF/src/bimanual_planning_ros
CXX=${CXX:-g++}
skip() { echo "make_pin.sh: SKIPPED -- $1" >&2; exit 77; }

[ -f "$B/src/cf_agent.cpp" ] && [ -f "$B/src/cf_manager.cpp" ] && [ -f "$B/include/bimanual_planning_ros/cf_manager.h" ] \
  || skip "no reference checkout at $REF (need src/bimanual_planning_ros/src/cf_agent.cpp, cf_manager.cpp)"

# ---- a REAL Eigen3: the directory that holds eigen3/Eigen/Dense (the reference includes "eigen3/Eigen/Dense") ----
EIGEN_PARENT=""
if command -v pkg-config > /dev/null 2>&1 && pkg-config --exists eigen3; then
  d=$(pkg-config --variable=includedir eigen3 2> /dev/null); [ -z "$d" ] && d=$(pkg-config --cflags-only-I eigen3 | sed -e 's/^-I//' -e 's/ .*//')
  case "$d" in */eigen3) EIGEN_PARENT=$(dirname "$d") ;; esac
fi
for d in /usr/include /usr/local/include /opt/homebrew/include; do
  [ -z "$EIGEN_PARENT" ] && [ -f "$d/eigen3/Eigen/Dense" ] && EIGEN_PARENT=$d
done
[ -n "$EIGEN_PARENT" ] || skip "Eigen3 not found (pkg-config eigen3, /usr/include/eigen3): install libeigen3-dev"
# the real library ships this marker file and the Core sources; a header-shaped stand-in does not
[ -f "$EIGEN_PARENT/eigen3/signature_of_eigen3_matrix_library" ] && [ -f "$EIGEN_PARENT/eigen3/Eigen/src/Core/Redux.h" ] \
  || skip "$EIGEN_PARENT/eigen3 is not a complete Eigen3 installation (signature_of_eigen3_matrix_library / Eigen/src/Core/Redux.h missing)"

# ---- REAL dqrobotics headers (dqrobotics + its V-REP interface package) ----
DQ_PARENT=""
for d in /usr/include /usr/local/include; do
  [ -z "$DQ_PARENT" ] && [ -f "$d/dqrobotics/DQ.h" ] && DQ_PARENT=$d
done
[ -n "$DQ_PARENT" ] || skip "dqrobotics not found (<dqrobotics/DQ.h>): install libdqrobotics (PPA dqrobotics-dev/release)"
for hdr in dqrobotics/interfaces/vrep/DQ_VrepInterface.h dqrobotics/robot_modeling/DQ_CooperativeDualTaskSpace.h dqrobotics/robot_modeling/DQ_SerialManipulator.h; do
  [ -f "$DQ_PARENT/$hdr" ] || skip "<$hdr> not found under $DQ_PARENT: install libdqrobotics-interface-vrep"
done
grep -q "class DQ" "$DQ_PARENT/dqrobotics/DQ.h" && [ "$(wc -l < "$DQ_PARENT/dqrobotics/DQ.h")" -gt 100 ] \
  || skip "$DQ_PARENT/dqrobotics/DQ.h does not look like the dqrobotics header"

mkdir -p "$ROOT/oracle/_ref"
EXE=$ROOT/oracle/_ref/pin_harness
FLAGS="-std=c++17 ${PIN_CXXFLAGS:--O2}"
echo "make_pin.sh: Eigen3 at $EIGEN_PARENT/eigen3, dqrobotics at $DQ_PARENT/dqrobotics, $CXX $FLAGS"
# the reference's two sources by path, with its own include directory; the harness uses CfManager's public surface only
set -e
$CXX $FLAGS -I"$B/include" -I"$EIGEN_PARENT" -I"$DQ_PARENT" -c "$B/src/cf_agent.cpp" -o "$ROOT/oracle/_ref/ref_cf_agent.o"
$CXX $FLAGS -I"$B/include" -I"$EIGEN_PARENT" -I"$DQ_PARENT" -c "$B/src/cf_manager.cpp" -o "$ROOT/oracle/_ref/ref_cf_manager.o"
$CXX $FLAGS -I"$B/include" -I"$EIGEN_PARENT" -I"$DQ_PARENT" -c "$HERE/pin_harness.cpp" -o "$ROOT/oracle/_ref/pin_harness.o"
$CXX -o "$EXE" "$ROOT/oracle/_ref/pin_harness.o" "$ROOT/oracle/_ref/ref_cf_agent.o" "$ROOT/oracle/_ref/ref_cf_manager.o" -lpthread
set +e
n=0
for sc in "$HERE"/scenarios/${PIN_SCENARIOS:-*}.txt; do
  name=$(basename "$sc" .txt)
  out=$ROOT/tests/golden/ref_$name.json
  echo "make_pin.sh: $name"
  if ! "$EXE" "$sc" "$out.tmp"; then echo "make_pin.sh: $name FAILED" >&2; rm -f "$out.tmp"; exit 1; fi
  mv "$out.tmp" "$out"
  n=$((n + 1))
done
echo "make_pin.sh: wrote $n file(s) tests/golden/ref_*.json -- now run: python -m pytest tests/test_reference_pin.py -q -s"
