cd $GRAFT_REPO_ROOT
{
echo "# round 3, third session (latch geometry, packed group tail, component-lane sum, path-cost prefetch): randomised parity campaigns with fresh seeds"
for args in "fuzz_parity.py 12000 31337001" "fuzz_parity.py 6000 424242" "fuzz_api.py 3000 909090"; do
  echo "== tools/$args"; python tools/$args 2>&1 | tail -1
done
} > gpurun_out/r3_fuzz_session3.txt 2>&1
cat gpurun_out/r3_fuzz_session3.txt
