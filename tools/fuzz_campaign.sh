cd $GRAFT_REPO_ROOT
{
echo "# round 3, third session, final kernels (… + closest-other table): randomised parity campaigns with fresh seeds"
for args in "fuzz_parity.py 12000 5550001" "fuzz_parity.py 6000 5550002" "fuzz_api.py 3000 5550003"; do
  echo "== tools/$args"; python tools/$args 2>&1 | tail -1
done
} > gpurun_out/r3_fuzz_session3b.txt 2>&1
cat gpurun_out/r3_fuzz_session3b.txt
