cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -i -E "marketing|gfx|compute unit" | head -8 > gpurun_out/r3_box.txt
lscpu | head -20 >> gpurun_out/r3_box.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>&1 | grep -i -B2 -A10 "pc.sampl" | head -60 > $GRAFT_REPO_ROOT/gpurun_out/r3_listavail.txt; cd $GRAFT_REPO_ROOT
python tools/quicktime.py C1:64 C2:64 C3:64 2>&1 | grep -v "^$" > gpurun_out/r3_base_quick.txt
python tools/c5time.py >> gpurun_out/r3_base_quick.txt 2>&1
bash tools/pcsamp.sh c2_stoch stochastic cycles 1048576 C2:64
bash tools/pcsamp.sh c2_host host_trap time 1 C2:64
cat gpurun_out/r3_base_quick.txt
