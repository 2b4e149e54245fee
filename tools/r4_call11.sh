cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python tools/hostprof_schnet.py --bf16-rows > $O/c11_hostprof.txt 2>&1; head -8 $O/c11_hostprof.txt
