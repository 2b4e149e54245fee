cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python tools/hostprof_schnet.py --f32 > $O/c14_hostprof_f32.txt 2>&1; head -50 $O/c14_hostprof_f32.txt | cut -c1-150
