cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
for s in 0 64 127 254 381 508; do echo "stagger $s"; MDG_CHAIN_STAGGER=$s python tools/kbench_chain.py --rows 32768 2>/dev/null | grep "dual\|single" | grep " 3 stage\| 6 stage" | cut -c1-60; done > $O/c24_stagger.txt; cat $O/c24_stagger.txt
