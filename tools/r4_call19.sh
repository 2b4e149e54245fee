cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python tools/hostprof_schnet.py --opt 8 > $O/c19_bf16_opt.txt 2>&1; grep "^pass\|collections" $O/c19_bf16_opt.txt | cut -c1-220
MDG_NO_GC=1 python tools/hostprof_schnet.py --opt 8 > $O/c19_bf16_opt_nogc.txt 2>&1; grep "^pass\|collections" $O/c19_bf16_opt_nogc.txt | cut -c1-220
