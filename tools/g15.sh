cd $GRAFT_REPO_ROOT
python tools/kbench_cfconv.py --bf16 2>/dev/null | head -7
timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-160
