cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python bench.py --workload schnet4096 --bf16 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-150; done
rocm-smi --showclocks 2>/dev/null | head -12
