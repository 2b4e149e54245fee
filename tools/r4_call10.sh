# round 4, GPU call 10: host side of the eager stacked pass; HIP-graph replay above the edge cap; chain kernels after the partial revert
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
python tools/hostprof_schnet.py > $O/c10_hostprof.txt 2>&1; head -60 $O/c10_hostprof.txt | cut -c1-160
for v in bf16 bf16-rows; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c10_bench_schnet_$v.json 2> $O/c10_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c10_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'])"; done
for v in bf16 bf16-rows; do (MDG_GRAPH_MAX_EDGES=1048576 timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c10_bench_graph_$v.json 2> $O/c10_bench_graph_$v.err); tail -3 $O/c10_bench_graph_$v.err; python -c "
import json;d=json.load(open('$O/c10_bench_graph_$v.json'));print('schnet graph $v',d['value'],d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
for v in bf16; do rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload schnet4096 --$v --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/q0 -name "*results.db" | head -1) 2>/dev/null | head -45 > $R/$O/c10_stats_$v.txt; grep chain $R/$O/c10_stats_$v.txt | cut -c1-150; done
