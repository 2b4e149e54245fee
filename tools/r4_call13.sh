# round 4, GPU call 13: the f32 stacked SchNet pass (two runs + kernel trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
for i in 1 2; do (timeout 600 python bench.py --workload schnet4096 --steps 8 --warmup 2 --no-cpu-baseline > $O/c13_bench_schnet_f32_$i.json 2> $O/c13_bench_schnet_f32.err); python -c "
import json;d=json.load(open('$O/c13_bench_schnet_f32_$i.json'));print('schnet f32',d['value'],d['ms_per_step'],d['roofline']['step_roof']['frac'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload schnet4096 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/q0 -name "*results.db" | head -1) 2>/dev/null | head -50 > $R/$O/c13_stats_f32.txt; head -40 $R/$O/c13_stats_f32.txt | cut -c1-150
