# round 4, GPU call 9: rows16 tests, chain kernels with early loads + transposed weight copies, A/B rates, kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_schnet_rows16.py -m gpu -q -s 2>&1) > $O/c9_rows16.log; grep "^DEV" $O/c9_rows16.log; tail -8 $O/c9_rows16.log
python tools/kbench_cfconv.py --rows16 > $O/c8_kbench_rows16.txt 2>&1; paste -d'|' $O/c7_kbench_bf16.txt $O/c8_kbench_rows16.txt 2>/dev/null | cut -c1-100; cat $O/c8_kbench_rows16.txt
for v in bf16 bf16-rows bf16 bf16-rows; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c8_bench_schnet_$v.json 2> $O/c8_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c8_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['forward_kernel']['kernel_ms'],d['config'].get('bf16_vs_f32'))"; done
cd /tmp && export TMPDIR=/tmp
for v in bf16 bf16-rows; do rm -rf /tmp/q0; rocprofv3 --kernel-trace --stats -d /tmp/q0 -o run -- python $R/bench.py --workload schnet4096 --$v --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py stats $(find /tmp/q0 -name "*results.db" | head -1) 2>/dev/null | head -45 > $R/$O/c8_stats_$v.txt; head -30 $R/$O/c8_stats_$v.txt | cut -c1-150; done
