# round 4, GPU call 7: bf16 node-row mirrors (rows16 kernels): tests with deviations, kernel timings, stacked SchNet rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_schnet_rows16.py -m gpu -q -s -x 2>&1 | tail -60) > $O/c7_rows16.log; tail -15 $O/c7_rows16.log
python tools/kbench_cfconv.py --bf16 > $O/c7_kbench_bf16.txt 2>&1; python tools/kbench_cfconv.py --rows16 > $O/c7_kbench_rows16.txt 2>&1; paste -d'|' $O/c7_kbench_bf16.txt $O/c7_kbench_rows16.txt | cut -c1-140
python tools/kbench_chain.py > $O/c7_kbench_chain.txt 2>&1; tail -12 $O/c7_kbench_chain.txt
for v in bf16 bf16-rows; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c7_bench_schnet_$v.json 2> $O/c7_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c7_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['forward_kernel']['kernel_ms'],d['config'].get('bf16_vs_f32'))"; done
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/c7_pytest.log; tail -5 $O/c7_pytest.log
