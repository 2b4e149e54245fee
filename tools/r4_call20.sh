# round 4, GPU call 20: rows16 dual reverse sweep at three waves per SIMD (QR = 4): kernel timings + rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python tools/kbench_cfconv.py --rows16 > $O/c20_kbench_rows16.txt 2>&1; head -12 $O/c20_kbench_rows16.txt
for v in bf16-rows bf16 bf16-rows bf16; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 16 --warmup 6 --no-cpu-baseline > $O/c20_bench_schnet_$v.json 2> $O/c20_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c20_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'])"; done
(timeout 600 python bench.py --workload schnet4096 --steps 12 --warmup 6 --no-cpu-baseline > $O/c20_bench_schnet_f32.json 2> $O/c20_bench_schnet_f32.err); python -c "
import json;d=json.load(open('$O/c20_bench_schnet_f32.json'));print('schnet f32',d['value'],d['ms_per_step'],d['roofline']['step_roof']['frac'])"
(timeout 900 python -m pytest tests/test_gpu_schnet_rows16.py tests/test_gpu_fused_block.py -m gpu -q -x 2>&1 | tail -5) > $O/c20_pytest.log; tail -3 $O/c20_pytest.log
