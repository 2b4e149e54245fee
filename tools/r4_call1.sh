# round 4, GPU call 1: the restructured cfconv kernels + the new 4096-bead pins + the reworked bench legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export MDG_TEST_REPORT=$PWD/$O/c1_test_report.txt; rm -f $MDG_TEST_REPORT
(timeout 1500 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_schnet.py tests/test_gpu_verlet.py tests/test_gpu_config5.py tests/test_gpu_secondary_pins.py -m gpu -q -x --durations=8 2>&1 | tail -40) > $O/c1_pytest.log; tail -5 $O/c1_pytest.log
unset MDG_TEST_REPORT
python tools/kbench_cfconv.py > $O/c1_kbench_f32.txt 2>&1; python tools/kbench_cfconv.py --bf16 > $O/c1_kbench_bf16.txt 2>&1; cat $O/c1_kbench_bf16.txt
(timeout 900 python bench.py --workload schnet4096 --bf16 --steps 10 --warmup 2 > $O/c1_bench_schnet_bf16.json 2> $O/c1_bench_schnet_bf16.err); tail -c 300 $O/c1_bench_schnet_bf16.err; python -c "
import json;d=json.load(open('$O/c1_bench_schnet_bf16.json'));print('bf16',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['step_roof']['frac']);print(json.dumps(d.get('cpu_baseline'))[:1500])"
(timeout 600 python bench.py --workload schnet4096 --steps 8 --warmup 2 --no-cpu-baseline > $O/c1_bench_schnet_f32.json 2> $O/c1_bench_schnet_f32.err); python -c "
import json;d=json.load(open('$O/c1_bench_schnet_f32.json'));print('f32',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['step_roof']['frac'])"
