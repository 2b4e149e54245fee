# round 4, GPU call 17: bias-column gradient (no neighbour sums / colsum job), half_fill with 4 atoms per workgroup: tests + rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out; R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_fused_block.py -m gpu -q -x -k "bias_gradient or row_chain_path" 2>&1 | tail -15) > $O/c17_b2.log; tail -4 $O/c17_b2.log
for v in bf16 bf16-rows bf16 bf16-rows; do (timeout 600 python bench.py --workload schnet4096 --$v --steps 12 --warmup 2 --no-cpu-baseline > $O/c17_bench_schnet_$v.json 2> $O/c17_bench_schnet_$v.err); python -c "
import json;d=json.load(open('$O/c17_bench_schnet_$v.json'));print('schnet $v',d['value'],d['ms_per_step'])"; done
(timeout 600 python bench.py --workload schnet4096 --steps 8 --warmup 2 --no-cpu-baseline > $O/c17_bench_schnet_f32.json 2> $O/c17_bench_schnet_f32.err); python -c "
import json;d=json.load(open('$O/c17_bench_schnet_f32.json'));print('schnet f32',d['value'],d['ms_per_step'],d['roofline']['step_roof']['frac'])"
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/c17_pytest.log; tail -5 $O/c17_pytest.log
