# round 4, GPU call 6: whole GPU suite (masked ring kernels, pair_ell dispatch, everything since call 2), masked-ring rate
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 600 python -m pytest tests/test_gpu_pins.py -m gpu -q -k "mask" 2>&1 | tail -30) > $O/c6_mask.log; tail -5 $O/c6_mask.log
python tools/kbench_ring_mask.py > $O/c6_ring_mask.txt 2>&1; tail -2 $O/c6_ring_mask.txt
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/c6_pytest.log; tail -5 $O/c6_pytest.log
(timeout 600 python bench.py --workload schnet4096 --bf16 --steps 12 --warmup 2 --no-cpu-baseline > $O/c6_bench_schnet_bf16.json 2> $O/c6_bench_schnet_bf16.err); python -c "
import json;d=json.load(open('$O/c6_bench_schnet_bf16.json'));print('schnet bf16',d['value'],d['ms_per_step'])"
