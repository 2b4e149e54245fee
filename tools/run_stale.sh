cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stale" 2>&1 | tail -25 > gpurun_out/stale1.log
timeout 1500 python -m pytest tests/test_gpu_secondary_pins.py -x -q -m gpu -k "stale" 2>&1 | tail -40 >> gpurun_out/stale1.log
