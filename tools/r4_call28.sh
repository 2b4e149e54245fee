# round 4, GPU call 28: NVE over a GNN -- cached-force forward, analytic verlet adjoint, graph replay: tests + step rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_schnet.py -m gpu -q -x -k "nve" 2>&1 | tail -15) > $O/c28_nve.log; tail -5 $O/c28_nve.log
python - <<'P' 2>&1 | grep -v Warning | tail -8
import time, torch, numpy as np, sys
sys.path.insert(0, '.')
import bench
from mdgrad_amd import units
from mdgrad_amd.md import NVE
from mdgrad_amd.sovlers import odeint_adjoint
dev = torch.device('cuda:0')
for size in (2, 4):
    wl = bench.build_schnet_workload(dev, 1, False, 5, size=size)
    integ = NVE(wl['integ'].model, wl['system']).to(dev)
    t = torch.Tensor([units.fs * i for i in range(21)]).to(dev)
    for analytic in (False, True):
        integ.analytic_verlet = analytic
        def one():
            for p in integ.parameters(): p.grad = None
            y0 = [s.clone().requires_grad_(True) for s in integ.get_inital_states(wrap=True)]
            v_t, q_t = odeint_adjoint(integ, tuple(y0), t, method='verlet')
            (q_t[-1].pow(2).mean() + v_t[::2].pow(2).mean()).backward()
        for _ in range(3): one()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): one()
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 5
        print('NVE + SchNet, %4d beads, 20 steps fwd + adjoint: %-38s %7.1f MD steps/s' % (wl['N'], 'analytic verlet + graph replay' if analytic else 'generic solver (reference control flow)', 20 / el), flush=True)
P
