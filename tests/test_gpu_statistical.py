"""Long-horizon behaviour against the one known answer the reference itself holds (VERDICT r4 missing #5 / SURVEY 4 item 5):
data/LJ_data/rdf_rho0.845_T1.0_dt0.01.csv and vacf_rho0.845_T1.0_dt0.01.csv, the equilibrium g(r) and velocity
autocorrelation of its 256-atom LJ liquid (scripts/fit_rdf_pair.py:159-204; fixture tests/golden/lj_liquid_reference.npz, made
by tests/golden/make_lj_reference_fixture.py).  Every other pin is <= 50 steps; this one runs 10^4 steps through
`Simulations` epochs -- checkpoints, wrapping across epochs, thermostat state carried over, the fine-grid RDF, the vacf
reduction -- on 256 replicas of the fused trajectory kernels, and compares the ensemble averages within sampling noise."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import DEV

pytestmark = pytest.mark.gpu


def test_lj_liquid_equilibrium_rdf_and_vacf_match_the_references_own_data():
    from mdgrad_amd import potentials as P
    from mdgrad_amd.interface import PairPotentials
    from mdgrad_amd.md import NoseHooverChain, Simulations
    from mdgrad_amd.observable import rdf, vacf
    from mdgrad_amd.system import System, FaceCenteredCubic
    ref = load_golden("lj_liquid_reference")
    rho, T, dt = float(ref["rho"]), float(ref["T"]), float(ref["dt"])
    a = (4.0 / rho) ** (1.0 / 3.0)                         # scripts/data.py get_unit_len(rho, N_unitcell = 4)
    base = System(FaceCenteredCubic("H", (4, 4, 4), a), device=DEV)
    assert len(base) == int(ref["n_atoms"])
    R = 256
    system = base.replicate(R)
    system.set_temperature(T, rng=np.random.default_rng(2024))          # independent Maxwell-Boltzmann velocities per replica
    pot = PairPotentials(system, P.LennardJones(1.0, 1.0), cutoff=float(ref["cutoff"])).to(DEV)
    integ = NoseHooverChain(pot, system, Q=float(ref["Q"]), T=T, num_chains=int(ref["chains"]), adjoint=True).to(DEV)
    sim = Simulations(system, integ)
    obs = rdf(system, nbins=100, r_range=(0.75, 3.3))
    vobs = vacf(system, t_range=60)
    skip, epochs = 50, 100                                 # (the reference: 200 epochs, 50 skipped, ONE system)
    g_sum, c_sum, n = torch.zeros(100, device=DEV), torch.zeros(60, device=DEV), 0
    with torch.no_grad():
        for ep in range(epochs):
            v_t, q_t, pv_t = sim.simulate(100, dt=dt, frequency=100)
            assert q_t.shape == (100, R * 256, 3)
            if ep >= skip:
                g_sum += obs(q_t[-1:])[2]
                c_sum += vobs(v_t)
                n += 1
    assert torch.isfinite(q_t).all()
    g, c = (g_sum / n).cpu().numpy(), (c_sum / n).cpu().numpy()
    assert len(sim.log["positions"]) == epochs and sim.log["positions"][-1].shape == (R * 256, 3)
    # the state the epochs hand on stays inside the cell (wrapped checkpoints) and at the thermostat's temperature
    L = 4 * a
    chk = sim.get_check_point()[1]
    assert float(chk.min()) >= -1e-4 and float(chk.max()) <= L + 1e-4
    kT = float((v_t[-1].pow(2).sum(-1) * 1.008).mean() / 3.0)
    assert abs(kT - T) < 0.02, "kinetic temperature of the last frame: %.4f" % kT
    dg = g - ref["g"]
    # sampling noise of the REFERENCE's curve (150 correlated frames of one 256-atom system) is ~1-2 % of g at the first
    # peak (2.63); ours (50 epochs x 256 replicas) is 10x smaller.  Observed on MI355X: max |dg| 0.03, rms 0.009.
    assert np.abs(dg).max() < 0.08 and np.sqrt((dg ** 2).mean()) < 0.03, (np.abs(dg).max(), np.sqrt((dg ** 2).mean()))
    assert abs(g.max() - ref["g"].max()) < 0.06 and abs(int(g.argmax()) - int(ref["g"].argmax())) <= 1, "first peak"
    ok = np.isfinite(ref["vacf"])                         # (the reference's file holds nan from lag 50 on)
    assert ok[:50].all() and ok.sum() == 50
    dc = (c - ref["vacf"])[ok]
    report = {"max_abs_dg": float(np.abs(dg).max()), "rms_dg": float(np.sqrt((dg ** 2).mean())), "g_peak": float(g.max()),
              "g_peak_reference": float(ref["g"].max()), "max_abs_dvacf": float(np.abs(dc).max()), "kT_last_frame": kT,
              "replicas": R, "epochs_sampled": n, "steps": epochs * 99}
    try:                                                   # (kept with the round's evidence when run through gpurun)
        import json, os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):
            json.dump(report, open(os.path.join(out, "statistical_lj.json"), "w"))
    except OSError:
        pass
    assert np.abs(dc).max() < 0.03, ("vacf", report)
    assert abs(c[0] - T / 1.008) < 0.02, "vacf(0) = <v_x^2> = kT / m"
