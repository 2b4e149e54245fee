"""The reference's own long-horizon known answer for the LJ liquid (SURVEY 4 item 5 / VERDICT r4 missing #5):
data/LJ_data/rdf_rho0.845_T1.0_dt0.01.csv and vacf_rho0.845_T1.0_dt0.01.csv, written by scripts/fit_rdf_pair.py:159-204
(get_target_obs: FCC 4^3 = 256 atoms at rho 0.845, LJ(1, 1), cutoff 2.5, NoseHooverChain(Q = 50, 5 chains, T = 1.0), dt 0.01,
200 epochs of 100 steps, the first 50 skipped; g(r) of each epoch's last frame on linspace(0.75, 3.3, 100), vacf over each
epoch's 100 frames, both averaged over the epochs).  Data only -- two small tables -- copied into a fixture:

    python tests/golden/make_lj_reference_fixture.py        # in the build container (reads /root/reference)
"""
import os
import numpy as np

REF = "/root/reference/data/LJ_data"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    rdf = np.loadtxt(os.path.join(REF, "rdf_rho0.845_T1.0_dt0.01.csv"), delimiter=",")
    vacf = np.loadtxt(os.path.join(REF, "vacf_rho0.845_T1.0_dt0.01.csv"), delimiter=",")
    assert rdf.shape == (2, 100) and vacf.shape == (60,)
    np.savez(os.path.join(HERE, "lj_liquid_reference.npz"), r=rdf[0].astype(np.float64), g=rdf[1].astype(np.float64),
             vacf=vacf.astype(np.float64), rho=0.845, T=1.0, dt=0.01, n_atoms=256, cutoff=2.5, Q=50.0, chains=5,
             epochs=200, skip=50, steps_per_epoch=100)
    print("wrote lj_liquid_reference.npz", rdf.shape, vacf.shape)
