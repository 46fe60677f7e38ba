#!/usr/bin/env python3
"""End-to-end timing of BaryonifySnapshot.process() + make_map through the Python API (host arrays in and out)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, nhalo, zs = 500.0, 20000, 0.25
cosmo = dict(syn.COSMO)
rng = np.random.default_rng(3)
ax = (np.arange(n1) + 0.5) * (L / n1)
P = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)
P = (P + (rng.random(P.shape) - 0.5) * (L / n1)) % L
H = rng.uniform(0, L, (nhalo, 3)); hM = 10 ** rng.uniform(13.0, 15.0, nhalo)
zax, Max, rax, d = syn.displacement_table()
model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=H[:, 2])
Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2], M=np.ones(P.shape[0]), L=L, redshift=zs, cosmo=cosmo)
R = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=10, model=model, verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        t0 = time.perf_counter(); new = R.process(); t1 = time.perf_counter()
        dd = R.process_device(); torch.cuda.synchronize(); t2 = time.perf_counter()
        S2 = bfg.ParticleSnapshot.from_catalog(new, L, zs, cosmo); t3 = time.perf_counter()
        m = S2.make_map(n1, mode="cic", device=True); t4 = time.perf_counter()
        print(f"{n1}^3 particles: process() {1e3*(t1-t0):.0f} ms, process_device() {1e3*(t2-t1):.0f} ms, new snapshot object {1e3*(t3-t2):.0f} ms, "
              f"make_map(device) {1e3*(t4-t3):.0f} ms; sum {m.sum():.6e}", flush=True)
