#!/usr/bin/env python3
"""End-to-end timing of the Python API (host catalog in, host map out) for the headline workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nside = 1024
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(n, seed=42)
zax, Max, rax, T = syn.pressure_table()
model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
R = bfg.PaintProfilesShell(Cat, Shell, 10, model, verbose=False)
for _ in range(2):
    R.process()
for rep in range(3):
    t0 = time.perf_counter(); out = R.process(); t1 = time.perf_counter()
    d = R.process_device(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"process(): {1e3*(t1-t0):.2f} ms ({n/(t1-t0):.3e} halos/s)   process_device(): {1e3*(t2-t1):.2f} ms", flush=True)
zd, Md, rd, dd = syn.displacement_table()
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, dd, cosmo, epsilon_max=20)
import warnings
B = bfg.BaryonifyShell(Cat[:100000] if n >= 100000 else Cat, bfg.LightconeShell(map=syn.mass_map(nside), cosmo=cosmo), 10, bm, verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    B.process()
    for rep in range(2):
        t0 = time.perf_counter(); B.process(); t1 = time.perf_counter()
        print(f"BaryonifyShell(1e5).process(): {1e3*(t1-t0):.2f} ms", flush=True)
