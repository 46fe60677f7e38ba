#!/usr/bin/env python3
"""GPU box: PaintProfilesShell.process() (host map out) against the number of slices the map leaves the GPU in"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
n, nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1024
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(n, seed=42)
zax, Max, rax, T = syn.pressure_table()
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                           bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False)
ctx = bfg.engine.get_context()
d = ctx.zeros(12 * nside * nside)
h = torch.empty(12 * nside * nside, dtype=torch.float64, pin_memory=True)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); print(f"plain D2H of the map: {(time.perf_counter() - t0) * 1e3:.3f} ms")
for sl in (1, 2, 4, 8, 16):
    os.environ["BFG_D2H_SLICES"] = str(sl)
    out = R.process(); out = R.process()
    t0 = time.perf_counter()
    for _ in range(10):
        out = R.process()
    print(f"slices {sl:2d}: process() {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
buf = np.empty(12 * nside * nside)
t0 = time.perf_counter()
for _ in range(5):
    R.process(out=buf)
print(f"process(out=pageable array): {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
