#!/usr/bin/env python3
"""host-side profile of PaintProfilesShell.process_device() / process() at the headline size (GPU box)"""
import cProfile, os, pstats, sys, time
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
for _ in range(3):
    R.process_device()
torch.cuda.synchronize()
for name, fn in (("process_device", R.process_device), ("process", R.process)):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        out = fn()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(14)
