#!/usr/bin/env python3
"""One-off scale check beyond 2^31 pixels: PaintProfilesShell at NSIDE 16384 (3.2e9 pixels, 25.8 GB map) against the oracle
on a few hundred halos (poles and the phi = 0 seam included): pixel-update count, non-zero set, values."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from oracle import oracle as orc

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 300
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(n, seed=21)
dec[:20] = 90 - np.abs(np.random.default_rng(1).normal(0, 0.05, 20))
dec[20:40] = -90 + np.abs(np.random.default_rng(2).normal(0, 0.05, 20))
ra[40:60] = np.random.default_rng(3).normal(0, 0.01, 20) % 360
zax, Max, rax, T = syn.pressure_table()
a, R, D = orc.halo_scalars(cosmo, M, z)
t0 = time.perf_counter()
with np.errstate(all="ignore"):
    ref, ptot = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), 10.0)
print(f"oracle: {time.perf_counter()-t0:.1f} s, P_tot {ptot}", flush=True)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
for variant in ("auto", "scatter_wave"):
    Run = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10.0, model,
                                 verbose=False, variant=variant)
    t0 = time.perf_counter()
    got = Run.process()
    print(f"{variant}: process() {time.perf_counter()-t0:.1f} s, pixel_updates {Run.last_stats['pixel_updates']}", flush=True)
    assert Run.last_stats["pixel_updates"] == ptot
    nz = np.flatnonzero(ref)
    assert np.count_nonzero(got) == nz.size
    rel = np.max(np.abs(got[nz] - ref[nz]) / np.abs(ref[nz]))
    print(f"{variant}: non-zero pixels {nz.size} (highest index {nz.max()}), max rel err {rel:.2e}", flush=True)
    assert rel < 1e-5
    del got
print("ok")
