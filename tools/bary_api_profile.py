#!/usr/bin/env python3
"""host-side profile of BaryonifyShell.process() at BASELINE configs[2] with a page-locked map (GPU box)"""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
nside = 1024
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(100_000, seed=42)
zd, Md, rd, d = syn.displacement_table()
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
B = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=syn.mass_map(nside), cosmo=cosmo, pinned=True), 10, bm, verbose=False)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(3):
        B.process()
    t0 = time.perf_counter()
    for _ in range(20):
        B.process()
    print(f"process(): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        B.process()
    pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(16)
