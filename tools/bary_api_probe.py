#!/usr/bin/env python3
"""where BaryonifyShell.process() spends its time at BASELINE configs[2] (1e5 halos, NSIDE 1024): host -> device, the kernels,
device -> host; and a list of shells through SimpleParallel (one GPU)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.engine import get_context

cosmo = dict(syn.COSMO)
nside, n = 1024, 100_000
ra, dec, M, z = syn.catalog(n, seed=42)
zd, Md, rd, d = syn.displacement_table()
m_in = syn.mass_map(nside)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bm, verbose=False)
ctx = get_context()
sync = torch.cuda.synchronize


def t(fn, reps=5):
    fn(); sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); sync(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


ms, _ = t(lambda: R.process())
print(f"BaryonifyShell.process(): {ms:.2f} ms")
flat = np.ascontiguousarray(m_in, dtype=np.float64).ravel()
ms_up, d_orig = t(lambda: ctx.to_device(flat))
ms_red, _ = t(lambda: ctx.absmax_sum(d_orig))
ms_off, d_off = t(lambda: R.offsets_device())
d_out = ctx.zeros(12 * nside * nside)
ms_rg, _ = t(lambda: (d_out.zero_(), ctx.regrid_shell(nside, d_off, d_orig, d_out, None)))
ms_dn, _ = t(lambda: ctx.to_host(d_out))
print(f"  upload {ms_up:.2f}  absmax+sum {ms_red:.2f} (x2)  offsets {ms_off:.2f}  regrid {ms_rg:.2f}  download {ms_dn:.2f}")
runners = [bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bm, verbose=False) for _ in range(8)]
ms8, outs = t(lambda: bfg.SimpleParallel(runners).process(), reps=3)
print(f"SimpleParallel(8 BaryonifyShell).process(): {ms8:.2f} ms = {ms8 / 8:.2f} ms per shell")
assert all(np.isclose(o.sum(), m_in.sum()) for o in outs)
