#!/usr/bin/env python3
"""Displacement-table builder at the reference's default table size (30 z x 30 M x 100 r, N_int = 500):
host (scipy, as the reference) against bfg_build_displacement_table.  Densities come from an analytic stand-in."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background, MassDef

bgd = Background(dict(syn.COSMO))
md = MassDef(200, "critical")


class Prof(object):
    def __init__(self, core, slope, ring=0.0):
        self.core, self.slope, self.ring, self.cutoff = core, slope, ring, None

    def set_parameter(self, k, v):
        setattr(self, k, v)

    def projected(self, cosmo, r, M, a):
        M = np.atleast_1d(M); r = np.atleast_1d(r)
        R = (md.get_radius(dict(syn.COSMO), M, a) / a)[:, None]
        x = r[None, :] / (self.core * R)
        S = M[:, None] / (2 * np.pi * (self.core * R) ** 2) * (1 + x * x) ** (-self.slope) * np.exp(-r[None, :] / (30 * R))
        return S * (1 + self.ring * np.sin(6 * np.log(r))[None, :] * (r[None, :] / R) ** 1.5)


res = {}
for dev in (False, True, True):
    B = bfg.Baryonification2D(Prof(0.25, 1.6), Prof(0.45, 1.6, ring=0.6), dict(syn.COSMO), epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        B.setup_interpolator(verbose=False, device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[dev] = B.raw_input_d
    print(f"setup_interpolator(device={dev}): {dt*1e3:9.1f} ms for a table of shape {B.raw_input_d.shape}")
print("max |device - host| / max|host| =", np.abs(res[True] - res[False]).max() / np.abs(res[False]).max())
# the launch alone (densities already on the device)
from baryonforge_amd.engine import get_context
from baryonforge_amd.Profiles.BaryonCorrection import _integration_grid
ctx = get_context(0)
r = np.geomspace(1e-3, 1e2, 100)
r_int, _ = _integration_grid(r, 1e-6, 1000, 500)
M = np.geomspace(1e12, 1e16, 30)
zs = np.geomspace(1e-2, 5, 30)
So = np.concatenate([Prof(0.25, 1.6).projected(None, r_int, M, 1 / (1 + z)) / (1 + z) for z in zs])
Sb = np.concatenate([Prof(0.45, 1.6, 0.6).projected(None, r_int, M, 1 / (1 + z)) / (1 + z) for z in zs])
for _ in range(3):
    t0 = time.perf_counter(); ctx.build_displacement_table(2, r_int, So, Sb, r); dt = time.perf_counter() - t0
    print(f"bfg_build_displacement_table, {So.shape[0]} rows x {r_int.size} -> {r.size}: {dt*1e3:.2f} ms incl. upload/download")
