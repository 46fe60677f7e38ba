#!/usr/bin/env python3
"""Probe: consecutive shells on two contexts / two HIP streams (own workspaces and map buffers), so that shell k + 1's
halo preparation and binning kernels run beside shell k's tile kernel.  Prints ms per shell for 1 and 2 streams."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import Context

halos = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nside, eps = 1024, 10.0
cosmo = dict(syn.COSMO)
bg = Background(cosmo)
ra, dec, M, z = syn.catalog(halos, seed=42)
recs = np.stack([M, z, ra, dec], axis=1)
zax, Max, rax, T = syn.pressure_table()
lanes = []
for k in range(2):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = Context(0)
        md = ctx.massdef_struct(bg, None)
        with np.errstate(all="ignore"):
            table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
        d_cat = ctx.to_device(recs)
        spline = ctx.da_spline(bg, float(np.max(z)))
        d_map = ctx.zeros(12 * nside * nside)
        sargs = ctx.shell_args(nside, d_cat, halos, 4, 0, eps, md)
    lanes.append((s, ctx, table, spline, d_map, sargs, d_cat))
torch.cuda.synchronize()

def step(k):
    s, ctx, table, spline, d_map, sargs, _ = lanes[k]
    with torch.cuda.stream(s):
        d_map.zero_()
        ctx.paint_shell(sargs, table, spline, d_map)

for nstream in (1, 2, 1, 2):
    for i in range(4):
        step(i % nstream)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for i in range(K):
        step(i % nstream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{nstream} stream(s): {dt*1e3:.3f} ms per shell -> {halos/dt:.3e} halos/s", flush=True)
