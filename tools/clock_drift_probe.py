#!/usr/bin/env python3
"""The headline tile kernel's time over 6 s of back-to-back steps, in windows of 50 steps, from a cold start: does the box speed up after
the first tens of milliseconds, and does it slow down again under sustained load?  (What length of ramp `bench.py` should run.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

os.environ["BFG_PLAN_REUSE"] = "0"
ctx = get_context(0)
cosmo = dict(syn.COSMO)
nside, npix, n = 1024, 12 * 1024 * 1024, 1_000_000
d_map = ctx.zeros(npix)
ra, dec, M, z = syn.catalog(n, seed=42)
d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
bg = Background(cosmo)
spline, md = ctx.da_spline(bg, float(np.max(z))), ctx.massdef_struct(bg, None)
zax, Max, rax, T = syn.pressure_table()
table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, out_overwrite=True)
torch.cuda.synchronize()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)      # idle first
t_start = time.perf_counter()
out = []
while time.perf_counter() - t_start < 6.0:
    ctx.timing_enable(True, which=[1])
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.paint_shell(sargs, table, spline, d_map)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    k = ctx.timing_read(1)
    ctx.timing_enable(False)
    out.append((time.perf_counter() - t_start, dt, k[0] / max(k[1], 1)))
for i, (t, dt, k) in enumerate(out):
    if i < 12 or i % 8 == 0 or i == len(out) - 1:
        print(f"t = {t:5.2f} s   step {dt:6.3f} ms   tile kernel {k:6.3f} ms")
