#!/usr/bin/env python3
"""Does the placement of the 101 MB output map matter?  The main bench loop paints into ONE buffer allocated early in the process; the
multi_model leg rotates five buffers allocated later and ran the same tile kernel 6 % faster in the same process on the same box.
Same catalog, same table, same kernels: only the output buffer differs."""
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
first_map = ctx.zeros(npix)                                     # as bench.py: allocated before the catalog goes up
ra, dec, M, z = syn.catalog(n, seed=42)
d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
bg = Background(cosmo)
spline, md = ctx.da_spline(bg, float(np.max(z))), ctx.massdef_struct(bg, None)
zax, Max, rax, T = syn.pressure_table()
table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, out_overwrite=True)


def run(maps, steps=40, label=""):
    for m in maps:
        ctx.paint_shell(sargs, table, spline, m)
    torch.cuda.synchronize()
    ctx.timing_enable(True, which=[1])
    t0 = time.perf_counter()
    for i in range(steps):
        ctx.paint_shell(sargs, table, spline, maps[i % len(maps)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    k = ctx.timing_read(1)
    ctx.timing_enable(False)
    print(f"{label:58s} step {dt:6.3f} ms   tile kernel {k[0] / max(k[1], 1):6.3f} ms   ptr % 2MiB {[m.data_ptr() % (2 << 20) for m in maps][:3]}", flush=True)


for _ in range(3):                                              # ramp
    run([first_map], 60, "(ramp) the early buffer")
late = ctx.empty(npix)
five = [ctx.empty(npix) for _ in range(5)]
slab = torch.empty(1 << 27, dtype=torch.float64, device=ctx.device)           # 1 GiB
views = [slab[i * (1 << 24):i * (1 << 24) + npix] for i in range(3)]
for rep in range(2):
    run([first_map], label="one buffer allocated early (the bench's main loop)")
    run([late], label="one buffer allocated late")
    run(five, label="five buffers in rotation (the multi_model leg)")
    run([five[2]], label="one of those five, alone")
    run([views[1]], label="a view into a 1 GiB slab")
    run(views, label="three views into the slab in rotation")
