#!/usr/bin/env python3
"""Periodic-grid runners at scale: PaintProfilesGrid + BaryonifyGrid on a 2D 4096^2 map (1e5 halos) and a 3D 512^3 map
(2e4 halos); times the C-ABI calls with inputs resident in HBM."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

cosmo = dict(syn.COSMO)
ctx = get_context(0)
bg = Background(cosmo)
md = ctx.massdef_struct(bg, None)
zax, Max, rax, T = syn.pressure_table()
ptab = ctx.table([zax, Max, rax], np.log(T), log_values=True)
zd, Md, rd, d = syn.displacement_table()
dtab = ctx.table([zd, Md, rd], d, log_values=False)
a = 1 / 1.25
for ndim, N, L, nhalo, eps in ((2, 4096, 2000.0, 100000, 10.0), (3, 512, 1000.0, 20000, 5.0)):
    rng = np.random.default_rng(5)
    H = rng.uniform(0, L, (nhalo, 3)).astype(">f4").astype(np.float64)
    hM = (10 ** rng.uniform(13.0, 15.3, nhalo)).astype(">f4")
    halos = ctx.to_device(np.stack([hM.astype(np.float64), np.log(hM).astype(np.float64), H[:, 0], H[:, 1], H[:, 2]], axis=1))
    bins = ctx.to_device((np.arange(N) + 0.5) * (L / N))
    npx = N ** ndim
    d_map, d_off = ctx.zeros(npx), ctx.zeros(npx, ndim)
    d_in = torch.rand(npx, dtype=torch.float64, device=ctx.device)
    d_out = ctx.zeros(npx)
    pa = ctx.grid_args(ndim, N, bins, halos, a, eps, md)
    ba = ctx.grid_args(ndim, N, bins, halos, a, eps, md, model_md=md, model_epsilon_max=20.0)
    for rep in range(3):
        d_map.zero_(); d_off.zero_(); d_out.zero_(); ctx.stats_reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.paint_grid(pa, ptab, d_map); torch.cuda.synchronize()
        t1 = time.perf_counter(); ctx.baryonify_grid_offsets(ba, dtab, d_off); torch.cuda.synchronize()
        t2 = time.perf_counter(); ctx.regrid_grid(ndim, N, d_off, d_in, d_out); torch.cuda.synchronize()
        t3 = time.perf_counter()
    st = ctx.stats()
    print(f"{ndim}D {N}^{ndim} map, {nhalo} halos, eps {eps:g}: paint {1e3*(t1-t0):.2f} ms ({nhalo/(t1-t0):.3e} halos/s), "
          f"offsets {1e3*(t2-t1):.2f} ms, regrid {1e3*(t3-t2):.2f} ms; window pixels (both passes) {st['pixel_updates']:.4g}; "
          f"mass in/out {float(d_in.sum()):.8e} / {float(d_out.sum()):.8e}")
