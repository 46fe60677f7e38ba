#!/usr/bin/env python3
"""GPU box: the regrid kernel's differential path against its general path (BFG_REGRID=general) on the offsets of real catalogs:
time per launch and the largest difference between the two output maps (relative to the map's largest value)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

ctx = get_context(0)
cosmo = dict(syn.COSMO)
bg = Background(cosmo)
md = ctx.massdef_struct(bg, None)
for nside, n in ((1024, 10_000), (1024, 100_000), (1024, 1_000_000), (2048, 1_250_000), (256, 50_000)):
    npix = 12 * nside * nside
    ra, dec, M, z = syn.catalog(n, seed=42)
    d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
    spline = ctx.da_spline(bg, float(np.max(z)))
    zax, Max, rax, T = syn.displacement_table()
    table = ctx.table([zax, Max, rax], T, log_values=False)
    d_off = ctx.zeros(npix, 3)
    d_in = ctx.to_device(syn.mass_map(nside))
    sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, md, model_md=md, model_epsilon_max=20.0, out_overwrite=True)
    ctx.baryonify_offsets(sargs, table, spline, d_off)
    outs = {}
    for mode in ("", "general", "all"):
        if mode:
            os.environ["BFG_REGRID"] = mode
        else:
            os.environ.pop("BFG_REGRID", None)
        d_map = ctx.zeros(npix)
        d_sums = ctx.zeros(2)
        for _ in range(3):
            d_map.zero_(); ctx.regrid_shell(nside, d_off, d_in, d_map, d_sums)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            d_map.zero_(); ctx.regrid_shell(nside, d_off, d_in, d_map, d_sums)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        outs[mode] = (d_map.cpu().numpy(), d_sums.cpu().numpy(), ms)
    ref = outs["all"][0]
    moved = int(torch.count_nonzero(d_off.abs().sum(dim=1)).item())
    line = f"NSIDE {nside} halos {n:8d} displaced pixels {moved / npix:6.1%}:"
    for mode in ("", "general", "all"):
        m, sums, ms = outs[mode]
        line += f"  [{mode or 'default'}] {ms:6.3f} ms (incl. 1 memset) max|diff|/max {np.max(np.abs(m - ref)) / ref.max():.1e} sum(dep)/sum(in)-1 {sums[1] / sums[0] - 1:+.1e}"
    print(line, flush=True)
    del d_off, d_in, d_cat
