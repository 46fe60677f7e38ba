#!/usr/bin/env python3
"""Probe: the paint step (memset + 6 kernels) replayed from a hipGraph (torch.cuda.CUDAGraph capture of the library's launches)
against plain stream launches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import Context

nside, eps = 1024, 10.0
cosmo = dict(syn.COSMO)
bg = Background(cosmo)
zax, Max, rax, T = syn.pressure_table()
for halos in (1_000_000, 100_000):
    ra, dec, M, z = syn.catalog(halos, seed=42)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx = Context(0)
        md = ctx.massdef_struct(bg, None)
        with np.errstate(all="ignore"):
            table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
        d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
        spline = ctx.da_spline(bg, float(np.max(z)))
        d_map = ctx.zeros(12 * nside * nside)
        sargs = ctx.shell_args(nside, d_cat, halos, 4, 0, eps, md)

        def step():
            d_map.zero_()
            ctx.paint_shell(sargs, table, spline, d_map)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        K = 30
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        plain = (time.perf_counter() - t0) / K
        ref = d_map.clone()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                step()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / K
            same = bool(torch.allclose(d_map, ref, rtol=1e-12, atol=0))
            print(f"{halos} halos: stream launches {plain*1e3:.3f} ms/step, graph replay {graph*1e3:.3f} ms/step, same map {same}", flush=True)
        except Exception as e:
            print(f"{halos} halos: stream launches {plain*1e3:.3f} ms/step; graph capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
