#!/usr/bin/env python3
"""Where PaintProfilesShell.process() spends its time at the headline size (host catalog on the device already, host map out):
the body of process() with a clock after every phase, for several slice counts.  usage: paint_api_probe.py [n_halo]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.engine import get_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nside = 1024
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(n, seed=42)
zax, Max, rax, T = syn.pressure_table()
model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
R = bfg.PaintProfilesShell(Cat, Shell, 10, model, verbose=False)
ctx = get_context()
npix = 12 * nside * nside


def body(slices):
    t = [time.perf_counter()]
    h = torch.empty(npix, dtype=torch.float64, pin_memory=True)
    d_map = ctx.empty(npix)
    main, side = torch.cuda.current_stream(ctx.device), ctx.copy_stream()
    t.append(time.perf_counter())

    def on_slice(k, n_, lo, hi):
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            h[lo:hi].copy_(d_map[lo:hi], non_blocking=True)
    ctx.stats_reset()
    R.process_device(d_map=d_map, overwrite=True, slices=slices, on_slice=on_slice, sync_stats=False)
    t.append(time.perf_counter())
    main.synchronize()
    t.append(time.perf_counter())
    side.synchronize()
    t.append(time.perf_counter())
    R.collect_stats()
    t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])] + [1e3 * (t[-1] - t[0])]


for slices in (1, 4, 8, 12, 16):
    for _ in range(3):
        body(slices)
    rows = np.array([body(slices) for _ in range(15)])
    med = np.median(rows, 0)
    print(f"slices {slices:2d}: alloc {med[0]:.3f}  enqueue {med[1]:.3f}  kernels-done +{med[2]:.3f}  copies-done +{med[3]:.3f}  stats {med[4]:.3f}"
          f"  total {med[5]:.3f} ms (min {rows[:, 5].min():.3f})", flush=True)
for s in (8, 16):
    os.environ["BFG_D2H_SLICES"] = str(s)
    for _ in range(3):
        R.process()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); R.process(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"process() with BFG_D2H_SLICES={s}: median {np.median(ts):.3f} ms, min {min(ts):.3f}", flush=True)
