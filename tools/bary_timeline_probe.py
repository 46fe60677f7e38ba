#!/usr/bin/env python3
"""Timeline of one BaryonifyShell.process() call at BASELINE configs[2] with a page-locked input map: the single-shell path of
Runners.HealpixRunner._baryonify_pipelined re-enacted with an event after every step (host time of each enqueue, device time of
each slice's upload / regrid / download)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import engine, synthetic as syn
from baryonforge_amd.engine import get_context
from baryonforge_amd.Runners.HealpixRunner import _regrid_band_groups

cosmo = dict(syn.COSMO)
nside, n = 1024, 100_000
npix = 12 * nside * nside
ra, dec, M, z = syn.catalog(n, seed=42)
zd, Md, rd, d = syn.displacement_table()
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
flat = engine.pinned_copy(syn.mass_map(nside))
R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=flat, cosmo=cosmo), 10, bm, verbose=False)
ctx = get_context()
dev = ctx.device
up, down = ctx.upload_stream(), ctx.copy_stream()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mode = sys.argv[2] if len(sys.argv) > 2 else "kernel"      # download by copy kernel or by DMA
upmode = sys.argv[3] if len(sys.argv) > 3 else "dma"      # upload by DMA or by a copy kernel reading the mapped host memory
import ctypes as C
from baryonforge_amd import _lib
h_in = torch.from_numpy(flat)


def kernel_copy(dst, src):
    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(ctx.lib.bfg_copy_to_mapped_host(ctx.handle, C.c_void_p(stream), C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                               src.numel() * 8), "copy")
h = torch.empty(npix, dtype=torch.float64, pin_memory=True)


def one(trace):
    main = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record(main)
    marks = []

    def mark(name, stream):
        e = torch.cuda.Event(enable_timing=True); e.record(stream)
        marks.append((name, e, (time.perf_counter() - t0) * 1e3))
    d_off = R.offsets_device(sync_stats=False)
    mark("offsets enqueued", main)
    d_out = ctx.zeros(npix); d_small = ctx.zeros(4)
    cuts_b, cuts_p = _regrid_band_groups(nside, S)
    Sn = len(cuts_p) - 1
    with torch.cuda.stream(up):
        d_orig = torch.empty(npix, dtype=torch.float64, device=dev)
    prev = None
    for sl in range(Sn):
        lo, hi = cuts_p[sl], cuts_p[sl + 1]
        with torch.cuda.stream(up):
            if upmode == "kernel":
                kernel_copy(d_orig[lo:hi], h_in[lo:hi])
            else:
                d_orig[lo:hi].copy_(h_in[lo:hi], non_blocking=True)
            ev_up = torch.cuda.Event(); ev_up.record(up)
            mark(f"up {sl}", up)
        main.wait_event(ev_up)
        ctx.regrid_shell_bands(nside, d_off, d_orig, d_out, d_small[1:], cuts_b[sl], cuts_b[sl + 1])
        ev_rg = torch.cuda.Event(); ev_rg.record(main)
        mark(f"regrid {sl}", main)
        if sl >= 1:
            down.wait_event(ev_rg)
            with torch.cuda.stream(down):
                a, b = cuts_p[sl - 1], cuts_p[sl]
                if mode == "kernel":
                    ctx.copy_to_pinned(h[a:b], d_out[a:b])
                else:
                    h[a:b].copy_(d_out[a:b], non_blocking=True)
                mark(f"down {sl - 1}", down)
        prev = ev_rg
    down.wait_event(prev)
    with torch.cuda.stream(down):
        a, b = cuts_p[Sn - 1], cuts_p[Sn]
        if mode == "kernel":
            ctx.copy_to_pinned(h[a:b], d_out[a:b])
        else:
            h[a:b].copy_(d_out[a:b], non_blocking=True)
        mark(f"down {Sn - 1}", down)
    t_enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    if trace:
        print(f"slices {Sn}, upload by {upmode}, download by {mode}: host enqueue done at {t_enq:.2f} ms, everything done at {t_all:.2f} ms")
        for name, e, th in marks:
            print(f"   {name:18s} enqueued at {th:5.2f} ms (host)   completed at {ev0.elapsed_time(e):5.2f} ms (device)")
    return t_all


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(3):
        one(False)
    print(f"slices {S} up {upmode} down {mode}: best of 7: %.2f ms" % min(one(False) for _ in range(7)))
    if os.environ.get("TRACE"):
        one(True)
