#!/usr/bin/env python3
"""host-side timeline of _baryonify_pipelined's phases (where the host thread spends a list of BaryonifyShell runners)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.engine import get_context

cosmo = dict(syn.COSMO)
nside, n = 1024, 100_000
npix = 12 * nside * nside
ra, dec, M, z = syn.catalog(n, seed=42)
zd, Md, rd, d = syn.displacement_table()
m_in = syn.mass_map(nside)
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bm, verbose=False)
ctx = get_context()
dev = ctx.device
R.process()
up, down = ctx.upload_stream(), ctx.copy_stream()
main = torch.cuda.current_stream(dev)
flat = np.ascontiguousarray(m_in).ravel()
hs = [torch.empty(npix, dtype=torch.float64, pin_memory=True) for _ in range(3)]
pin_src = torch.empty(npix, dtype=torch.float64, pin_memory=True)
pin_src.copy_(torch.from_numpy(flat))
torch.cuda.synchronize()
for mode in ("pageable", "pinned source", "pageable, no download", "pinned, no download", "pageable, kernel download", "pinned, kernel download"):
    marks = []
    torch.cuda.synchronize()
    t00 = time.perf_counter()
    evs = []
    for k in range(8):
        t0 = time.perf_counter()
        d_off = R.offsets_device(sync_stats=False)
        t1 = time.perf_counter()
        with torch.cuda.stream(up):
            if mode.startswith("pinned"):
                d_orig = pin_src.to(dev, non_blocking=True)
            else:
                d_orig = torch.from_numpy(flat).to(dev)
        main.wait_stream(up)
        d_orig.record_stream(main)
        t2 = time.perf_counter()
        d_out = ctx.zeros(npix)
        ctx.regrid_shell(nside, d_off, d_orig, d_out, None)
        t3 = time.perf_counter()
        if "no download" not in mode:
            down.wait_stream(main)
            with torch.cuda.stream(down):
                if "kernel download" in mode:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(down)
                    ctx.copy_to_pinned(hs[k % 3], d_out)
                    e1.record(down)
                    evs.append((e0, e1))
                else:
                    hs[k % 3].copy_(d_out, non_blocking=True)
            d_out.record_stream(down)
        t4 = time.perf_counter()
        marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t00) * 1e3
    if "kernel download" in mode:
        assert np.allclose(hs[1].numpy(), hs[2].numpy(), rtol=1e-9, atol=1e-9) and abs(hs[1].sum().item() - m_in.sum()) < 1e-9 * m_in.sum(), (hs[1].sum().item(), m_in.sum())
    if evs:
        print("   copy kernel durations (ms):", " ".join(f"{a.elapsed_time(b):.2f}" for a, b in evs))
    m = np.array(marks[2:]).mean(axis=0) * 1e3
    print(f"{mode:24s}: {tot / 8:.2f} ms per shell; host: offsets issue {m[0]:.2f}  upload {m[1]:.2f}  regrid issue {m[2]:.2f}  download issue {m[3]:.2f}")
