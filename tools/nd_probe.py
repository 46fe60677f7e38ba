#!/usr/bin/env python3
"""Tables with more p_keys axes than the shell kernels read, on the tile path: step time of bfg_paint_shell / bfg_baryonify_offsets on
the headline catalog (1e6 halos, NSIDE 1024, inputs resident) for the 3-D table and for 4 / 5 extra axes of 3 nodes each
(values independent of the extra axes: every run must give the 3-D map)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nside = 1024
npix = 12 * nside * nside
ctx = get_context(0)
bg = Background(dict(syn.COSMO))
md = ctx.massdef_struct(bg, None)
ra, dec, M, z = syn.catalog(n, seed=42)
spline = ctx.da_spline(bg, float(z.max()))
rng = np.random.default_rng(1)


def run(kind, n_extra):
    if kind == "paint":
        zax, Max, rax, T = syn.pressure_table()
        with np.errstate(all="ignore"):
            V = np.log(T)
    else:
        zax, Max, rax, V = syn.displacement_table()
    ax = [np.array([0.0, 0.5, 1.0])] * n_extra
    VN = np.ascontiguousarray(np.broadcast_to(V.reshape(V.shape + (1,) * n_extra), V.shape + (3,) * n_extra))
    table = ctx.table([zax, Max, rax] + ax, VN, log_values=(kind == "paint"))
    cols = [M, z, ra, dec] + [rng.uniform(0, 1, n) for _ in range(n_extra)]
    d_cat = ctx.to_device(np.stack(cols, axis=1))
    if kind == "paint":
        d_out = ctx.empty(npix)
        a = ctx.shell_args(nside, d_cat, n, 4 + n_extra, n_extra, 10.0, md, out_overwrite=True)
        step = lambda: ctx.paint_shell(a, table, spline, d_out)
    else:
        d_out = ctx.empty(npix, 3)
        a = ctx.shell_args(nside, d_cat, n, 4 + n_extra, n_extra, 10.0, md, model_md=md, model_epsilon_max=20.0, out_overwrite=True)
        step = lambda: ctx.baryonify_offsets(a, table, spline, d_out)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ctx.stats_reset()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    st = ctx.stats()
    return ms, d_out.clone(), st["pixel_updates"] // reps, st["fallback_halos"] // reps


only = os.environ.get("ND_PROBE_ONLY")        # e.g. paint:4 -- one configuration (for a kernel profile)
for kind in ("paint", "bary"):
    ref = None
    for n_extra in (0, 1, 3, 4, 5):
        if only and only != f"{kind}:{n_extra}":
            continue
        ms, out, ptot, fb = run(kind, n_extra)
        if ref is None:
            ref = out
            cmp = ""
        else:
            nz = ref != 0
            rel = float(((out - ref).abs()[nz] / ref.abs()[nz]).max()) if kind == "paint" else float((out - ref).abs().max() / ref.abs().max())
            cmp = f" max rel diff vs 3-D {rel:.2e} same non-zero set {bool(torch.equal(out != 0, ref != 0))}"
        print(f"{kind:5s} n={n} extra axes {n_extra}: {ms:8.3f} ms per step  P_tot {ptot} fallback {fb}{cmp}", flush=True)
