#!/usr/bin/env python3
"""A/B of library switches in ONE process, over the bench workloads (inputs resident in HBM, like bench.py).

  python tools/quick_bench.py [--modes VAR=a,b,...] [--workloads paint1e6,paint1e5,bary1e5,steep,bary1e6,eps20,n2048] [--reps 2]

Default: BFG_TILE_LIGHT=0,1.  The library reads its switches with getenv at every call, so the modes are toggled
in-process; for every workload the first mode's output is the reference the others are compared with (maximum relative
difference on its non-zero pixels, identical non-zero sets, identical P_tot)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from baryonforge_amd import synthetic as syn  # noqa: E402
from baryonforge_amd.background import Background  # noqa: E402
from baryonforge_amd.engine import get_context  # noqa: E402

WORK = {
    "paint1e6": dict(kind="paint", n=1_000_000, nside=1024, eps=10.0),
    "paint1e5": dict(kind="paint", n=100_000, nside=1024, eps=10.0),
    "bary1e5": dict(kind="bary", n=100_000, nside=1024, eps=10.0),
    "bary1e6": dict(kind="bary", n=1_000_000, nside=1024, eps=10.0),
    "steep": dict(kind="paint", n=1_000_000, nside=1024, eps=10.0, steep=True),
    "eps20": dict(kind="paint", n=1_000_000, nside=1024, eps=20.0),
    "n2048": dict(kind="paint", n=1_250_000, nside=2048, eps=10.0),
    "stress": dict(kind="paint", n=1_000_000, nside=1024, eps=10.0, shape=(2, 30, 2000)),
    "paint1e4": dict(kind="paint", n=10_000, nside=1024, eps=10.0),
    "paint3e4": dict(kind="paint", n=30_000, nside=1024, eps=10.0),
    "paint3e5": dict(kind="paint", n=300_000, nside=1024, eps=10.0),
    "bary1e4": dict(kind="bary", n=10_000, nside=1024, eps=10.0),
    "bary3e5": dict(kind="bary", n=300_000, nside=1024, eps=10.0),
    "bary2048": dict(kind="bary", n=1_250_000, nside=2048, eps=10.0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="BFG_TILE_LIGHT=0,1")
    ap.add_argument("--out-zero", action="store_true", help="set BFG_SHELL_OUT_IS_ZERO (the step clears the output first)")
    ap.add_argument("--overwrite", action="store_true", help="BFG_SHELL_OUT_OVERWRITE: no clearing pass, the call defines the output")
    ap.add_argument("--workloads", default="paint1e6,paint1e5,bary1e5,steep")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-timing", action="store_true", help="no per-kernel HIP events in the timed steps (the step time without them)")
    a = ap.parse_args()
    # "VAR=a,b" (one variable, several values) or "A=1 B=2;A=3" (';'-separated sets of assignments; '-' = nothing set)
    if ";" in a.modes or " " in a.modes or a.modes == "-":
        sets = [dict(kv.split("=") for kv in m.split()) if m.strip() != "-" else {} for m in a.modes.split(";")]
    else:
        var, vals = a.modes.split("=")
        sets = [{var: v} for v in vals.split(",")]
    allvars = sorted({k for st in sets for k in st})
    vals = [" ".join(f"{k}={v}" for k, v in st.items()) or "-" for st in sets]
    ctx = get_context(0)
    cosmo = dict(syn.COSMO)
    bg = Background(cosmo)
    md = ctx.massdef_struct(bg, None)
    for name in a.workloads.split(","):
        w = WORK[name]
        n, nside, npix = w["n"], w["nside"], 12 * w["nside"] ** 2
        ra, dec, M, z = syn.catalog(n, seed=42, steep=w.get("steep", False))
        d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
        spline = ctx.da_spline(bg, float(np.max(z)))
        shape = w.get("shape", (10, 30, 100))
        if w["kind"] == "paint":
            zax, Max, rax, T = syn.pressure_table(*shape)
            with np.errstate(all="ignore"):
                table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
            d_out = ctx.zeros(npix)
            sargs = ctx.shell_args(nside, d_cat, n, 4, 0, w["eps"], md, out_is_zero=a.out_zero, out_overwrite=a.overwrite)

            def step():
                if not a.overwrite:
                    d_out.zero_()
                ctx.paint_shell(sargs, table, spline, d_out)
        else:
            zax, Max, rax, T = syn.displacement_table(*shape)
            table = ctx.table([zax, Max, rax], T, log_values=False)
            d_out = ctx.zeros(npix, 3)
            d_in = ctx.to_device(syn.mass_map(nside))
            d_map = ctx.zeros(npix)
            sargs = ctx.shell_args(nside, d_cat, n, 4, 0, w["eps"], md, model_md=md, model_epsilon_max=20.0,
                                   out_is_zero=a.out_zero, out_overwrite=a.overwrite)

            def step():
                if not a.overwrite:
                    d_out.zero_()
                d_map.zero_()
                ctx.baryonify_offsets(sargs, table, spline, d_out)
                ctx.regrid_shell(nside, d_out, d_in, d_map, None)
        ref = None
        for rep in range(a.reps):
            for v, st_env in zip(vals, sets):
                for k in allvars:
                    os.environ.pop(k, None)
                os.environ.update(st_env)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                ctx.stats_reset()
                ctx.timing_enable(not a.no_timing)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.steps * 1e3
                st = ctx.stats()
                k = ctx.timing_read(1); p = ctx.timing_read(0); b = ctx.timing_read(3); l = ctx.timing_read(4); r = ctx.timing_read(2); f = ctx.timing_read(5)
                ctx.timing_enable(False)
                ms = lambda x: x[0] / max(x[1], 1)
                ptot = st["pixel_updates"] // a.steps
                per_px = 16.0 if w["kind"] == "paint" else 48.0
                frac = (32.0 * n + per_px * ptot) / (ms(k) * 1e-3) / 8e12 if ms(k) > 0 else 0
                extra = ""
                if rep == 0:
                    out = d_out.cpu().numpy()
                    extra = f" | sum {out.sum():.12e} nonzero {np.count_nonzero(out)} P_tot {ptot}"
                    if ref is None:
                        ref = (out, ptot)
                    else:
                        nz = ref[0] != 0
                        rel = np.max(np.abs(out[nz] - ref[0][nz]) / np.abs(ref[0][nz])) if nz.any() else 0.0
                        extra = (f" | vs {vals[0]}: max rel {rel:.2e} nonzero-set equal {np.array_equal(out != 0, nz)} "
                                 f"P_tot equal {ptot == ref[1]}")
                print(f"{name:9s} {v:40s} step {dt:7.3f} ms  kernel {ms(k):7.3f}  prep {ms(p):6.3f}  bin {ms(b):6.3f}  "
                      f"left {ms(l):6.3f}  defer {ms(f):6.3f}  regrid {ms(r):6.3f}  frac {frac:5.3f}  fallback {st['fallback_halos'] // a.steps}{extra}",
                      flush=True)
        del d_cat, d_out


if __name__ == "__main__":
    main()
