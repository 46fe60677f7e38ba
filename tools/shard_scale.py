#!/usr/bin/env python3
"""Per-rank compute of the weak-scaling bench without the collective: for N = 1, 2, 4, 8 build the N x 1e6-halo catalog,
cut it into sky-patch shards exactly as bench.py does and time the paint step of a few ranks' shards on this one GPU.
Shows what sharding by sky patch does to the tile kernel (fewer, fuller tiles per rank)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baryonforge_amd import sharding, synthetic as syn
from baryonforge_amd.background import Background
from baryonforge_amd.engine import get_context

halos = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
strong = len(sys.argv) > 2 and sys.argv[2] == "strong"          # `halos` in total, cut into `world` shards
nside, eps = 1024, 10.0
cosmo = dict(syn.COSMO)
ctx = get_context(0)
bg = Background(cosmo)
md = ctx.massdef_struct(bg, None)
zax, Max, rax, T = syn.pressure_table()
with np.errstate(all="ignore"):
    table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
d_map = ctx.zeros(12 * nside * nside)
for world in (1, 2, 4, 8):
    ra, dec, M, z = syn.catalog(halos if strong else halos * world, seed=42)
    spline = ctx.da_spline(bg, float(np.max(z)))
    if world > 1:
        w = sharding.estimate_disc_pixels(cosmo, M, z, eps, nside)
        kw = {}                                          # the library's default: NSIDE-64 patches, interleaved
        if os.environ.get("LAYOUT"):
            kw["layout"] = os.environ["LAYOUT"]
            kw["nside_patch"] = int(os.environ.get("PATCH", 8 if kw["layout"] == "contiguous" else 64))
        shards = sharding.shard_by_sky_patch(ra, dec, w, world, **kw)
    else:
        shards = [np.arange(ra.size)]                    # (the caller's order, as bench.py at N = 1)
    for rank in sorted(set([0, world // 2, world - 1])):
        idx = shards[rank]
        d_cat = ctx.to_device(np.stack([M[idx], z[idx], ra[idx], dec[idx]], axis=1))
        sargs = ctx.shell_args(nside, d_cat, idx.size, 4, 0, eps, md, out_overwrite=True)
        for _ in range(3):
            ctx.paint_shell(sargs, table, spline, d_map)
        torch.cuda.synchronize()
        ctx.stats_reset(); ctx.timing_enable(True)
        t0 = time.perf_counter(); K = 10
        for _ in range(K):
            ctx.paint_shell(sargs, table, spline, d_map)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        k_ms, k_n = ctx.timing_read(1); p_ms, p_n = ctx.timing_read(0); b_ms, b_n = ctx.timing_read(3); l_ms, l_n = ctx.timing_read(4)
        st = ctx.stats(); ctx.timing_enable(False)
        print(f"N={world} rank {rank}: {idx.size} halos, {st['pixel_updates']/K:.4g} pixel-updates, step {dt*1e3:.3f} ms "
              f"(tile kernel {k_ms/max(k_n,1):.3f}, prep {p_ms/max(p_n,1):.3f}, binning {b_ms/max(b_n,1):.3f}, left-overs {l_ms/max(l_n,1) if l_n else 0:.3f}) "
              f"-> {idx.size/dt:.3e} halos/s", flush=True)
