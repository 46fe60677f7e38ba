#!/usr/bin/env python3
"""
Bytes every rank has to SEND in the one exchange step of the multi-GPU path, for the alternatives to a full all-reduce,
computed from real sky-patch shards of the synthetic catalogs (no GPU needed; numpy + scipy only):

  A  all-reduce of the replicated output (what SplitJoinParallel does; ring: 2 (N-1)/N S per rank)
  B  reduce-scatter only -- the output stays distributed, rank r owns pixel range r (ring: (N-1)/N S)
  C  contiguous layout, owner-computes: every rank paints one compact sky region; only what its discs deposit OUTSIDE the
     region it owns is sent to the owners ("border", dense blocks of NSIDE_b pixels), then
       C1  all-gather of the owned parts if every rank needs the whole map ((N-1)/N S)
       C0  nothing more if the output may stay distributed
  for the interleaved layout (every rank covers the whole sky) the "border" is the whole map: C degenerates to B.

S = 8 Npix bytes for PaintProfilesShell; BaryonifyShell exchanges the offset field (24 Npix: reduce-scatter in A, border in
C) and afterwards the regridded map (8 Npix: all-reduce in A; in C the deposits that cross a region border: bounded by the
same border blocks).

The border is measured on a coarse raster: a block (= pixel of NSIDE_b = 128, i.e. 8 x 8 pixels at NSIDE 1024, 16 x 16 at
2048) counts as touched by a rank if the nearest halo of a radius class of that rank lies within the class's largest disc
radius of the block centre plus the block's half-diagonal (cKDTree per rank and radius class; an upper bound at block
granularity).

  python tools/exchange_bytes.py [headline|config4|both]  ->  markdown table on stdout
"""
import os
import sys
import time

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from baryonforge_amd import sharding, synthetic as syn            # noqa: E402
from baryonforge_amd.background import Background                 # noqa: E402

NSIDE_B = 128
N_CLASSES = 12


def ring_pixel_centres(nside):
    """(ra, dec) in degrees of every RING pixel centre (closed-form ring formulae)"""
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    p = np.arange(npix, dtype=np.int64)
    z = np.empty(npix)
    phi = np.empty(npix)
    north = p < ncap
    i = ((1 + np.sqrt(1 + 2 * p[north].astype(np.float64))) / 2).astype(np.int64)
    i = np.where(2 * i * (i - 1) > p[north], i - 1, i)
    i = np.where(2 * (i + 1) * i <= p[north], i + 1, i)
    j = p[north] - 2 * i * (i - 1)
    z[north] = 1 - i * i / (3.0 * nside * nside)
    phi[north] = (j + 0.5) * (np.pi / 2) / i
    belt = (p >= ncap) & (p < npix - ncap)
    q = p[belt] - ncap
    i = q // (4 * nside) + nside
    j = q % (4 * nside)
    z[belt] = (2 * nside - i) * 2.0 / (3.0 * nside)
    phi[belt] = (j + 0.5 * (((i - nside) & 1) == 0)) * (np.pi / 2) / nside
    south = p >= npix - ncap
    q = npix - 1 - p[south]                                     # mirror of the north cap
    i = ((1 + np.sqrt(1 + 2 * q.astype(np.float64))) / 2).astype(np.int64)
    i = np.where(2 * i * (i - 1) > q, i - 1, i)
    i = np.where(2 * (i + 1) * i <= q, i + 1, i)
    j = 4 * i - 1 - (q - 2 * i * (i - 1))
    z[south] = -(1 - i * i / (3.0 * nside * nside))
    phi[south] = (j + 0.5) * (np.pi / 2) / i
    return np.degrees(phi), np.degrees(np.arcsin(z))


def unit(ra, dec):
    r, d = np.radians(ra), np.radians(dec)
    return np.stack([np.cos(d) * np.cos(r), np.cos(d) * np.sin(r), np.sin(d)], axis=1)


def disc_radius(cosmo, M, z, eps):
    bg = Background(cosmo)
    a = 1 / (1 + z)
    R = (M / (4.18879020479 * 200 * bg.rho_x(a, "critical"))) ** (1 / 3)
    zg = np.linspace(0, z.max() * 1.001 + 1e-3, 512)
    return R * eps / np.interp(z, zg, bg.angular_diameter_distance(1 / (1 + zg)))


def owners_of_blocks(ra, dec, w, world, nside_patch, bra, bdec):
    """owner rank of every coarse block under the contiguous layout (the rule of sharding.shard_by_sky_patch)"""
    patch = sharding.ang2pix_nest(nside_patch, ra, dec)
    npatch = 12 * nside_patch ** 2
    w_patch = np.bincount(patch, weights=w, minlength=npatch)
    cum = np.cumsum(w_patch)
    mid = cum - 0.5 * w_patch
    owner_of_patch = np.maximum.accumulate(np.minimum((mid / cum[-1] * world).astype(np.int64), world - 1))
    return owner_of_patch[sharding.ang2pix_nest(nside_patch, bra, bdec)]


def border_blocks(vec_blocks, owner_block, halo_vec, theta, shards, half_diag):
    """per rank: number of coarse blocks OUTSIDE its own region that its discs reach"""
    edges = np.geomspace(theta.min(), theta.max() * (1 + 1e-12), N_CLASSES + 1)
    cls = np.clip(np.searchsorted(edges, theta, side="right") - 1, 0, N_CLASSES - 1)
    out = []
    for r, idx in enumerate(shards):
        foreign = np.flatnonzero(owner_block != r)
        touched = np.zeros(foreign.size, dtype=bool)
        for c in range(N_CLASSES):
            sel = idx[cls[idx] == c]
            if sel.size == 0:
                continue
            reach = edges[c + 1] + half_diag                       # angular reach of the class from a block centre
            chord = 2 * np.sin(min(reach, np.pi) / 2)
            todo = np.flatnonzero(~touched)
            d, _ = cKDTree(halo_vec[sel]).query(vec_blocks[foreign[todo]], k=1, distance_upper_bound=chord)
            touched[todo[np.isfinite(d)]] = True
        out.append(int(touched.sum()))
    return out


def run(name, n_halo, nside, eps, out):
    cosmo = dict(syn.COSMO)
    t0 = time.time()
    ra, dec, M, z = syn.catalog(n_halo, seed=42)
    theta = disc_radius(cosmo, M, z, eps)
    w = sharding.estimate_disc_pixels(cosmo, M, z, eps, nside)
    bra, bdec = ring_pixel_centres(NSIDE_B)
    vb, hv = unit(bra, bdec), unit(ra, dec)
    nblock = bra.size
    px_per_block = (nside // NSIDE_B) ** 2
    half_diag = np.sqrt(4 * np.pi / nblock)                          # ~ block side: a generous half-diagonal
    npix = 12 * nside * nside
    MB = 1e-6
    out.append(f"\n**{name}**: {n_halo:.0e} halos, NSIDE {nside}, eps {eps:g}; map S = {8 * npix * MB:.0f} MB, offset field "
               f"{24 * npix * MB:.0f} MB; border blocks of {nside // NSIDE_B} x {nside // NSIDE_B} pixels\n")
    out.append("| N | A paint: all-reduce | B paint: reduce-scatter only | C0 paint: border only (max rank / mean) | "
               "C1 paint: border + all-gather | A baryonify: RS(offsets) + AR(map) | C1 baryonify: borders + all-gather(map) |")
    out.append("|---|---|---|---|---|---|---|")
    for world in (2, 4, 8):
        shards = sharding.shard_by_sky_patch(ra, dec, w, world, nside_patch=8, layout="contiguous")
        owner = owners_of_blocks(ra, dec, w, world, 8, bra, bdec)
        nb = border_blocks(vb, owner, hv, theta, shards, half_diag)
        border = np.array(nb, dtype=np.float64) * px_per_block * 8.0          # bytes of map values a rank sends to owners
        f = (world - 1) / world
        S = 8.0 * npix
        A = 2 * f * S
        B = f * S
        C0mx, C0mean = border.max(), border.mean()
        C1 = C0mx + f * S
        A_b = f * 3 * S + 2 * f * S
        C1_b = 3 * C0mx + C0mx + f * S                                # offsets border + regrid deposits across the border + all-gather
        out.append(f"| {world} | {A * MB:.0f} MB | {B * MB:.0f} MB | {C0mx * MB:.1f} / {C0mean * MB:.1f} MB "
                   f"({100 * max(nb) / nblock:.1f} % of the sky) | {C1 * MB:.0f} MB | {A_b * MB:.0f} MB | {C1_b * MB:.0f} MB |")
        print(f"[{name}] N={world} done after {time.time() - t0:.0f} s", file=sys.stderr, flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    lines = []
    if which in ("headline", "both"):
        run("headline (BASELINE metric)", 1_000_000, 1024, 10.0, lines)
    if which in ("config4", "both"):
        run("BASELINE config 4", 10_000_000, 2048, 10.0, lines)
    print("\n".join(lines))
