"""
GPU tests at the resolution and size of the two BASELINE configurations the rest of the suite only reaches in miniature:

  * config 4 (8 x MI355X BaryonifyShell, 1e7 halos, NSIDE 2048): ONE rank's share on one GPU -- the full runner (offsets +
    regrid) at NSIDE 2048 against the oracle on a catalog the oracle finishes in seconds, and the 1.25e6-halo per-rank
    share through properties the path offers at any size: mass conservation (HealpixRunner.py:368-370), finite output,
    independence of the halo order, P_tot equal to the oracle's count on a sample of the same catalog;
  * config 5 (BaryonifySnapshot, 512^3 particles, 1e5 halos, CIC deposit): the full size through properties -- deposited
    mass = number of particles, coordinates inside [0, L), particles outside every halo's sphere bit-unchanged (checked on
    a sample against a KD-tree of the halos), the two particle-pass variants agree.
Reference anchors: Runners/HealpixRunner.py:357-365, Runners/SnapshotRunner.py:217-273, utils/io.py:629-677.
"""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from oracle import oracle as orc
from util import assert_maps_close, oracle_baryonify

RTOL = 1e-5


def test_baryonify_shell_nside2048_full_runner_vs_oracle(cosmo):
    """BASELINE config[3]'s resolution: offsets + regrid of a 5.0e7-pixel mass map, 2 500 halos incl. both poles and the
    phi = 0 seam, against the oracle (the regrid alone is 5e7 get_interpol calls on the CPU)"""
    nside, n, eps = 2048, 2500, 10.0
    ra, dec, M, z = syn.catalog(n, seed=12)
    dec[:40] = 90 - np.abs(np.random.default_rng(1).normal(0, 0.3, 40))
    dec[40:80] = -90 + np.abs(np.random.default_rng(2).normal(0, 0.3, 40))
    ra[80:120] = np.random.default_rng(3).normal(0, 0.05, 40) % 360
    zd, Md, rd, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    m_in[::11] = 0.0
    ref = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, model, verbose=False).process()
    assert_maps_close(got, ref, RTOL, floor=1e-9, what="BaryonifyShell nside 2048")
    assert np.isclose(got.sum(), m_in.sum(), rtol=1e-10)
    assert np.count_nonzero(~np.isclose(got, m_in)) > 100000          # the halos did move mass


def test_baryonify_shell_config4_per_rank_share_properties(cosmo):
    """1.25e6 halos at NSIDE 2048 (= one of eight ranks of BASELINE config[3]; 1.39e9 pixel-updates): properties"""
    nside, n, eps = 2048, 1_250_000, 10.0
    npix = 12 * nside * nside
    ra, dec, M, z = syn.catalog(n, seed=42)
    zd, Md, rd, d = syn.displacement_table()
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    m_in = syn.mass_map(nside)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = R.process()                                            # asserts sum(new) ~ sum(old) itself (:368-370)
    assert got.shape == (npix,) and np.all(np.isfinite(got)) and got.min() >= 0.0
    assert np.isclose(got.sum(), m_in.sum(), rtol=1e-9)
    assert R.last_stats["fallback_halos"] == 0 and R.last_stats["halos_out_of_table"] == 0
    ptot_all = R.last_stats["pixel_updates"]
    assert 1.2e9 < ptot_all < 1.6e9
    changed = np.count_nonzero(~np.isclose(got, m_in, rtol=1e-12, atol=0))
    assert changed > 0.5 * npix                                       # 1.25e6 discs of ~1100 pixels cover most of the sky
    # the offset field is a sum over halos: any order of the catalog gives the same map (to summation rounding)
    perm = np.random.default_rng(5).permutation(n)
    Cat2 = bfg.HaloLightConeCatalog(ra[perm], dec[perm], M[perm], z[perm], cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got2 = bfg.BaryonifyShell(Cat2, bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False).process()
    assert_maps_close(got2, got, 1e-9, floor=1e-12, what="halo order")
    del got2
    # P_tot: the kernels' count on a sample of the same catalog equals the oracle's; and the full count is the sample's
    # scaled up to a few per cent (same mass function)
    ns = 20000
    a, Rr, D = orc.halo_scalars(cosmo, M[:ns], z[:ns])
    _, ptot_ref = orc.baryonify_offsets(nside, ra[:ns], dec[:ns], M[:ns], a, D, Rr, Rr / a, (zd, Md, rd), d, eps, 20.0, False, None)
    Rs = bfg.BaryonifyShell(bfg.HaloLightConeCatalog(ra[:ns], dec[:ns], M[:ns], z[:ns], cosmo),
                            bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Rs.offsets_device()
    assert Rs.last_stats["pixel_updates"] == ptot_ref
    assert abs(ptot_all / n - ptot_ref / ns) < 0.05 * ptot_ref / ns


def test_baryonify_shell_config4_whole_catalog_one_gpu(cosmo):
    """BASELINE config[3] on ONE GPU (the N = 1 anchor of the 8-GPU configuration): the whole 1e7-halo catalog at NSIDE 2048,
    1.1e10 pixel-updates -- mass conserved (HealpixRunner.py:368-370), finite and non-negative output, no halo through the
    scatter fallback, P_tot equal to the oracle's count on a 2e4-halo sample of the same catalog and the full count the
    sample's scaled up (same mass function), and the eighth of it that is rank 0's shard reproduces that shard's own run
    when the offsets are summed over the eight shards (the linearity the multi-GPU join relies on, :355)."""
    import torch
    from baryonforge_amd import sharding
    from baryonforge_amd.engine import get_context
    nside, n, eps = 2048, 10_000_000, 10.0
    npix = 12 * nside * nside
    ra, dec, M, z = syn.catalog(n, seed=42)
    zd, Md, rd, d = syn.displacement_table()
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    m_in = syn.mass_map(nside)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = R.process()
    assert got.shape == (npix,) and np.all(np.isfinite(got)) and got.min() >= 0.0
    assert np.isclose(got.sum(), m_in.sum(), rtol=1e-9)
    assert R.last_stats["fallback_halos"] == 0 and R.last_stats["halos_out_of_table"] == 0
    ptot_all = R.last_stats["pixel_updates"]
    ns = 20000
    a, Rr, D = orc.halo_scalars(cosmo, M[:ns], z[:ns])
    _, ptot_ref = orc.baryonify_offsets(nside, ra[:ns], dec[:ns], M[:ns], a, D, Rr, Rr / a, (zd, Md, rd), d, eps, 20.0, False, None)
    Rs = bfg.BaryonifyShell(bfg.HaloLightConeCatalog(ra[:ns], dec[:ns], M[:ns], z[:ns], cosmo),
                            bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False)
    Rs._spline_z_max = float(z.max())                               # the whole catalog's D_A spline, as a shard would use
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Rs.offsets_device()
    assert Rs.last_stats["pixel_updates"] == ptot_ref
    assert abs(ptot_all / n - ptot_ref / ns) < 0.03 * ptot_ref / ns
    # linearity over sky-patch shards: sum of the eight shards' offset fields == the whole catalog's (to summation rounding)
    w = sharding.estimate_disc_pixels(cosmo, M, z, eps, nside)
    shards = sharding.shard_by_sky_patch(ra, dec, w, 8)
    assert sum(s.size for s in shards) == n
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d_full = R.offsets_device()
        d_sum = torch.zeros_like(d_full)
        for idx in shards:
            Rk = bfg.BaryonifyShell(bfg.HaloLightConeCatalog(ra[idx], dec[idx], M[idx], z[idx], cosmo),
                                    bfg.LightconeShell(map=m_in, cosmo=cosmo), eps, model, verbose=False)
            Rk._spline_z_max = float(z.max())
            d_sum += Rk.offsets_device()
    scale = float(d_full.abs().max())
    assert float((d_sum - d_full).abs().max()) < 1e-9 * scale
    del d_full, d_sum
    get_context().synchronize()


def test_snapshot_config5_full_size_properties(cosmo, monkeypatch):
    """BASELINE config[4]: 512^3 particles (a jittered lattice, built on the device), 1e5 halos, L = 1000 Mpc, eps = 10,
    + CIC deposit on a 512^3 grid"""
    import torch
    from scipy.spatial import cKDTree
    from baryonforge_amd.background import Background
    from baryonforge_amd.engine import get_context
    n1, nhalo, L, zs, eps, ngrid = 512, 100_000, 1000.0, 0.25, 10.0, 512
    ctx = get_context(0)
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    ax = (torch.arange(n1, device=dev, dtype=torch.float64) + 0.5) * (L / n1)
    P = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
    P = (P + (torch.rand(P.shape, generator=g, device=dev, dtype=torch.float64) - 0.5) * (L / n1)) % L
    npart = P.shape[0]
    rng = np.random.default_rng(3)
    H = rng.uniform(0, L, (nhalo, 3)).astype(">f4").astype(np.float64)        # float32 halo columns, as io.py:204
    hM = (10 ** rng.uniform(13.0, 15.3, nhalo)).astype(">f4")
    halos = np.stack([hM.astype(np.float64), np.log(hM).astype(np.float64), H[:, 0], H[:, 1], H[:, 2]], axis=1)
    d_halo = ctx.to_device(halos)
    zax, Max, rax, d = syn.displacement_table()
    table = ctx.table([zax, Max, rax], d, log_values=False)
    bg = Background(cosmo)
    md = ctx.massdef_struct(bg, None)
    a = 1 / (1 + zs)
    outs = {}
    for path in ("direct", "cell"):
        monkeypatch.setenv("BFG_SNAPSHOT", path)
        d_out = torch.empty_like(P)
        ctx.stats_reset()
        ctx.baryonify_snapshot(P, d_halo, 3, L, a, eps, md, md, 20.0, False, 0, table, d_out)
        outs[path] = (d_out, ctx.stats()["pixel_updates"])
    monkeypatch.delenv("BFG_SNAPSHOT")
    new, pairs = outs["direct"]
    assert outs["cell"][1] == pairs and 1e8 < pairs < 1e9                     # (halo, particle) pairs: same set both ways
    diff = (outs["cell"][0] - new).abs()
    diff = torch.minimum(diff, L - diff)
    assert float(diff.max()) < 1e-9                                           # the two particle passes agree (summation order only)
    del outs, diff
    assert bool(torch.isfinite(new).all()) and float(new.min()) >= 0.0 and float(new.max()) <= L   # wrapped into the box (:263-273)
    shift = (new - P).abs()
    shift = torch.minimum(shift, L - shift).amax(dim=1)
    moved = shift > 0
    n_moved = int(moved.sum())
    assert 0.02 * npart < n_moved < 0.9 * npart and float(shift.max()) < 5.0   # displacements are sub-Mpc .. Mpc
    # particles outside every halo's query sphere keep their coordinates bit for bit; checked on a 2e6-particle sample
    # against a periodic KD-tree of the HALOS (R_q = clip(eps R / a, 0, L / 2), SnapshotRunner.py:222-228)
    ns = 2_000_000
    idx = torch.randperm(npart, device=dev, generator=g)[:ns]
    Ps, Ns = P[idx].cpu().numpy(), new[idx].cpu().numpy()
    Rq = np.clip(eps * orc.get_radius(cosmo, hM.astype(np.float64), a) / a, 0, L / 2)
    tree = cKDTree(H % L, boxsize=L)
    # candidates within the largest radius, then the halo's own radius with the reference's periodic distance
    inside = np.zeros(ns, dtype=bool)
    for lo in range(0, ns, 250_000):
        sl = slice(lo, lo + 250_000)
        cand = tree.query_ball_point(Ps[sl], float(Rq.max()) * 1.0000001, return_sorted=False)
        for k, c in enumerate(cand):
            if c:
                c = np.asarray(c)
                dd = Ps[lo + k] - H[c]
                dd = np.where(dd > L / 2, dd - L, dd)
                dd = np.where(dd < -L / 2, dd + L, dd)
                inside[lo + k] = np.any(np.sqrt(np.sum(dd * dd, axis=1)) <= Rq[c] * (1 + 1e-12))
    unchanged = np.all(Ns == Ps, axis=1)
    assert np.all(unchanged[~inside]), "a particle outside every halo sphere moved"
    assert np.count_nonzero(~unchanged) > 0.01 * ns
    # CIC deposit of the displaced particles (unit masses): the grid holds exactly the particles' mass; NGP too
    for mode in ("cic", "ngp"):
        grid = ctx.deposit_grid(new, None, L, ngrid, mode)
        assert grid.shape == (ngrid,) * 3
        assert abs(float(grid.sum()) - npart) < 1e-6 * npart and float(grid.min()) >= 0.0
        del grid
