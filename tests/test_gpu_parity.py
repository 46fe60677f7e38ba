"""
GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the
reference-shaped Python API -> ctypes C-ABI -> libbfg_mi355.so, against
  (1) the golden vectors captured from the reference's own code (tests/golden/), and
  (2) the CPU oracle on the same seeded inputs,
at the north-star tolerance: <= 1e-5 relative on non-zero pixels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from oracle import oracle as orc
from util import assert_maps_close, oracle_baryonify, oracle_paint

RTOL = 1e-5  # north_star: "to within 1e-5 relative on non-zero pixels"
# Regridded maps contain deposits with bilinear weights ~1e-12 that are pure rounding noise of
# acos()/atan2() near the poles (healpy's vec2ang has the same conditioning); they are compared
# absolutely, at 1e-9 of the map's maximum.
BFLOOR = 1e-9
VARIANTS = ["scatter_wave", "scatter_quarter", "tile_lds"]


def _paint_model(zax, Max, rax, T, pax=None):
    if pax is None:
        return bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
    return bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, T, other_params={"cdelta": pax})


# --------------------------------------------------------------------------- read-out
def test_table_readout_matches_reference(golden):
    g = golden("readout.npz")
    prof = bfg.TabulatedProfile.from_arrays(g["ro_zax"], g["ro_Max"], g["ro_rax"], g["ro_T2D"], g["ro_T2D"] * 2.0)
    for i, M in enumerate(g["ro_M"]):
        for j, a in enumerate(g["ro_a"]):
            for fn, ref in ((prof.projected, g["ro_projected"]), (prof.real, g["ro_real"])):
                got = fn(None, g["ro_r"], M, a)
                assert np.array_equal(np.isnan(got), np.isnan(ref[i, j]))
                m = ~np.isnan(got)
                np.testing.assert_allclose(got[m], ref[i, j][m], rtol=1e-12, atol=0)
    # scalar-argument shape semantics (Tabulate.py:321-325)
    assert np.ndim(prof.projected(None, 0.5, 1e14, 0.8)) == 0
    assert prof.projected(None, np.array([0.5, 1.0]), np.array([1e13, 1e14, 1e15]), 0.8).shape == (3, 2)


def test_param_table_readout_matches_reference(golden):
    g = golden("readout.npz")
    prof = bfg.ParamTabulatedProfile.from_arrays(g["rp_zax"], g["rp_Max"], g["rp_rax"], g["rp_T2D"],
                                                 other_params={"cdelta": g["rp_pax"]})
    for i, M in enumerate(g["rp_M"]):
        for j, c in enumerate(g["rp_cd"]):
            got = prof.projected(None, g["ro_r"], M, float(g["rp_a"]), cdelta=c)
            ref = g["rp_projected"][i, j]
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            np.testing.assert_allclose(got[~np.isnan(got)], ref[~np.isnan(ref)], rtol=1e-12, atol=0)
    with pytest.raises(AssertionError):
        prof.projected(None, g["ro_r"], 1e14, 0.8)


@pytest.mark.parametrize("tag", ["rd0", "rd1"])
def test_displacement_readout_matches_reference(golden, cosmo, tag):
    import warnings
    g = golden("readout.npz")
    disp = bfg.Baryonification2D.from_arrays(g[f"rb_{tag}_zax"], g[f"rb_{tag}_Max"], g[f"rb_{tag}_rax"], g[f"rb_{tag}_d"],
                                             cosmo, epsilon_max=float(g[f"rb_{tag}_eps"]), Rdelta_sampling=(tag == "rd1"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, M in enumerate(g["ro_M"]):
            for j, a in enumerate(g["ro_a"]):
                got = disp.displacement(g["ro_r"], M, a)
                ref = g[f"rb_{tag}_disp"][i, j]
                assert np.array_equal(np.isnan(got), np.isnan(ref))
                np.testing.assert_allclose(got[~np.isnan(got)], ref[~np.isnan(ref)], rtol=1e-11, atol=1e-300)
    with pytest.warns(UserWarning):
        disp.displacement(g["ro_r"], 1e17, 0.8)


# --------------------------------------------------------------------------- golden runner outputs
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_paint_shell_golden(golden, cosmo, tag, variant):
    g = golden("paint_shell.npz")
    nside = int(g[f"{tag}_nside"])
    Cat = bfg.HaloLightConeCatalog(g[f"{tag}_ra"], g[f"{tag}_dec"], g[f"{tag}_M"], g[f"{tag}_z"], cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    model = _paint_model(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_T2D"])
    R = bfg.PaintProfilesShell(Cat, Shell, epsilon_max=float(g[f"{tag}_eps"]), model=model,
                               include_pixel_size=bool(g[f"{tag}_ips"]), verbose=False, variant=variant)
    got = R.process()
    ref = g[f"{tag}_map"]
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert_maps_close(got, ref, RTOL, what=f"paint {tag}")
    assert np.array_equal(got != 0, ref != 0)
    assert R.last_stats["pixel_updates"] >= np.count_nonzero(ref)


def test_paint_headline_full_size_vs_oracle(cosmo):
    """The headline workload itself (bench.py defaults: 1e6 halos, NSIDE 1024, eps 10, seed 42) against the oracle run
    split-join over the host cores (Parallelize.py:255-318 analogue): every non-zero pixel within 1e-5 relative,
    identical pixel-update count and identical non-zero pixel set."""
    import os
    nside, n = 1024, 1000000
    ra, dec, M, z = syn.catalog(n, seed=42)
    zax, Max, rax, T = syn.pressure_table()
    a, R, D = orc.halo_scalars(cosmo, M, z)
    njobs = max(1, min(32, (os.cpu_count() or 1)))
    with np.errstate(all="ignore"):
        ref, ptot = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), 10, njobs=njobs)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Run = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                                 _paint_model(zax, Max, rax, T), verbose=False)
    got = Run.process()
    assert Run.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what="headline paint 1e6 halos")


def test_paint_config1_full_size_vs_oracle(cosmo):
    """BASELINE configs[1] at full size (PaintProfilesShell, 1e5 halos, NSIDE 1024, eps 10, TabulatedProfile(Pressure), seed 42) against
    the oracle, halo for halo: identical pixel-update count, identical non-zero pixel set, every non-zero pixel within 1e-5 relative"""
    import os
    nside, n = 1024, 100000
    ra, dec, M, z = syn.catalog(n, seed=42)
    zax, Max, rax, T = syn.pressure_table()
    a, R, D = orc.halo_scalars(cosmo, M, z)
    with np.errstate(all="ignore"):
        ref, ptot = orc.paint_shell(nside, ra, dec, M, a, D, R, (zax, Max, rax), np.log(T), 10, njobs=max(1, min(16, os.cpu_count() or 1)))
    Run = bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo),
                                 bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                                 _paint_model(zax, Max, rax, T), verbose=False)
    got = Run.process()
    assert Run.last_stats["pixel_updates"] == ptot and Run.last_stats["fallback_halos"] == 0
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what="configs[1]: paint 1e5 halos")


def test_baryonify_config2_full_size_vs_oracle(cosmo):
    """BASELINE config[2] at full size (1e5 halos, NSIDE 1024, eps 10, model eps 20), the whole BaryonifyShell pipeline
    (offsets + regrid) against the oracle; mass conserved as the reference asserts (HealpixRunner.py:368-370)"""
    import warnings
    nside, n = 1024, 100000
    ra, dec, M, z = syn.catalog(n, seed=42)
    zax, Max, rax, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    ref = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, nside, 10, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False).process()
    assert np.isclose(got.sum(), m_in.sum(), rtol=1e-5, atol=1e-8)
    assert_maps_close(got, ref, RTOL, floor=BFLOOR, what="baryonify config 2")


def test_paint_nside4096_and_offsets_nside2048_vs_oracle(cosmo):
    """large maps: paint at NSIDE 4096 (2.0e8 pixels, discs of ~4400 pixels) and the baryonify offset field at NSIDE 2048
    (BASELINE config[3]'s resolution) against the oracle, halos on both poles and on the phi = 0 seam"""
    nside, n, eps = 4096, 2500, 10.0
    ra, dec, M, z = syn.catalog(n, seed=12)
    dec[:40] = 90 - np.abs(np.random.default_rng(1).normal(0, 0.3, 40))
    dec[40:80] = -90 + np.abs(np.random.default_rng(2).normal(0, 0.3, 40))
    ra[80:120] = np.random.default_rng(3).normal(0, 0.05, 40) % 360
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, eps)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Run = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps,
                                 _paint_model(zax, Max, rax, T), verbose=False)
    got = Run.process()
    assert Run.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what="paint nside 4096")
    del got, ref
    nside = 2048                                                   # the offset field has 3 doubles per pixel: 1.2 GB here
    zd, Md, rd, d = syn.displacement_table()
    refo, ptot = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, None, offsets_only=True)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=np.ones(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        goto = R.offsets_device().cpu().numpy()
    assert R.last_stats["pixel_updates"] == ptot
    assert_maps_close(goto, refo, RTOL, floor=1e-10, what="offsets nside 2048")


def test_geometry_fuzz_against_oracle(cosmo):
    """Randomised geometry: odd and even NSIDE from 8 to 700, halos drawn towards the poles, the phi = 0 seam and the
    equatorial-belt / polar-cap transition, discs from sub-pixel to larger than a hemisphere (eps up to 400); every case
    must give the oracle's pixel-update count, non-zero pixel set and values for paint and baryonify, on all variants."""
    import warnings
    rng = np.random.default_rng(2024)
    zax, Max, rax, T = syn.pressure_table()
    zd, Md, rd, d = syn.displacement_table()
    for case in range(14):
        nside = int(rng.choice([8, 11, 16, 29, 64, 100, 255, 256, 513, 700]))
        n = 60
        ra = rng.uniform(0, 360, n)
        u = rng.uniform(-1, 1, n)
        dec = np.degrees(np.arcsin(u))
        k = n // 6
        dec[:k] = 90 - np.abs(rng.normal(0, 0.7, k))                       # north cap, some almost on the pole
        dec[k:2 * k] = -90 + np.abs(rng.normal(0, 0.7, k))
        ra[2 * k:3 * k] = rng.normal(0, 0.3, k) % 360                       # seam
        dec[3 * k:4 * k] = np.degrees(np.arcsin(2 / 3)) + rng.normal(0, 0.5, k)   # belt / cap transition
        dec[4 * k:5 * k] = rng.normal(0, 0.5, k)                            # equator
        M = 10 ** rng.uniform(12.5, 15.5, n)
        z = rng.uniform(0.02, 0.9, n)
        eps = float(rng.choice([0.5, 3, 10, 40, 400]))
        ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, eps)
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
        for variant in VARIANTS:
            R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps,
                                       _paint_model(zax, Max, rax, T), verbose=False, variant=variant)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = R.process()
            assert R.last_stats["pixel_updates"] == ptot, (case, nside, eps, variant)
            assert np.array_equal(got != 0, ref != 0), (case, nside, eps, variant)
            assert_maps_close(got, ref, RTOL, what=f"fuzz paint case {case} nside {nside} eps {eps} {variant}")
        m_in = syn.mass_map(nside)
        refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, m_in)
        model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, model, verbose=False).process()
        assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what=f"fuzz baryonify case {case} nside {nside} eps {eps}")


@pytest.mark.parametrize("order", ["random", "sorted"])
def test_crowded_sky_patch_splits_tiles(cosmo, order):
    """a catalog crowded into a few degrees of sky (an octant light cone or a compact multi-GPU shard in the small):
    the heavy tiles' pair lists are cut into several work items whose workgroups add to the same map tile with
    atomics, and -- sorted by position -- neighbouring lanes of the binning passes hit the same tile counter
    (wave_merged_inc); paint and baryonify against the oracle"""
    import warnings
    rng = np.random.default_rng(77)
    nside, n, eps = 512, 6000, 6.0
    ra = (40.0 + rng.normal(0, 2.0, n)) % 360
    dec = 25.0 + rng.normal(0, 2.0, n)
    M = 10 ** rng.uniform(13.0, 14.8, n)
    z = rng.uniform(0.3, 0.5, n)
    if order == "sorted":
        from baryonforge_amd import sharding
        o = np.argsort(sharding.ang2pix_nest(1024, ra, dec), kind="stable")
        ra, dec, M, z = ra[o], dec[o], M[o], z[o]
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, eps)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps,
                               _paint_model(zax, Max, rax, T), verbose=False)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what="crowded paint")
    zd, Md, rd, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, model, verbose=False).process()
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what="crowded baryonify")


def _grid_inputs(g, tag, cosmo):
    is2D = bool(g[f"{tag}_is2D"])
    N, bins, H = int(g[f"{tag}_Npix"]), g[f"{tag}_bins"], g[f"{tag}_H"]
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g[f"{tag}_hM"], float(g[f"{tag}_redshift"]), cosmo, z=None if is2D else H[:, 2])
    return is2D, N, bins, Cat


GRID_PATHS = ["direct", "small"]      # BFG_GRID: one workgroup per halo + global atomics / tile-privatised (small tiles,
                                      # so that the golden fixtures' 16..96-pixel maps are cut into several tiles)


@pytest.mark.parametrize("path", GRID_PATHS)
@pytest.mark.parametrize("tag", ["p2", "p3"])
def test_paint_grid_golden(golden, cosmo, tag, path, monkeypatch):
    """PaintProfilesGrid (Map2DRunner.py:624-829) against the reference's own run"""
    monkeypatch.setenv("BFG_GRID", path)
    g = golden("grid.npz")
    is2D, N, bins, Cat = _grid_inputs(g, tag, cosmo)
    model = bfg.TabulatedProfile.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_T2D"], g[f"{tag}_T3D"])
    Map = bfg.GriddedMap(map=np.zeros((N,) * (2 if is2D else 3)), redshift=float(g[f"{tag}_redshift"]), bins=bins, cosmo=cosmo)
    got = bfg.PaintProfilesGrid(Cat, Map, float(g[f"{tag}_eps"]), model, include_pixel_size=bool(g[f"{tag}_ips"]),
                                verbose=False).process()
    ref = g[f"{tag}_map"]
    assert got.shape == ref.shape and got.dtype == np.float64
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what=f"grid paint {tag}")


@pytest.mark.parametrize("path", GRID_PATHS)
@pytest.mark.parametrize("tag", ["b2", "b3", "x2", "x3"])
def test_baryonify_grid_golden(golden, cosmo, tag, path, monkeypatch):
    """BaryonifyGrid (Map2DRunner.py:376-621, incl. regrid_pixels_2D/_3D) against the reference's own run.
    x2 / x3: catalogues with an infinite mass, a position at +infinity and a zero mass -- rows whose NaN displacements the
    reference ADDS to their cut-out (the half box for M = inf) before the regrid zeroes non-finite offsets, so that the
    other halos' displacements on those pixels are lost too (~27 % / 8 % of the map differs from a run without the rows)"""
    monkeypatch.setenv("BFG_GRID", path)
    import warnings
    g = golden("grid.npz")
    is2D, N, bins, Cat = _grid_inputs(g, tag, cosmo)
    model = bfg.Baryonification2D.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_d"], cosmo,
                                              epsilon_max=float(g[f"{tag}_eps_model"]), Rdelta_sampling=bool(g[f"{tag}_rdelta"]))
    Map = bfg.GriddedMap(map=g[f"{tag}_map_in"].copy(), redshift=float(g[f"{tag}_redshift"]), bins=bins, cosmo=cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyGrid(Cat, Map, float(g[f"{tag}_eps"]), model, verbose=False).process()
    assert_maps_close(got, g[f"{tag}_map_out"], RTOL, floor=BFLOOR, what=f"grid baryonify {tag}")
    if tag.startswith("x"):
        clean = g[f"{tag}_map_out_without_bad_rows"]
        assert np.count_nonzero(~np.isclose(got, clean)) == np.count_nonzero(~np.isclose(g[f"{tag}_map_out"], clean)) > 100


@pytest.mark.parametrize("path", GRID_PATHS)
def test_grid_ellipticity_and_anis_golden(golden, cosmo, path, monkeypatch):
    """2D ellipticity (Map2DRunner.py:281-350) in PaintProfilesGrid / BaryonifyGrid and PaintProfilesAnisGrid (:833-1015)
    against the reference's own run"""
    monkeypatch.setenv("BFG_GRID", path)
    import warnings
    g = golden("grid.npz")
    N, bins, H = int(g["e_Npix"]), g["e_bins"], g["e_H"]
    zs, eps = float(g["e_redshift"]), float(g["e_eps"])
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g["e_hM"], zs, cosmo, q_ell=g["e_q"], A_ell=g["e_A"])
    axes = (g["e_zax"], g["e_Max"], g["e_rax"])
    mk = lambda m: bfg.GriddedMap(map=m, redshift=zs, bins=bins, cosmo=cosmo)
    paint, tracer, mtot = (bfg.TabulatedProfile.from_arrays(*axes, g[k]) for k in ("e_T_paint", "e_T_tracer", "e_T_mtot"))
    mtot.proj_cutoff = float(g["e_proj_cutoff"])
    got = bfg.PaintProfilesGrid(Cat, mk(np.zeros((N, N))), eps, paint, use_ellipticity=True, verbose=False).process()
    assert_maps_close(got, g["e_paint_ell"], RTOL, what="grid paint ellipticity")
    disp = bfg.Baryonification2D.from_arrays(g["e_zd"], g["e_Md"], g["e_rd"], g["e_d"], cosmo, epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyGrid(Cat, mk(g["e_map_in"].copy()), eps, disp, use_ellipticity=True, verbose=False).process()
    assert_maps_close(got, g["e_bary_ell"], RTOL, floor=BFLOOR, what="grid baryonify ellipticity")
    bv, gf = float(g["e_background_val"]), float(g["e_global_tracer_fraction"])
    got = bfg.PaintProfilesAnisGrid(Cat, mk(g["e_map_in"].copy()), eps, paint, tracer, mtot, bv, gf, include_pixel_size=True,
                                    verbose=False).process()
    assert_maps_close(got, g["e_anis"], RTOL, what="anis grid")
    got = bfg.PaintProfilesAnisGrid(Cat, mk(g["e_map_in"].copy()), eps, paint, tracer, mtot, bv, gf, include_pixel_size=False,
                                    use_ellipticity=True, verbose=False).process()
    assert_maps_close(got, g["e_anis_ell"], RTOL, what="anis grid ellipticity")
    G3 = bfg.GriddedMap(map=np.zeros((8, 8, 8)), redshift=zs, bins=bins[:8], cosmo=cosmo)
    Cat3 = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g["e_hM"], zs, cosmo, z=H[:, 0], q_ell=g["e_q"], A_ell=g["e_A"], c_ell=g["e_q"])
    with pytest.raises(ValueError):
        bfg.PaintProfilesGrid(Cat3, G3, eps, paint, use_ellipticity=True, verbose=False).process()


@pytest.mark.parametrize("path", ["direct", "small", "tile"])
@pytest.mark.parametrize("is2D", [True, False])
def test_grid_runners_vs_oracle(cosmo, is2D, path, monkeypatch):
    """larger grids than the golden cases (2D 512^2, 3D 64^3), against the oracle; "tile" = the production tile size"""
    monkeypatch.setenv("BFG_GRID", path)
    import warnings
    rng = np.random.default_rng(31 + int(is2D))
    N, L, nhalo = (512, 400.0, 300) if is2D else (64, 120.0, 120)
    nd = 2 if is2D else 3
    bins = (np.arange(N) + 0.5) * (L / N)
    H = rng.uniform(0, L, (nhalo, 3))
    hM = 10 ** rng.uniform(13.0, 15.2, nhalo)
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, 0.2, cosmo, z=None if is2D else H[:, 2])
    zax, Max, rax, T = syn.pressure_table()
    model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T, T * 0.5)
    Map = bfg.GriddedMap(map=np.zeros((N,) * nd), redshift=0.2, bins=bins, cosmo=cosmo)
    got = bfg.PaintProfilesGrid(Cat, Map, 6, model, verbose=False).process()
    ref = orc.paint_grid(cosmo, bins, (N,) * nd, 0.2, H[:, :nd], hM, (zax, Max, rax), T if is2D else T * 0.5, 6, True)
    assert_maps_close(got, ref, RTOL, what="grid paint vs oracle")
    zax, Max, rax, d = syn.displacement_table()
    dm = bfg.Baryonification2D.from_arrays(zax, Max, rax, d * 5, cosmo, epsilon_max=20)
    m_in = rng.uniform(0, 10, (N,) * nd)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gotb = bfg.BaryonifyGrid(Cat, bfg.GriddedMap(map=m_in.copy(), redshift=0.2, bins=bins, cosmo=cosmo), 6, dm,
                                 verbose=False).process()
        refb = orc.baryonify_grid(cosmo, bins, m_in, 0.2, H[:, :nd], hM, (zax, Max, rax), d * 5, 6, 20)
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what="grid baryonify vs oracle")


@pytest.mark.parametrize("ndim,N", [(2, 150), (2, 333), (3, 40), (3, 53)])
def test_grid_tile_pass_matches_direct_pass(cosmo, ndim, N, monkeypatch):
    """tile-privatised window pass against the one-workgroup-per-halo pass on maps that are not a multiple of the tile
    side, with halos on the box faces (windows that wrap) and windows at the npix / 2 clip: same maps, same counters"""
    from baryonforge_amd.background import Background
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    md = ctx.massdef_struct(Background(cosmo), None)
    rng = np.random.default_rng(70 + N)
    L, nhalo, a = 300.0, 400, 1 / 1.2
    H = rng.uniform(0, L, (nhalo, 3))
    H[:60] = np.where(rng.uniform(size=(60, 3)) < 0.5, rng.uniform(0, 1.0, (60, 3)), L - rng.uniform(0, 1.0, (60, 3)))
    hM = 10 ** rng.uniform(12.5, 15.6, nhalo)                      # the heaviest windows hit the npix / 2 clip
    halos = ctx.to_device(np.stack([hM, np.log(hM), H[:, 0], H[:, 1], H[:, 2]], axis=1))
    bins = ctx.to_device((np.arange(N) + 0.5) * (L / N))
    zax, Max, rax, T = syn.pressure_table()
    ptab = ctx.table([zax, Max, rax], np.log(T), log_values=True)
    zd, Md, rd, d = syn.displacement_table()
    dtab = ctx.table([zd, Md, rd], d, log_values=False)
    pa = ctx.grid_args(ndim, N, bins, halos, a, 8.0, md)
    ba = ctx.grid_args(ndim, N, bins, halos, a, 8.0, md, model_md=md, model_epsilon_max=20.0)
    res = {}
    for path in ("direct", "tile"):
        monkeypatch.setenv("BFG_GRID", path)
        d_map, d_off = ctx.to_device(np.full(N ** ndim, 0.25)), ctx.zeros(N ** ndim, ndim)      # accumulates INTO the map
        ctx.stats_reset()
        ctx.paint_grid(pa, ptab, d_map)
        ctx.baryonify_grid_offsets(ba, dtab, d_off)
        res[path] = (d_map.cpu().numpy(), d_off.cpu().numpy(), ctx.stats())
    np.testing.assert_allclose(res["tile"][0], res["direct"][0], rtol=1e-8, atol=1e-300)
    scale = np.abs(res["direct"][1]).max()
    assert scale > 0
    np.testing.assert_allclose(res["tile"][1], res["direct"][1], rtol=1e-7, atol=1e-9 * scale)
    for key in ("pixel_updates", "pixels_out_of_table", "halos_out_of_table", "warn_mask"):
        assert res["tile"][2][key] == res["direct"][2][key], key


@pytest.mark.parametrize("ndim,N", [(2, 6), (2, 7), (2, 45), (3, 5), (3, 7), (3, 20)])
def test_regrid_grid_kernel_vs_host_rule(ndim, N):
    """bfg_regrid_grid (regrid_pixels_2D/_3D, Map2DRunner.py:14-162) on random offsets of up to several pixels -- exact
    integers, the periodic faces, non-finite entries (zeroed, :591 / :603) -- against the host restatement of the
    reference's candidate-cell scan (pinned by the golden file in tests/test_host_cpu.py).  N < 7: the literal scan
    (cells visited twice through the wrap), N >= 7: the two-cell form."""
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(100 * ndim + N)
    npx = N ** ndim
    off = rng.normal(0, 1.5, (npx, ndim))
    off[rng.uniform(size=npx) < 0.2] = 0.0
    k = rng.integers(0, npx, 30)
    off[k] = np.round(off[k])                                           # exact pixel shifts
    off[rng.integers(0, npx, 10), 0] = np.nan
    off[rng.integers(0, npx, 10), ndim - 1] = np.inf
    val = rng.uniform(0, 3, npx)
    val[rng.uniform(size=npx) < 0.1] = 0.0
    out = ctx.zeros(npx)
    ctx.regrid_grid(ndim, N, ctx.to_device(off), ctx.to_device(val), out)
    got = out.cpu().numpy().reshape((N,) * ndim)
    idx = np.indices((N,) * ndim).reshape(ndim, -1).T.astype(np.float64)       # (i, j[, k]) of every pixel, C order
    g = idx[:, [1, 0] + ([2] if ndim == 3 else [])]                             # x = second index, y = first, z = third
    clean = np.where(np.isfinite(off), off, 0.0)
    ref = np.zeros((N,) * ndim)
    (bfg.regrid_pixels_2D if ndim == 2 else bfg.regrid_pixels_3D)(ref, g + clean, val)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)
    if N >= 7:
        assert np.isclose(got.sum(), val.sum(), rtol=1e-12)


def _snapshot_inputs(g, tag, cosmo):
    is2D = bool(g[f"{tag}_is2D"])
    P, H = g[f"{tag}_P"], g[f"{tag}_H"]
    L, zs = float(g[f"{tag}_L"]), float(g[f"{tag}_redshift"])
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g[f"{tag}_hM"], zs, cosmo, z=None if is2D else H[:, 2])
    Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=None if is2D else P[:, 2], M=np.ones(P.shape[0]), L=L,
                                redshift=zs, cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_d"], cosmo,
                                              epsilon_max=float(g[f"{tag}_eps_model"]),
                                              Rdelta_sampling=bool(g[f"{tag}_rdelta"]))
    return Cat, Part, model, is2D, L


def _periodic_close(got, ref, L, atol):
    d = np.abs(got - ref)
    d = np.minimum(d, L - d)                          # a particle that lands on the box edge may wrap either way
    assert d.max() <= atol, f"max periodic deviation {d.max():.3e} > {atol:.1e}"


@pytest.mark.parametrize("path", ["direct", "cell"])
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_baryonify_snapshot_golden(golden, cosmo, tag, path, monkeypatch):
    """BaryonifySnapshot (SnapshotRunner.py:162-275) against the reference's own run; displacements are ~0.1 Mpc, so
    1e-9 Mpc absolute is 1e-8 relative on the shift (tolerance of the path: 1e-5)"""
    # one thread per particle, a wavefront's hits spread over its lanes (the default) / every lane its own hits / particle indices
    # grouped by cell
    monkeypatch.setenv("BFG_SNAPSHOT", path)
    import warnings
    g = golden("snapshot.npz")
    Cat, Part, model, is2D, L = _snapshot_inputs(g, tag, cosmo)
    R = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=float(g[f"{tag}_eps"]), model=model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new = R.process()
    assert new.dtype == Part.cat.dtype and new.size == Part.cat.size
    got = np.stack([new["x"], new["y"]] + ([] if is2D else [new["z"]]), axis=1)
    _periodic_close(got, g[f"{tag}_P_new"], L, 1e-9)
    assert np.array_equal(new["M"], Part.cat["M"])
    moved_ref = np.any(g[f"{tag}_P_new"] != g[f"{tag}_P"], axis=1)
    moved_got = np.any(got != g[f"{tag}_P"], axis=1)
    assert np.array_equal(moved_ref, moved_got)       # exactly the same particles are displaced


@pytest.mark.parametrize("path", ["direct", "cell"])
@pytest.mark.parametrize("is2D", [False, True])
def test_baryonify_snapshot_vs_oracle(cosmo, is2D, path, monkeypatch):
    """larger box than the golden cases, vs the oracle (KDTree restatement): many cells, halos on the box faces"""
    monkeypatch.setenv("BFG_SNAPSHOT", path)
    import warnings
    rng = np.random.default_rng(123 + int(is2D))
    L, npart, nhalo = 300.0, 400000, 1500
    nd = 2 if is2D else 3
    P = rng.uniform(0, L, (npart, nd))
    H = rng.uniform(0, L, (nhalo, nd))
    H[:20] = np.where(rng.uniform(size=(20, nd)) < 0.5, rng.uniform(0, 0.5, (20, nd)), L - rng.uniform(0, 0.5, (20, nd)))
    hM = 10 ** rng.uniform(13.0, 15.4, nhalo)
    zax, Max, rax, d = syn.displacement_table()
    zs = 0.3
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=None if is2D else H[:, 2])
    Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=None if is2D else P[:, 2], M=np.ones(npart), L=L, redshift=zs,
                                cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    R = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=10, model=model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new = R.process()
        ref = orc.baryonify_snapshot(cosmo, L, zs, P[:, 0], P[:, 1], None if is2D else P[:, 2], hM, H[:, 0], H[:, 1],
                                     None if is2D else H[:, 2], (zax, Max, rax), d, 10, 20)
    got = np.stack([new["x"], new["y"]] + ([] if is2D else [new["z"]]), axis=1)
    _periodic_close(got, ref, L, 1e-9)
    assert R.last_stats["pixel_updates"] > 0
    # the packed (M, x, y, z) float64 catalogue goes to the GPU as it lies in memory (strided entry points); any other
    # layout -- here: an extra column -- has its coordinates gathered on the host: same result
    wide = np.zeros(npart, dtype=Part.cat.dtype.descr + [("id", np.int64)])
    for k in Part.cat.dtype.names:
        wide[k] = Part.cat[k]
    wide["id"] = np.arange(npart)
    Part2 = bfg.ParticleSnapshot.from_catalog(wide, L, zs, cosmo, is2D=is2D)
    assert Part.records() is not None and Part2.records() is None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new2 = bfg.BaryonifySnapshot(Cat, Part2, epsilon_max=10, model=model, verbose=False).process()
    assert new2.dtype == wide.dtype and np.array_equal(new2["id"], wide["id"])
    assert np.array_equal(new2["M"], new["M"])
    got2 = np.stack([new2["x"], new2["y"]] + ([] if is2D else [new2["z"]]), axis=1)
    _periodic_close(got2, got, L, 1e-12)             # the candidate lists are filled in atomic order: sums differ by rounding
    m1 = bfg.ParticleSnapshot.from_catalog(new, L, zs, cosmo, is2D=is2D).make_map(32, mode="cic", device=True)
    m2 = bfg.ParticleSnapshot.from_catalog(new2, L, zs, cosmo, is2D=is2D).make_map(32, mode="cic", device=True)
    np.testing.assert_allclose(m1, m2, rtol=1e-12, atol=1e-12)


def test_snapshot_positional_mass_def_sets_the_query_radius(cosmo):
    """the reference's argument order (SnapshotRunner.py:84-85): a positional mass_def is the mass definition of R_j and
    R_q (:222-225) -- against the oracle run with the same Delta, and different from the 200c result"""
    import warnings
    rng = np.random.default_rng(77)
    L, npart, nhalo, zs = 200.0, 100000, 300, 0.3
    P, H = rng.uniform(0, L, (npart, 3)), rng.uniform(0, L, (nhalo, 3))
    hM = 10 ** rng.uniform(13.5, 15.2, nhalo)
    zax, Max, rax, d = syn.displacement_table()
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=H[:, 2])
    Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2], M=np.ones(npart), L=L, redshift=zs, cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, args in (("500c", (bfg.MassDef(500, "critical"), False)), ("200c", ())):
            new = bfg.BaryonifySnapshot(Cat, Part, 3, model, *args).process() if args else \
                bfg.BaryonifySnapshot(Cat, Part, 3, model, verbose=False).process()
            out[tag] = np.stack([new["x"], new["y"], new["z"]], axis=1)
        ref = orc.baryonify_snapshot(cosmo, L, zs, P[:, 0], P[:, 1], P[:, 2], hM, H[:, 0], H[:, 1], H[:, 2],
                                     (zax, Max, rax), d, 3, 20, Delta=500, rho_type="critical")
    _periodic_close(out["500c"], ref, L, 1e-9)
    moved500 = np.any(out["500c"] != P, axis=1).sum()
    moved200 = np.any(out["200c"] != P, axis=1).sum()
    assert 0 < moved500 < moved200            # R_500c < R_200c: fewer particles inside eps * R


def test_kernels_follow_torchs_current_stream(cosmo):
    """engine.Context binds the library to torch's CURRENT stream at every call (bfg_ctx_set_stream): a paint issued
    inside `with torch.cuda.stream(s)` is ordered after the zero-fill and before the copy that torch enqueues on s, and a
    later call on the default stream is ordered after it (the workspaces are shared)."""
    import torch
    from baryonforge_amd.engine import get_context
    nside, n = 256, 20000
    ra, dec, M, z = syn.catalog(n, seed=9)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False)
    base = R.process()                                            # default stream; also warms every workspace
    ctx = get_context()
    side = torch.cuda.Stream()
    big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
    for _ in range(3):
        with torch.cuda.stream(side):
            big.fill_(1.0)                                        # ~1 GB of stores ahead of the paint on this stream
            d_map = ctx.zeros(12 * nside * nside)
            d_map.fill_(123.0)
            d_map.zero_()                                         # the paint must come after this ...
            R.process_device(d_map)
            got = d_map.clone()                                   # ... and before this
        again = R.process()                                       # default stream, right behind: shares the workspaces
        side.synchronize()
        assert ctx._stream_ptr == torch.cuda.current_stream().cuda_stream
        assert_maps_close(got.cpu().numpy(), ref, RTOL, what="paint on a side stream")
        assert np.array_equal(again != 0, base != 0)
        assert_maps_close(again, ref, RTOL, what="paint on the default stream after a side-stream call")


@pytest.mark.parametrize("path", ["direct", "tile"])
@pytest.mark.parametrize("ndim,N", [(2, 200), (3, 48), (3, 50), (2, 40), (3, 9)])
@pytest.mark.parametrize("mode", ["ngp", "cic"])
def test_deposit_grid_vs_oracle(cosmo, ndim, N, mode, path, monkeypatch):
    """bfg_deposit_grid: NGP = numpy.histogramdd of ParticleSnapshot.make_map (io.py:629-677), incl. particles exactly on
    bin edges and on the box faces; CIC vs the oracle's numpy cloud-in-cell.  Both device paths: direct global atomics
    and the tile-privatised one (grids that are / are not a multiple of the tile size, a grid smaller than one tile)"""
    monkeypatch.setenv("BFG_DEPOSIT", path)
    rng = np.random.default_rng(17 + ndim)
    L, n = 75.0, 200000
    P = rng.uniform(0, L, (n, ndim))
    edges = np.linspace(0, L, N + 1)
    P[:2000] = edges[rng.integers(0, N + 1, (2000, ndim))]          # exactly on edges, 0 and L included
    P[2000:3000] = rng.normal(L / 3, L / 500, (1000, ndim))          # a clump: many particles in one cell
    M = rng.uniform(0.5, 2.0, n)
    S = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2] if ndim == 3 else None, M=M, L=L, redshift=0.1, cosmo=cosmo)
    got = S.make_map(N, mode=mode, device=True)
    ref = orc.make_map(P, M, L, N, mode)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    if mode == "ngp":
        np.testing.assert_allclose(S.make_map(N), ref, rtol=1e-12, atol=1e-12)     # the host path is the reference's
    else:
        assert np.isclose(got.sum(), M.sum(), rtol=1e-12)


@pytest.mark.parametrize("mode", ["ngp", "cic"])
@pytest.mark.parametrize("ndim", [2, 3])
def test_deposit_tile_overflow_list(cosmo, ndim, mode, monkeypatch):
    """tiles denser than their slot count spill into the overflow list (deposited with global atomics): forced with
    3 slots per tile, plus a clump that overflows any realistic slot count; against the oracle"""
    monkeypatch.setenv("BFG_DEPOSIT", "tile")
    monkeypatch.setenv("BFG_DEPOSIT_CAP", "3")
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(400 + ndim)
    L, N, n = 60.0, 40 if ndim == 3 else 150, 120000
    P = rng.uniform(0, L, (n, ndim))
    P[:30000] = rng.normal(L / 2, L / 300, (30000, ndim)) % L          # a clump inside one or two cells
    M = rng.uniform(0.5, 2.0, n)
    got = ctx.deposit_grid(ctx.to_device(P), ctx.to_device(M), L, N, mode).cpu().numpy()
    ref = orc.make_map(P, M, L, N, mode)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    monkeypatch.delenv("BFG_DEPOSIT_CAP")                              # the production slot count, same clump
    got = ctx.deposit_grid(ctx.to_device(P), ctx.to_device(M), L, N, mode).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("mode", ["ngp", "cic"])
def test_deposit_grid_paths_agree_outside_the_box(mode, monkeypatch):
    """positions outside [0, L]: NGP drops them (histogramdd), CIC wraps them; unit masses (d_mass = NULL); the deposit
    accumulates INTO the grid.  Direct and tiled paths must agree to rounding."""
    import torch
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(5)
    L, N, n = 40.0, 70, 300000
    P = rng.uniform(-0.3 * L, 1.3 * L, (n, 3))
    d_pos = ctx.to_device(P)
    out = {}
    for path in ("direct", "tile"):
        monkeypatch.setenv("BFG_DEPOSIT", path)
        g = ctx.deposit_grid(d_pos, None, L, N, mode)
        out[path] = g.cpu().numpy()
    np.testing.assert_allclose(out["tile"], out["direct"], rtol=1e-12, atol=1e-12)
    inside = np.all((P >= 0) & (P <= L), axis=1).sum()
    assert np.isclose(out["tile"].sum(), inside if mode == "ngp" else n, rtol=1e-12)
    np.testing.assert_allclose(out["tile"], orc.make_map(P, None, L, N, mode), rtol=1e-12, atol=1e-12)
    monkeypatch.setenv("BFG_DEPOSIT", "tile")                       # accumulates INTO the grid
    import ctypes as C
    g = ctx.to_device(np.full((N, N, N), 2.5))
    from baryonforge_amd import _lib
    _lib.check(ctx.lib.bfg_deposit_grid(ctx.handle, 3, n, C.c_void_p(d_pos.data_ptr()), None, L, N,
                                        {"ngp": 0, "cic": 1}[mode], C.c_void_p(g.data_ptr())), "bfg_deposit_grid")
    np.testing.assert_allclose(g.cpu().numpy(), out["tile"] + 2.5, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_paint_anis_shell_golden(golden, cosmo, tag):
    """PaintProfilesAnisShell (HealpixRunner.py:486-640) against the reference's own run"""
    g = golden("anis_shell.npz")
    nside = int(g[f"{tag}_nside"])
    axes = (g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"])
    paint = bfg.TabulatedProfile.from_arrays(*axes, g[f"{tag}_T_paint"])
    tracer = bfg.TabulatedProfile.from_arrays(*axes, g[f"{tag}_T_tracer"])
    mtot = bfg.TabulatedProfile.from_arrays(*axes, g[f"{tag}_T_mtot"])
    mtot.proj_cutoff = float(g[f"{tag}_proj_cutoff"])
    Cat = bfg.HaloLightConeCatalog(g[f"{tag}_ra"], g[f"{tag}_dec"], g[f"{tag}_M"], g[f"{tag}_z"], cosmo)
    Shell = bfg.LightconeShell(map=g[f"{tag}_map_in"].copy(), cosmo=cosmo, redshift=float(g[f"{tag}_redshift"]))
    R = bfg.PaintProfilesAnisShell(Cat, Shell, float(g[f"{tag}_eps"]), paint, tracer, mtot,
                                   float(g[f"{tag}_background_val"]), float(g[f"{tag}_global_tracer_fraction"]),
                                   include_pixel_size=bool(g[f"{tag}_ips"]), verbose=False)
    got = R.process()
    ref = g[f"{tag}_map_out"]
    assert got.dtype == np.float64 and got.shape == ref.shape
    assert_maps_close(got, ref, RTOL, what=f"anis {tag}")
    assert np.array_equal(got != 0, ref != 0)


def test_paint_anis_shell_vs_oracle(cosmo):
    """a larger case than the golden one, against the oracle's halo-by-halo restatement"""
    nside = 128
    ra, dec, M, z = syn.catalog(1500, seed=91, z=(0.05, 0.3), logM=(13.0, 15.3))
    zax, Max, rax, T = syn.pressure_table()
    zz, MM, rr = np.meshgrid(np.exp(zax) - 1, np.exp(Max), np.exp(rax), indexing="ij")
    Ttr = 2.0 * (MM / 1e14) ** 0.7 / (1 + (rr / 0.4) ** 2)
    Tm = MM / (1 + (rr / 0.2) ** 2) ** 1.5
    m_in = syn.mass_map(nside)
    ref = orc.paint_anis_shell(cosmo, nside, m_in, 0.15, ra, dec, M, z, (zax, Max, rax), T, Ttr, Tm, 30.0, 1.3, 0.2,
                                  10, include_pixel_size=True)
    mk = lambda t: bfg.TabulatedProfile.from_arrays(zax, Max, rax, t)
    mtot = mk(Tm)
    mtot.proj_cutoff = 30.0
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo, redshift=0.15)
    got = bfg.PaintProfilesAnisShell(Cat, Shell, 10, mk(T), mk(Ttr), mtot, 1.3, 0.2, include_pixel_size=True,
                                     verbose=False).process()
    assert_maps_close(got, ref, RTOL, what="anis vs oracle")


@pytest.mark.parametrize("variant", VARIANTS)
def test_paint_shell_golden_param(golden, cosmo, variant):
    g = golden("paint_shell.npz")
    Cat = bfg.HaloLightConeCatalog(g["p_ra"], g["p_dec"], g["p_M"], g["p_z"], cosmo, cdelta=g["p_cdelta"])
    Shell = bfg.LightconeShell(map=np.zeros(12 * 32 * 32), cosmo=cosmo)
    model = _paint_model(g["p_zax"], g["p_Max"], g["p_rax"], g["p_T2D"], g["p_pax"])
    got = bfg.PaintProfilesShell(Cat, Shell, epsilon_max=float(g["p_eps"]), model=model, verbose=False,
                                 variant=variant).process()
    assert_maps_close(got, g["p_map"], RTOL, what="paint p_keys")
    assert np.array_equal(got != 0, g["p_map"] != 0)


@pytest.mark.parametrize("windows", ["table", "hbm"])
def test_paint_fine_radial_axis_with_extra_dimension(cosmo, windows, monkeypatch):
    """a finely sampled radial axis (1500 nodes) on a table with an extra p_keys dimension: the tile kernel reads the 8
    corner rows of every halo straight from the table (or, BFG_WINDOWS=hbm, pre-blended row windows from HBM) -- against
    the oracle's N-linear read-out"""
    if windows == "hbm":
        monkeypatch.setenv("BFG_WINDOWS", "hbm")
    nside, n, eps = 512, 6000, 10.0
    ra, dec, M, z = syn.catalog(n, seed=314)
    cdelta = np.random.default_rng(5).uniform(0.65, 1.45, n)
    zax, Max, rax, T = syn.pressure_table(3, 12, 1500)
    pax = np.array([0.6, 0.9, 1.2, 1.5])
    T4 = T[..., None] * (1.0 + 0.3 * (pax - 1.0)[None, None, None, :] * np.tanh(np.exp(rax))[None, None, :, None])
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax, pax), T4, nside, eps, extra=cdelta[:, None])
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, cdelta=cdelta)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps,
                               _paint_model(zax, Max, rax, T4, pax), verbose=False)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what=f"fine axis + p_keys ({windows})")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_baryonify_shell_golden(golden, cosmo, tag, variant):
    import warnings
    g = golden("baryonify_shell.npz")
    Cat = bfg.HaloLightConeCatalog(g[f"{tag}_ra"], g[f"{tag}_dec"], g[f"{tag}_M"], g[f"{tag}_z"], cosmo)
    Shell = bfg.LightconeShell(map=g[f"{tag}_map_in"].copy(), cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_d"], cosmo,
                                              epsilon_max=float(g[f"{tag}_eps_model"]),
                                              Rdelta_sampling=bool(g[f"{tag}_rdelta"]))
    R = bfg.BaryonifyShell(Cat, Shell, epsilon_max=float(g[f"{tag}_eps"]), model=model, verbose=False, variant=variant)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = R.process()
    ref = g[f"{tag}_map_out"]
    assert_maps_close(got, ref, RTOL, floor=BFLOOR, what=f"baryonify {tag}")
    assert np.isclose(got.sum(), g[f"{tag}_map_in"].sum())
    assert R.last_stats["halos_fallback4"] > 0   # the fixtures contain discs with < 4 pixels


# --------------------------------------------------------------------------- oracle on seeded synthetic inputs
@pytest.mark.parametrize("variant", VARIANTS)
def test_paint_config0_nside256_1e3(cosmo, variant):
    """BASELINE config[0]: PaintProfilesShell, 1e3 halos, NSIDE = 256, pressure table"""
    ra, dec, M, z = syn.catalog(1000, seed=42)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 256, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 256 * 256), cosmo=cosmo)
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant=variant)
    got = R.process()
    assert_maps_close(got, ref, RTOL, what="config0")
    assert R.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("shape", [(10, 30, 100), (2, 30, 2000)])
def test_paint_nside1024_vs_oracle(cosmo, variant, shape, monkeypatch):
    """BASELINE config[1] geometry (NSIDE = 1024, eps = 10) on a 2e4-halo sample the oracle finishes in seconds;
    default table shape and the notebooks' 2 x 30 x 2000 stress shape; one table with non-finite nodes."""
    ra, dec, M, z = syn.catalog(20000, seed=42)
    zax, Max, rax, T = syn.pressure_table(*shape, bad_block=(shape[2] == 100))
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 1024, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 1024 * 1024), cosmo=cosmo)
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant=variant)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert_maps_close(got, ref, RTOL, what=f"paint 1024 {shape}")
    if variant == "tile_lds" and shape[2] == 2000:
        # the fine axis takes the table-direct read-out; the row-windows-in-HBM form (what tables with extra dimensions
        # and a fine axis still use) must agree with it bit for bit
        monkeypatch.setenv("BFG_WINDOWS", "hbm")
        got2 = R.process()
        assert R.last_stats["pixel_updates"] == ptot
        assert np.array_equal(got2 != 0, got != 0)
        assert_maps_close(got2, got, 1e-13, what="row windows in HBM vs table-direct")


@pytest.mark.parametrize("variant", VARIANTS)
def test_paint_eps20_steep_and_massdef(cosmo, variant):
    ra, dec, M, z = syn.catalog(3000, seed=43, steep=True)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 512, 20, include_pixel_size=True,
                             Delta=500, rho_type="matter")
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 512 * 512), cosmo=cosmo)
    R = bfg.PaintProfilesShell(Cat, Shell, 20, _paint_model(zax, Max, rax, T), mass_def=bfg.MassDef(500, "matter"),
                               include_pixel_size=True, verbose=False, variant=variant)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert_maps_close(got, ref, RTOL, what="eps20 steep")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("rdelta", [False, True])
def test_baryonify_nside256_vs_oracle(cosmo, variant, rdelta):
    import warnings
    ra, dec, M, z = syn.catalog(2000, seed=44)
    zax, Max, rax, d = syn.displacement_table(rdelta=rdelta)
    m_in = syn.mass_map(256)
    m_in[::7] = 0.0
    ref = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, 256, 10, 20, m_in, rdelta=rdelta)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20, Rdelta_sampling=rdelta)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyShell(Cat, Shell, 10, model, verbose=False, variant=variant).process()
    assert_maps_close(got, ref, RTOL, floor=BFLOOR, what="baryonify 256")
    assert not np.allclose(got, m_in)


def test_baryonify_offsets_nside1024_vs_oracle(cosmo):
    """BASELINE config[2] geometry on a 5e3-halo sample: the offset field itself, then mass conservation"""
    import warnings
    ra, dec, M, z = syn.catalog(5000, seed=45)
    zax, Max, rax, d = syn.displacement_table()
    off_ref, ptot = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, 1024, 10, 20, None, offsets_only=True)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=syn.mass_map(1024), cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    R = bfg.BaryonifyShell(Cat, Shell, 10, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        off = R.offsets_device().cpu().numpy()
        assert R.last_stats["pixel_updates"] == ptot
        assert_maps_close(off, off_ref, RTOL, floor=1e-10, what="offsets 1024")
        out = R.process()
    assert np.isclose(out.sum(), Shell.map.sum(), rtol=1e-12)


@pytest.mark.parametrize("nside", [16, 128])
def test_regrid_phi_two_pi_quirk(nside, monkeypatch):
    """healpix_cxx get_interpol at phi == 2 pi exactly: a pixel at phi = 0 of an unshifted ring whose displaced direction
    has y = -1e-22 gets phi = 2 pi after healpy's wrap, tmp = phi / dphi = nr exactly, and its weight-1 deposit goes to
    pixel startpix + nr -- the first pixel of the next ring.  Both regrid kernels must do what the oracle does (found by
    tests/soak/soak.py: the tile-privatised kernel used to wrap that index back into the ring)."""
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    npix = 12 * nside * nside
    rng = np.random.default_rng(nside)
    off = rng.normal(0, 2e-4, (npix, 3))
    off[rng.uniform(size=npix) < 0.5] = 0.0
    ncap, nr = 2 * nside * (nside - 1), 4 * nside
    firsts = ncap + nr * np.arange(1, 2 * nside, 2)              # first pixels (phi = 0) of the unshifted belt rings
    off[firsts] = 0.0
    off[firsts[::2], 1] = -1e-22                                  # nudged to phi = 2 pi
    off[firsts[1::2], 1] = 1e-22                                  # control: stays at phi = +0
    m_in = rng.uniform(1, 10, npix)
    ref = orc.regrid_shell(nside, off, m_in)
    moved = np.abs(ref[firsts[::2]]) < 1e-9                       # the quirk moved these pixels' own mass away
    assert moved.any()
    for path in ("tile", "pixel"):
        if path == "pixel":
            monkeypatch.setenv("BFG_REGRID", "pixel")
        out = ctx.zeros(npix)
        ctx.regrid_shell(nside, ctx.to_device(off), ctx.to_device(m_in), out, None)
        assert_maps_close(out.cpu().numpy(), ref, RTOL, floor=BFLOOR, what=f"regrid {path}")


@pytest.mark.parametrize("nside", [32, 256])
def test_regrid_by_bands_equals_the_whole_regrid(nside):
    """bfg_regrid_shell_bands: the regrid of the source pixels band group by band group (any cut of the 64-ring bands) sums to the
    whole regrid (atomics reorder: rounding), {sum(in), sum(deposits)} accumulate over the calls, and displacements of more than
    4 rings are counted; against the oracle as well"""
    from baryonforge_amd.engine import get_context
    from baryonforge_amd.Runners.HealpixRunner import _regrid_band_groups
    ctx = get_context(0)
    npix = 12 * nside * nside
    rng = np.random.default_rng(nside + 5)
    m_in = rng.uniform(1, 10, npix)
    m_in[rng.uniform(size=npix) < 0.1] = 0.0
    for far in (False, True):
        off = rng.normal(0, 0.3 / nside, (npix, 3))                  # a third of a pixel
        off[rng.uniform(size=npix) < 0.3] = 0.0
        if far:
            off[rng.integers(0, npix, 50)] += rng.normal(0, 12.0 / nside, (50, 3))    # ~10 pixels: beyond the 4-ring halo of a tile
        d_off, d_in = ctx.to_device(off), ctx.to_device(m_in)
        whole, sums = ctx.zeros(npix), ctx.zeros(2)
        ctx.regrid_shell(nside, d_off, d_in, whole, sums)
        ref = orc.regrid_shell(nside, off, m_in)
        assert_maps_close(whole.cpu().numpy(), ref, RTOL, floor=BFLOOR, what="whole regrid")
        for groups in (1, 2, 3, 8):
            cuts_b, cuts_p = _regrid_band_groups(nside, groups)
            assert cuts_p[0] == 0 and cuts_p[-1] == npix and all(a < b for a, b in zip(cuts_p, cuts_p[1:]))
            out, s3 = ctx.zeros(npix), ctx.zeros(3)
            for g in range(len(cuts_b) - 1):
                ctx.regrid_shell_bands(nside, d_off, d_in, out, s3, cuts_b[g], cuts_b[g + 1], clear_sums=(g == 0))
                if g + 1 < len(cuts_b) - 1:
                    # what has been regridded so far deposits nothing beyond 4 rings past its last band -- unless counted as far
                    beyond = out[cuts_p[g + 1] + 5 * 4 * nside:]
                    if not far:
                        assert not beyond.any()
            np.testing.assert_allclose(out.cpu().numpy(), whole.cpu().numpy(), rtol=1e-11, atol=1e-11)
            s3h, sh = s3.cpu().numpy(), sums.cpu().numpy()
            np.testing.assert_allclose(s3h[:2], sh, rtol=1e-12)
            assert (s3h[2] > 0) == far


def test_baryonify_process_with_displacements_beyond_a_band_halo(cosmo, monkeypatch):
    """BaryonifyShell.process() sends the map up, regrids it and sends it back band slice by band slice; a displacement of more
    than 4 rings deposits into slices that may already have left -- counted by the kernel, and the whole map is copied again.
    A displacement table of ~10 pixels at NSIDE 64, against the oracle; with one slice and with eight."""
    import warnings
    nside, n = 64, 400
    ra, dec, M, z = syn.catalog(n, seed=91, logM=(14.0, 15.3), z=(0.05, 0.08))
    zd, Md, rd, d = syn.displacement_table(4, 10, 60)
    zd = np.log(1 + np.geomspace(0.01, 1.0, 4))
    d = d * 1500.0                                                      # tens of Mpc at D_A ~ 250 Mpc: ~0.2 rad = a dozen pixels
    m_in = syn.mass_map(nside)
    ref = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, 10, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    outs = {}
    for slices in ("1", "8"):
        monkeypatch.setenv("BFG_BARY_SLICES", slices)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            outs[slices] = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False).process()
        assert np.isclose(outs[slices].sum(), m_in.sum(), rtol=1e-10)
        assert_maps_close(outs[slices], ref, RTOL, floor=BFLOOR, what=f"baryonify, large displacements, {slices} slice(s)")
    np.testing.assert_allclose(outs["8"], outs["1"], rtol=1e-10, atol=1e-10)
    # the displacements really are larger than a band's halo: the far counter of the band-wise regrid is non-zero
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d_off = R.offsets_device()
    s3 = ctx.zeros(3)
    ctx.regrid_shell_bands(nside, d_off, ctx.to_device(m_in), ctx.zeros(12 * nside * nside), s3, 0, 4, clear_sums=True)
    assert float(s3[2].item()) > 0


# --------------------------------------------------------------------------- full-size properties (no oracle)
def test_paint_full_size_linearity_1e5(cosmo):
    """BASELINE config[1] at full size (1e5 halos, NSIDE 1024): painting is linear in halos
    (HealpixRunner.py:481) -> map(A u B) == map(A) + map(B); variants agree; P_tot adds up."""
    ra, dec, M, z = syn.catalog(100000, seed=42)
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 1024 * 1024), cosmo=cosmo)

    def run(sel, variant="auto"):
        Cat = bfg.HaloLightConeCatalog(ra[sel], dec[sel], M[sel], z[sel], cosmo)
        R = bfg.PaintProfilesShell(Cat, Shell, 10, model, verbose=False, variant=variant)
        return R.process(), R.last_stats["pixel_updates"]
    full, p_full = run(slice(None))
    a, p_a = run(slice(0, 50000))
    b, p_b = run(slice(50000, None))
    assert p_full == p_a + p_b
    assert_maps_close(a + b, full, 1e-10, what="linearity")
    for other in ("scatter_wave", "scatter_quarter"):   # "auto" is the LDS tile variant
        w, p_w = run(slice(None), other)
        assert p_w == p_full
        assert_maps_close(w, full, 1e-8, what=f"variants {other}")
    assert np.all(full >= 0) and np.isfinite(full).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_non_finite_and_unphysical_catalog_rows_are_inert(cosmo, variant):
    """rows healpy's query_disc would find no pixel for (NaN / infinite / negative radius: HealpixRunner.py:329 with a NaN,
    zero or negative mass, z <= -1, non-finite angles) paint nothing, displace nothing and disturb no other halo"""
    import warnings
    nside = 256
    ra, dec, M, z = syn.catalog(4000, seed=404)
    good = np.ones(ra.size, bool)
    bad = {7: ("M", np.nan), 19: ("M", -1e14), 33: ("M", 0.0), 101: ("M", np.inf), 251: ("z", -1.0),
           252: ("z", -3.0), 999: ("ra", np.nan), 1000: ("ra", np.inf), 1500: ("dec", np.nan),
           1501: ("dec", -np.inf), 3999: ("M", np.nan), 0: ("M", np.nan)}
    cols = {"ra": ra, "dec": dec, "M": M, "z": z}
    for i, (c, v) in bad.items():
        cols[c][i] = v
        good[i] = False
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    dz, dM, dr, d = syn.displacement_table()
    dmodel = bfg.Baryonification2D.from_arrays(dz, dM, dr, d, cosmo, epsilon_max=20)
    m_in = syn.mass_map(nside)

    def both(sel):
        Cat = bfg.HaloLightConeCatalog(ra[sel], dec[sel], M[sel], z[sel], cosmo)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10, model,
                                       verbose=False, variant=variant)
            p = R.process()
            B = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, dmodel, verbose=False,
                                   variant=variant)
            return p, R.last_stats["pixel_updates"], B.process()
    p_all, n_all, b_all = both(slice(None))
    p_good, n_good, b_good = both(good)
    assert np.isfinite(p_all).all() and np.isfinite(b_all).all()
    assert n_all >= n_good                      # an infinite mass is a whole-sky disc of zeros: visited, adds nothing
    assert_maps_close(p_all, p_good, 1e-10, what="paint with inert rows")
    assert_maps_close(b_all, b_good, 1e-9, floor=BFLOOR, what="baryonify with inert rows")
    z[5] = np.nan                                # a NaN (or > 30) redshift fails the reference's max(z) assertion (:301, :433)
    with pytest.raises(AssertionError, match="max"):
        both(slice(None))


def test_snapshot_and_deposit_non_finite_rows_are_contained():
    """NaN / infinite particle coordinates and NaN / infinite / negative halo rows (the reference's KDTree refuses such
    input): nothing faults, every clean particle moves exactly as in the clean run, NGP drops what histogramdd drops"""
    import torch
    from baryonforge_amd.background import Background
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(11)
    L, npart, nhalo = 200.0, 200000, 600
    P = rng.uniform(0, L, (npart, 3))
    H = rng.uniform(0, L, (nhalo, 3))
    hM = 10 ** rng.uniform(13.0, 15.0, nhalo)
    zax, Max, rax, d = syn.displacement_table()
    table = ctx.table([zax, Max, rax], d, log_values=False)
    md = ctx.massdef_struct(Background(dict(syn.COSMO)), None)

    def run(P, H, hM):
        with np.errstate(all="ignore"):
            halos = np.stack([hM, np.log(hM), H[:, 0], H[:, 1], H[:, 2]], axis=1)
        d_p, d_h = ctx.to_device(P), ctx.to_device(halos)
        d_out = torch.empty_like(d_p)
        ctx.baryonify_snapshot(d_p, d_h, 3, L, 0.8, 10.0, md, md, 20.0, False, 0, table, d_out)
        grids = [ctx.deposit_grid(d_out, None, L, 64, m).cpu().numpy() for m in ("ngp", "cic")]
        return d_out.cpu().numpy(), grids
    clean, (ngp_clean, cic_clean) = run(P, H, hM)
    P2, H2, M2 = P.copy(), H.copy(), hM.copy()
    badp = np.array([3, 77, 1000, 5000, 123456, npart - 1])
    P2[3, 0] = np.nan; P2[77, 1] = np.inf; P2[1000, 2] = -np.inf; P2[5000] = np.nan; P2[123456, 0] = 1e300; P2[npart - 1, 2] = -1e300
    extra_h = np.array([[np.nan, 50, 50], [np.inf, 10, 10], [20, -np.inf, 20], [30, 30, 30], [40, 40, 40], [60, 60, 60], [70, 70, 70]])
    extra_m = np.array([1e14, 1e14, 1e14, np.nan, -1e14, 0.0, np.inf])
    H2, M2 = np.concatenate([H2, extra_h]), np.concatenate([M2, extra_m])
    got, (ngp, cic) = run(P2, H2, M2)
    ok = np.ones(npart, bool); ok[badp] = False
    np.testing.assert_allclose(got[ok], clean[ok], rtol=1e-12, atol=1e-9)
    # NGP: only in-box, finite particles are counted (np.histogramdd semantics)
    inbox = np.all((got >= 0) & (got <= L), axis=1)
    assert ngp.sum() == inbox.sum() and np.isfinite(ngp).all()
    assert ngp_clean.sum() == npart and np.isclose(cic_clean.sum(), npart, rtol=1e-12)


def test_workspace_regrowth_and_call_order(cosmo):
    """One context serves shells of very different sizes in any order (the per-halo, tile-list and row-window
    workspaces grow, are reused while larger than needed, and serve paint and baryonify alternately): every catalog
    gives the same map whatever ran before it, and linearity ties the sizes together."""
    ra, dec, M, z = syn.catalog(1500000, seed=77)
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    dz, dM, dr, d = syn.displacement_table()
    dmodel = bfg.Baryonification2D.from_arrays(dz, dM, dr, d, cosmo, epsilon_max=20)
    nside = 512
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    m_in = syn.mass_map(nside)

    def paint(n0, n1):
        Cat = bfg.HaloLightConeCatalog(ra[n0:n1], dec[n0:n1], M[n0:n1], z[n0:n1], cosmo)
        R = bfg.PaintProfilesShell(Cat, Shell, 10, model, verbose=False)
        return R.process(), R.last_stats["pixel_updates"]

    def baryonify(n0, n1):
        import warnings
        Cat = bfg.HaloLightConeCatalog(ra[n0:n1], dec[n0:n1], M[n0:n1], z[n0:n1], cosmo)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, dmodel, verbose=False).process()

    cuts = [(0, 1000000), (1000000, 1000007), (0, 1500000), (1000007, 1100000), (5, 5), (1100000, 1500000)]
    first = {c: paint(*c) for c in cuts}
    b_first = baryonify(0, 20000)
    second = {c: paint(*c) for c in reversed(cuts)}               # shrinking, growing, empty: the other way round
    b_second = baryonify(0, 20000)
    for c in cuts:
        assert first[c][1] == second[c][1]
        assert_maps_close(second[c][0], first[c][0], 1e-10, what=f"order independence {c}")
    assert first[(5, 5)][1] == 0 and not first[(5, 5)][0].any()
    parts = [(0, 1000000), (1000000, 1000007), (1000007, 1100000), (1100000, 1500000)]
    assert sum(first[c][1] for c in parts) == first[(0, 1500000)][1]
    assert_maps_close(sum(first[c][0] for c in parts), first[(0, 1500000)][0], 1e-10, what="linearity across sizes")
    assert_maps_close(b_second, b_first, 1e-9, floor=BFLOOR, what="baryonify between paints")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("nside", [8, 24, 37, 100])
def test_paint_and_baryonify_any_nside(cosmo, nside, variant):
    """RING-scheme NSIDE need not be a power of two (healpy accepts any NSIDE for RING maps)"""
    import warnings
    ra, dec, M, z = syn.catalog(300, seed=nside, z=(0.03, 0.2), logM=(13.5, 15.5))
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant=variant)
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert_maps_close(got, ref, RTOL, what=f"paint nside {nside}")
    zax, Max, rax, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    refb = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, nside, 10, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False,
                                  variant=variant).process()
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what=f"baryonify nside {nside}")


@pytest.mark.parametrize("switch", ["BFG_TILE_LIGHT=1", "BFG_TILE_LIGHT=0", "BFG_TILE_PERSIST=0", "BFG_TILE_PERSIST=7",
                                    "BFG_FINAL_DRAIN=inline", "BFG_FINAL_DRAIN=kernel", "BFG_TILE_SCAN=1", "BFG_OUT_OVERWRITE=0",
                                    "BFG_ROWS=separate", "BFG_BLEND=0", "BFG_BLEND=0 BFG_ROWS=separate", "BFG_EAGER_SOA=1",
                                    "BFG_ITEM_COUNTERS=1", "BFG_ITEM_COUNTERS=3", "BFG_ITEM_COUNTERS=16 BFG_TILE_PERSIST=40"])
def test_tile_kernel_switches_agree(cosmo, switch, monkeypatch):
    """every A/B switch of the tile path paints the default build's map: the wave-private-chunk kernel, the
    256-thread instantiation, one workgroup per item / a tiny persistent grid, the deferred pixels drained by every item / added by
    the follow-up kernel instead of by the workgroups after their last item, the scan-built work
    list, a cleared instead of an overwritten output, row windows from halo_row4_kernel instead of the prep kernel, one / three /
    sixteen item counters (the last with a grid they do not divide) -- paint and
    baryonify, against the default path and the oracle"""
    import warnings
    nside, n, eps = 512, 20000, 10.0
    ra, dec, M, z = syn.catalog(n, seed=91)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, eps)
    zd, Md, rd, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, m_in)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)

    def run():
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps,
                                   _paint_model(zax, Max, rax, T), verbose=False)
        got = R.process()
        assert R.last_stats["pixel_updates"] == ptot
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, bm, verbose=False).process()
        return got, gotb
    base, baseb = run()
    for kv in switch.split():
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    got, gotb = run()
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what=f"paint with {switch}")
    assert_maps_close(got, base, 1e-12, what=f"paint with {switch} vs default")
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what=f"baryonify with {switch}")
    assert_maps_close(gotb, baseb, 1e-8, floor=BFLOOR, what=f"baryonify with {switch} vs default")


def test_tile_deferred_pixel_queue_overflow(cosmo, monkeypatch):
    """pixels below a halo's staged row window are queued in LDS and drained in batches; with a full queue they are
    painted inline (forced here with debug bit 32: a 4-entry queue)"""
    nside = 512
    ra, dec, M, z = syn.catalog(2000, seed=77, z=(0.02, 0.1), logM=(14.5, 15.5))     # large discs: many inner pixels
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    outs = []
    for dbg in ("0", "32"):
        monkeypatch.setenv("BFG_DEBUG", dbg)
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                                   _paint_model(zax, Max, rax, T), verbose=False, variant="tile_lds")
        outs.append(R.process())
        assert R.last_stats["pixel_updates"] == ptot
        assert_maps_close(outs[-1], ref, RTOL, what=f"paint deferred queue debug={dbg}")


@pytest.mark.parametrize("nside", [8, 16])
def test_tile_wrapped_window_spill_path(cosmo, nside, monkeypatch):
    """ring windows that wrap around inside a sector produce a second segment; when a chunk runs out of segment
    records the piece is painted through the direct read-out (forced here with debug bit 16)"""
    import warnings
    monkeypatch.setenv("BFG_DEBUG", "16")
    ra, dec, M, z = syn.catalog(400, seed=5 + nside, z=(0.03, 0.2), logM=(13.5, 15.5))
    ra[:200] = (ra[:200] * 0.02 - 3.0) % 360.0                      # many discs across phi = 0
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False, variant="tile_lds")
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert_maps_close(got, ref, RTOL, what=f"paint spill nside {nside}")
    zax, Max, rax, d = syn.displacement_table()
    m_in = syn.mass_map(nside)
    refb = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, nside, 10, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False,
                                  variant="tile_lds").process()
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what=f"baryonify spill nside {nside}")


def test_tile_pair_buffer_overflow_falls_back_to_scatter(cosmo, monkeypatch):
    """If the (halo, tile) pair buffer is too small the whole call degrades to the scatter kernel -- same map."""
    ra, dec, M, z = syn.catalog(3000, seed=46)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 256, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 256 * 256), cosmo=cosmo)
    monkeypatch.setenv("BFG_TILE_CAP", "2")                 # two fixed slots per tile: nearly every pair overflows ...
    monkeypatch.setenv("BFG_PAIR_CAP", "100")               # ... into lists that cannot hold them
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant="tile_lds")
    with pytest.warns(UserWarning, match="scatter kernel"):  # the degradation is reported, not silent
        got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert R.last_stats["fallback_halos"] == 3000             # bfg_stats.halos_scatter_fallback: every halo
    assert_maps_close(got, ref, RTOL, what="pair overflow fallback")
    monkeypatch.delenv("BFG_PAIR_CAP")
    monkeypatch.delenv("BFG_TILE_CAP")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                        # ... and the normal path neither counts nor warns
        got = R.process()
    assert R.last_stats["fallback_halos"] == 0
    assert_maps_close(got, ref, RTOL, what="after fallback")
    # a disc that overlaps more than 64 sky tiles is left to the scatter kernel on its own: counted as one
    big = bfg.HaloLightConeCatalog(np.append(ra[:50], 10.0), np.append(dec[:50], 5.0), np.append(M[:50], 5e15),
                                   np.append(z[:50], 0.012), cosmo)
    zb = np.append(z[:50], 0.012)
    refb, ptb = oracle_paint(cosmo, np.append(ra[:50], 10.0), np.append(dec[:50], 5.0), np.append(M[:50], 5e15), zb,
                             (zax, Max, rax), T, 256, 10)
    Rb = bfg.PaintProfilesShell(big, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant="tile_lds")
    with warnings.catch_warnings(record=True) as wrec:
        warnings.simplefilter("always")
        gotb = Rb.process()
    assert Rb.last_stats["pixel_updates"] == ptb
    assert_maps_close(gotb, refb, RTOL, what="one huge disc")
    if Rb.last_stats["fallback_halos"]:
        assert Rb.last_stats["fallback_halos"] == 1 and any("scatter kernel" in str(w.message) for w in wrec)


@pytest.mark.parametrize("slots", [1, 3, 17])
def test_tile_slots_and_overflow_lists(cosmo, slots, monkeypatch):
    """halo -> tile binning: the count pass drops a pair into one of the tile's fixed slots, pairs that find them full are
    appended to the tile's overflow list by the fill pass; with 1 / 3 / 17 slots per tile both kinds of work item occur
    for most tiles (and write to the same map tile with atomics) -- paint and baryonify against the oracle"""
    import warnings
    monkeypatch.setenv("BFG_TILE_CAP", str(slots))
    ra, dec, M, z = syn.catalog(6000, seed=146)
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 256, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * 256 * 256), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False, variant="tile_lds")
    got = R.process()
    assert R.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what=f"{slots} slots per tile")
    zd, Md, rd, d = syn.displacement_table()
    m_in = syn.mass_map(256)
    refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, 256, 10, 20, m_in)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, model, verbose=False).process()
    assert_maps_close(gotb, refb, RTOL, floor=BFLOOR, what=f"baryonify, {slots} slots per tile")


# --------------------------------------------------------------------------- edge cases and errors
def test_edge_cases(cosmo):
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    nside = 64
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    # empty catalog
    e = np.array([])
    out = bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(e, e, e, e, cosmo), Shell, 10, model, verbose=False).process()
    assert out.shape == (12 * nside * nside,) and not out.any()
    # a halo whose (z, M) is outside the table paints nothing; poles; phi wrap; huge disc (pole inside)
    ra = np.array([10.0, 0.0, 359.999, 45.0, 200.0])
    dec = np.array([5.0, 90.0, -90.0, 89.9, -89.95])
    M = np.array([1e17, 1e15, 1e15, 5e15, 5e15])
    z = np.array([0.3, 0.05, 0.05, 0.02, 0.02])
    with pytest.warns(UserWarning):
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    for variant in VARIANTS:
        R = bfg.PaintProfilesShell(Cat, Shell, 10, model, verbose=False, variant=variant)
        got = R.process()
        ref, ptot = oracle_paint(cosmo, Cat.cat["ra"], Cat.cat["dec"], M, z, (zax, Max, rax), T, nside, 10)
        assert R.last_stats["pixel_updates"] == ptot
        assert R.last_stats["halos_out_of_table"] == 1
        assert_maps_close(got, ref, RTOL, what="edge")


def test_error_conventions(cosmo):
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 16 * 16), cosmo=cosmo)
    one = np.array([1.0])
    Cat = bfg.HaloLightConeCatalog(one, one, one * 1e14, one * 0.3, cosmo)
    with pytest.raises(NotImplementedError):
        bfg.PaintProfilesShell(Cat, Shell, 10, model, use_ellipticity=True)
    with pytest.raises(AssertionError):
        bfg.PaintProfilesShell(Cat, Shell, 10, None, verbose=False).process()
    with pytest.raises(AssertionError):   # z > 30
        bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(one, one, one * 1e14, one * 31.0, cosmo), Shell, 10, model,
                               verbose=False).process()
    with pytest.raises(NameError):        # table never built
        bfg.PaintProfilesShell(Cat, Shell, 10, bfg.TabulatedProfile(object(), cosmo), verbose=False).process()
    with pytest.raises(TypeError):
        bfg.PaintProfilesShell(Cat, Shell, 10, object(), verbose=False).process()
    # zero map early return hands back the input object (HealpixRunner.py:293-294)
    zd = syn.displacement_table()
    disp = bfg.Baryonification2D.from_arrays(*zd, cosmo)
    zero = np.zeros(12 * 16 * 16)
    assert bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=zero, cosmo=cosmo), 10, disp, verbose=False).process() is zero
    # p_keys with a model of the wrong type
    class Fake(object):
        pass
    f = Fake()
    f.p_keys = ["cdelta"]
    with pytest.raises(AssertionError):
        bfg.PaintProfilesShell(Cat, Shell, 10, f, verbose=False).process()


# ------------------------------------------------------------------ displacement-table builder on the device (a6)
class _Profile(object):
    """analytic density stand-in for the (out-of-scope) profile zoo; `ring` adds an oscillating, partly NEGATIVE tail
    (what FFTLog ringing does to the reference's projected profiles) and `hole` zeroes a band of radii"""

    def __init__(self, core, slope, ring=0.0, hole=None, cdelta=1.0):
        self.core, self.slope, self.ring, self.hole, self.cdelta, self.cutoff = core, slope, ring, hole, cdelta, None

    def set_parameter(self, k, v):
        setattr(self, k, v)

    def projected(self, cosmo, r, M, a):
        M = np.atleast_1d(M)
        r = np.atleast_1d(r)
        R = (orc.get_radius(syn.COSMO, M, a) / a)[:, None]
        x = r[None, :] / (self.core * self.cdelta * R)
        S = M[:, None] / (2 * np.pi * (self.core * R) ** 2) * (1 + x * x) ** (-self.slope)
        S = S * np.exp(-r[None, :] / (30 * R))
        if self.ring:
            S = S * (1 + self.ring * np.sin(6 * np.log(r))[None, :] * (r[None, :] / R) ** 1.5)
        if self.hole is not None:
            S = np.where((r[None, :] > self.hole[0]) & (r[None, :] < self.hole[1]), 0.0, S)
        return S

    real = projected


def test_table_builder_device_vs_reference_golden(golden):
    """bfg_build_displacement_table on the Sigma arrays of the golden file against the table the reference's own
    Baryonification2D.setup_interpolator built from them (BaryonCorrection.py:142-328)"""
    from baryonforge_amd.engine import get_context
    g = golden("table_builder.npz")
    Sd, Sb = g["tb_Sigma_DMO_z"], g["tb_Sigma_DMB_z"]                 # [Nz][NM][N_int]
    nz, nm, nint = Sd.shape
    d, status = get_context(0).build_displacement_table(2, g["tb_rint"], Sd.reshape(nz * nm, nint),
                                                        Sb.reshape(nz * nm, nint), g["tb_r"])
    assert np.all(status == 0)
    ref = g["tb_d_interp"]
    np.testing.assert_allclose(d.reshape(ref.shape), ref, rtol=1e-9, atol=1e-12)
    assert np.abs(ref).max() > 1e-3                                       # a non-trivial table


@pytest.mark.parametrize("case", ["smooth2d", "ringing2d", "hole3d", "rdelta", "params", "flat"])
def test_table_builder_device_vs_host(case):
    """setup_interpolator(device=True) against the host builder (the reference's scipy calls, pinned by the golden
    file in tests/test_host_cpu.py) on profiles that exercise the masks: negative / zero densities, nearly flat mass
    profiles (iterative mask, warnings, d = 0 rows), Rdelta_sampling, an extra table dimension, the 3D variant"""
    import warnings
    kw = dict(z_min=0.1, z_max=0.6, N_samples_z=3, M_min=1e12, M_max=1e16, N_samples_Mass=6, R_min=1e-3, R_max=1e2,
              N_samples_R=60, verbose=False)
    cls = bfg.Baryonification2D
    if case == "smooth2d":
        mk = lambda: (_Profile(0.25, 1.6), _Profile(0.45, 1.6))
    elif case == "ringing2d":
        mk = lambda: (_Profile(0.25, 1.6, ring=0.9), _Profile(0.45, 1.5, ring=1.4))
    elif case == "hole3d":
        cls = bfg.Baryonification3D
        mk = lambda: (_Profile(0.25, 1.9, hole=(0.02, 0.05)), _Profile(0.4, 1.9, hole=(0.3, 0.9)))
    elif case == "rdelta":
        mk = lambda: (_Profile(0.25, 1.6), _Profile(0.45, 1.6, ring=0.5))
        kw.update(Rdelta_sampling=True, Rdelta_min=1e-2, Rdelta_max=20)
    elif case == "params":
        mk = lambda: (_Profile(0.25, 1.6), _Profile(0.45, 1.6))
        kw.update(other_params={"cdelta": np.array([0.7, 1.0, 1.6])})
    else:                                                                 # DMB == DMO almost everywhere: rows default to 0
        mk = lambda: (_Profile(0.25, 1.6), _Profile(0.25, 1.6, hole=(50.0, 60.0)))
    out, warned = {}, {}
    for dev in (False, True):
        DMO, DMB = mk()
        B = cls(DMO, DMB, dict(syn.COSMO), epsilon_max=20, N_int=400)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            B.setup_interpolator(device=dev, **kw)
        out[dev] = B.raw_input_d
        warned[dev] = sorted(str(x.message)[:70] for x in w if issubclass(x.category, UserWarning))
        assert B.Rdelta_sampling == bool(kw.get("Rdelta_sampling", False))
    assert out[True].shape == out[False].shape
    scale = np.abs(out[False]).max()
    np.testing.assert_allclose(out[True], out[False], rtol=1e-8, atol=1e-11 * max(scale, 1.0))
    assert warned[True] == warned[False]
    if case == "flat":
        assert len(warned[True]) > 0 and np.count_nonzero(out[True]) < out[True].size
    else:
        assert scale > 1e-4


def test_tabulated_correlation3d_matches_scipy():
    """TabulatedCorrelation3D (Tabulate.py:733-784): exp of the bilinear ln xi table, NaN outside the table and wherever
    a node of the cell is non-positive -- against the scipy RegularGridInterpolator the reference builds"""
    from scipy import interpolate
    xi_of = lambda cosmo, a, r: a ** 2 * (r / 5.0) ** -1.8 * (1 + 0.05 * np.sin(r / 15.0)) - 2e-4     # < 0 at large r
    X = bfg.utils.TabulatedCorrelation3D(dict(syn.COSMO), R_range=[1e-2, 3e2], N_samples=120, correlation_3d=xi_of)
    with pytest.raises(NameError):
        X(1.0, 0.8)
    X.setup_interpolator(z_min=0, z_max=2, N_samples_z=7)
    assert X.raw_input_3D.shape == (7, 120)
    with np.errstate(all="ignore"):
        rgi = interpolate.RegularGridInterpolator((X.raw_input_z_range, X.raw_input_r_range), np.log(X.raw_input_3D),
                                                  bounds_error=False)
        rng = np.random.default_rng(4)
        r = np.concatenate([10 ** rng.uniform(-2.3, 2.7, 4000), np.exp(X.raw_input_r_range[[0, 5, -1]])])
        for a in (1.0, 0.71, 1 / 3.0, 0.3):                                  # the last one lies outside the z range
            got = X(r, a)
            ref = np.exp(rgi((np.log(1 / a) * np.ones_like(r), np.log(r))))
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            ok = ~np.isnan(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12)
        assert np.isnan(X(r, 0.3)).all() and np.isnan(got).any()
    Y = bfg.utils.TabulatedCorrelation3D.from_arrays(X.raw_input_z_range, X.raw_input_r_range, X.raw_input_3D)
    np.testing.assert_array_equal(Y(r[:50], 0.9), X(r[:50], 0.9))


def test_snapshot_and_grid_edge_cases(cosmo):
    """empty halo catalogue / empty particle set, particles sitting exactly on a halo centre (NaN, as in the reference),
    a halo whose mass lies outside the table (contributes nothing, warns),
    a halo with a NaN mass; grid paint with no halos"""
    import warnings
    L, zs = 80.0, 0.3
    rng = np.random.default_rng(9)
    P = rng.uniform(0, L, (5000, 3))
    H = rng.uniform(0, L, (6, 3)).astype(np.float32).astype(np.float64)
    hM = np.array([3e13, 8e14, 2e14, 1e18, np.nan, 5e13])
    P[:3] = H[:3]                                                          # particles on halo centres
    zax, Max, rax, d = syn.displacement_table()
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2], M=np.ones(len(P)), L=L, redshift=zs, cosmo=cosmo)
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], hM, zs, cosmo, z=H[:, 2])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        new = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=10, model=model, verbose=False).process()
        c = Cat.cat
        ref = orc.baryonify_snapshot(cosmo, L, zs, P[:, 0], P[:, 1], P[:, 2], c["M"], c["x"], c["y"], c["z"], (zax, Max, rax),
                                     d, 10, 20)
    got = np.stack([new["x"], new["y"], new["z"]], axis=1)
    # 0 / 0 in the reference's unit vector: the three particles on halo centres come back NaN (SnapshotRunner.py:228-232)
    assert np.isnan(ref[:3]).all() and np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref).any(axis=1)
    _periodic_close(got[ok], ref[ok], L, 1e-9)
    assert any("outside table" in str(x.message) for x in w)               # the 1e18 Msun halo
    # no halos: the particles come back unchanged; no particles: an empty catalogue
    Cat0 = bfg.HaloNDCatalog(H[:0, 0], H[:0, 1], hM[:0], zs, cosmo, z=H[:0, 2])
    new0 = bfg.BaryonifySnapshot(Cat0, Part, epsilon_max=10, model=model, verbose=False).process()
    for k, col in zip(("x", "y", "z"), range(3)):
        assert np.array_equal(new0[k], P[:, col])
    Part0 = bfg.ParticleSnapshot(x=P[:0, 0], y=P[:0, 1], z=P[:0, 2], M=np.ones(0), L=L, redshift=zs, cosmo=cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert bfg.BaryonifySnapshot(Cat, Part0, epsilon_max=10, model=model, verbose=False).process().size == 0
    # grid paint without halos: the zero map
    N = 32
    bins = (np.arange(N) + 0.5) * (L / N)
    zp, Mp, rp, T = syn.pressure_table()
    paint = bfg.TabulatedProfile.from_arrays(zp, Mp, rp, T, T)
    out = bfg.PaintProfilesGrid(Cat0, bfg.GriddedMap(map=np.zeros((N, N, N)), redshift=zs, bins=bins, cosmo=cosmo), 5, paint,
                                verbose=False).process()
    assert out.shape == (N, N, N) and not out.any()


def test_call_sequences_on_one_context_keep_the_tile_counters_clean(cosmo, monkeypatch):
    """the tile counters come in two sets, each call's scan kernel clearing the set the next call counts in (no memset launch):
    sequences of calls that differ in catalog size, resolution, workload and work-list path (scan-free / scan kernel) on the
    process's one context must paint what each call paints on its own"""
    import warnings
    zax, Max, rax, T = syn.pressure_table()
    zd, Md, rd, d = syn.displacement_table()
    bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    cases = [(256, 30000, 3), (128, 40, 4), (256, 30000, 3), (512, 3000, 5), (128, 20000, 6), (128, 40, 4)]

    def paint(nside, n, seed):
        ra, dec, M, z = syn.catalog(n, seed=seed)
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                                   _paint_model(zax, Max, rax, T), verbose=False)
        out = R.process()
        return out, R.last_stats["pixel_updates"]

    def bary(nside, n, seed):
        ra, dec, M, z = syn.catalog(n, seed=seed)
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=syn.mass_map(nside), cosmo=cosmo), 10, bm, verbose=False).process()

    first = {}
    for rnd in range(2):
        for k, (nside, n, seed) in enumerate(cases):
            if k == 4:
                monkeypatch.setenv("BFG_TILE_SCAN", "1")            # this call builds its work list with the scan kernel
            got, ptot = paint(nside, n, seed)
            gb = bary(nside, n, seed) if k % 2 == 0 else None
            monkeypatch.delenv("BFG_TILE_SCAN", raising=False)
            key = (nside, n, seed)
            if key not in first:
                ref, pref = oracle_paint(cosmo, *syn.catalog(n, seed=seed), (zax, Max, rax), T, nside, 10)
                assert ptot == pref
                assert_maps_close(got, ref, RTOL, what=f"call {k} of round {rnd}")
                first[key] = (got, ptot, gb)
            else:
                assert ptot == first[key][1]
                assert_maps_close(got, first[key][0], 1e-12, what=f"repeat of {key}")
                if gb is not None and first[key][2] is not None:
                    assert_maps_close(gb, first[key][2], 1e-8, floor=BFLOOR, what=f"baryonify repeat of {key}")


def test_timing_select_times_only_the_chosen_kernel_classes(cosmo):
    """bfg_timing_select: events for the chosen kernel classes only (bench.py times the dominant kernel in its timed region)"""
    from baryonforge_amd.background import Background
    from baryonforge_amd.engine import get_context
    ctx = get_context(0)
    nside, n = 256, 20000
    ra, dec, M, z = syn.catalog(n, seed=12)
    zax, Max, rax, T = syn.pressure_table()
    bg = Background(cosmo)
    d_cat = ctx.to_device(np.stack([M, z, ra, dec], axis=1))
    table = ctx.table([zax, Max, rax], np.log(T), log_values=True)
    spline = ctx.da_spline(bg, float(z.max()))
    d_map = ctx.zeros(12 * nside * nside)
    sargs = ctx.shell_args(nside, d_cat, n, 4, 0, 10.0, ctx.massdef_struct(bg, None), out_overwrite=True)
    ctx.timing_enable(True, which=[1])
    for _ in range(3):
        ctx.paint_shell(sargs, table, spline, d_map)
    assert ctx.timing_read(1)[1] == 3 and ctx.timing_read(1)[0] > 0
    assert all(ctx.timing_read(k)[1] == 0 for k in (0, 2, 3, 4, 5))
    ctx.timing_enable(True)                                           # all classes again
    ctx.paint_shell(sargs, table, spline, d_map)
    assert ctx.timing_read(0)[1] == 1 and ctx.timing_read(1)[1] == 1 and ctx.timing_read(3)[1] == 1
    ctx.timing_enable(False)


# --------------------------------------------------------------------------- sliced calls (the multi-GPU join inside one call)
def _sliced_paint(runner, slices, poison=True):
    """process_device through bfg_paint_shell_sliced; every slice is copied out the moment it is reported (stream order), and
    the slice is then poisoned: a later launch that still wrote into a reported slice would be caught"""
    import torch
    ctx = bfg.engine.get_context()
    npix = 12 * runner.LightconeShell.NSIDE ** 2
    d_map = ctx.empty(npix)
    got = torch.empty(npix, dtype=torch.float64, device=d_map.device)
    seen = []

    def on_slice(k, n, lo, hi):
        seen.append((k, n, lo, hi))
        got[lo:hi] = d_map[lo:hi]                  # stream-ordered: what the exchange of this slice would read
        if poison:
            d_map[lo:hi] = float("nan")
    runner.process_device(d_map=d_map, overwrite=True, slices=slices, on_slice=on_slice)
    return got.cpu().numpy(), seen


@pytest.mark.parametrize("case", ["plain", "scan", "leftover", "degraded", "scatter_variant"])
def test_sliced_paint_equals_the_plain_call(cosmo, case, monkeypatch):
    """bfg_paint_shell_sliced: the slices cover the map exactly once in ascending order, each is final when it is reported,
    and their union equals the plain call's map (to the rounding of reordered atomic adds) -- on the scan-free path, with the scan-built work list
    (shared tiles), with a left-over halo (the scatter kernel then runs first and the tiles are added), with the pair
    buffer exhausted (everything through the scatter kernel) and for a scatter variant (one slice)."""
    import warnings
    nside, n = 256, 5000
    ra, dec, M, z = syn.catalog(n, seed=77)
    if case == "leftover":                                    # one disc over more than 64 sky tiles
        ra, dec, M, z = np.append(ra, 10.0), np.append(dec, 5.0), np.append(M, 5e15), np.append(z, 0.012)
    zax, Max, rax, T = syn.pressure_table()
    if case == "scan":
        monkeypatch.setenv("BFG_TILE_CAP", "3")               # overflow lists + tiles cut into several (shared) items
        monkeypatch.setenv("BFG_TILE_SCAN", "1")
    if case == "degraded":
        monkeypatch.setenv("BFG_TILE_CAP", "2")
        monkeypatch.setenv("BFG_PAIR_CAP", "100")
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    variant = "scatter_quarter" if case == "scatter_variant" else "tile_lds"
    R = bfg.PaintProfilesShell(Cat, Shell, 10, _paint_model(zax, Max, rax, T), verbose=False, variant=variant)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = R.process()
        assert R.last_stats["pixel_updates"] == ptot
        # the reference ranges: an EMPTY catalog on the same shell (what a rank whose sky-patch shard holds no halo reports)
        ranges = {}
        R0 = bfg.PaintProfilesShell(Cat[np.arange(0)], Shell, 10, _paint_model(zax, Max, rax, T), verbose=False)
        for slices in (1, 3, 7, 64):
            got0, ranges[slices] = _sliced_paint(R0, slices)
            assert not got0.any()
        for slices in (1, 3, 7, 64):
            got, seen = _sliced_paint(R, slices)
            assert R.last_stats["pixel_updates"] == ptot
            nrep = seen[0][1]
            assert [s[0] for s in seen] == list(range(nrep)) and all(s[1] == nrep for s in seen)
            assert seen[0][2] == 0 and seen[-1][3] == got.size
            assert all(a[3] == b[2] and a[2] < a[3] for a, b in zip(seen, seen[1:]))      # contiguous, ascending, non-empty
            # the number of slices and their ranges are a function of (nside, loop, slices) only: a scatter variant -- or a rank
            # with an empty shard, below -- reports the ranges its peers report (one collective per callback on every rank)
            assert nrep == min(slices, 16, (4 * nside - 1 + 31) // 32)
            ranges.setdefault(slices, seen)
            assert [s[2:] for s in seen] == [s[2:] for s in ranges[slices]]
            # equal up to the order in which atomics add (deferred pixels, shared tiles, the scatter kernel): rounding only
            assert np.array_equal(got != 0, plain != 0)
            np.testing.assert_allclose(got, plain, rtol=1e-12, atol=0, err_msg=f"{case}: {slices} slices vs the plain call")
    assert_maps_close(plain, ref, RTOL, what=f"sliced paint ({case})")


@pytest.mark.parametrize("case", ["plain", "leftover", "degraded"])
def test_sliced_paint_accumulates_into_a_given_map(cosmo, case, monkeypatch):
    """process_device(d_map=existing, slices > 1) without overwrite adds INTO the map -- also when left-over halos exist or the
    binning degrades (the sliced call used to clear the output in exactly those cases)"""
    import warnings
    nside, n = 128, 1500
    ra, dec, M, z = syn.catalog(n, seed=79)
    if case == "leftover":
        ra, dec, M, z = np.append(ra, 10.0), np.append(dec, 5.0), np.append(M, 5e15), np.append(z, 0.012)
    if case == "degraded":
        monkeypatch.setenv("BFG_TILE_CAP", "2")
        monkeypatch.setenv("BFG_PAIR_CAP", "100")
    zax, Max, rax, T = syn.pressure_table()
    ctx = bfg.engine.get_context()
    R = bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo),
                               bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False)
    before = np.random.default_rng(3).uniform(1.0, 2.0, 12 * nside * nside)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = R.process()
        d_map = ctx.to_device(before)
        R.process_device(d_map=d_map, slices=4, on_slice=lambda k, n, lo, hi: None)
    np.testing.assert_allclose(ctx.to_host(d_map), before + plain, rtol=1e-12, atol=0)


def test_sliced_offsets_equal_the_plain_call(cosmo):
    """bfg_baryonify_offsets_sliced: element ranges are 3 doubles per pixel; union == the plain call's offset field"""
    import torch
    import warnings
    nside = 128
    ra, dec, M, z = syn.catalog(2500, seed=78)
    zd, Md, rd, d = syn.displacement_table()
    ctx = bfg.engine.get_context()
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    model = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=syn.mass_map(nside), cosmo=cosmo), 10, model, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = R.offsets_device().cpu().numpy()
        keys = R._checked_model_keys()
        bg, spline, d_cat, stride = R._device_inputs(ctx, keys)
        table = ctx.table(bfg.Runners.HealpixRunner._table_axes(model, keys), np.asarray(model.raw_input_d), log_values=False)
        md = ctx.massdef_struct(bg, R.mass_def)
        args = ctx.shell_args(nside, d_cat, d_cat.shape[0], stride, 0, 10, md, model_md=md, model_epsilon_max=20.0,
                              out_overwrite=True)
        d_off = ctx.empty(12 * nside * nside, 3)
        got = torch.empty_like(d_off)
        seen = []

        def on_slice(k, n, lo, hi):
            seen.append((lo, hi))
            got.view(-1)[lo:hi] = d_off.view(-1)[lo:hi]
            d_off.view(-1)[lo:hi] = float("nan")
        ctx.baryonify_offsets(args, table, spline, d_off, slices=5, on_slice=on_slice)
    assert len(seen) == 5 and seen[0][0] == 0 and seen[-1][1] == 3 * 12 * nside * nside
    assert all(lo % 3 == 0 and hi % 3 == 0 for lo, hi in seen)
    got = got.cpu().numpy()
    # 4-neighbour fallback halos (< 4 pixels) go through the scatter kernel, whose atomics reorder the sums
    assert np.array_equal(got != 0, plain != 0)
    np.testing.assert_allclose(got, plain, rtol=1e-11, atol=1e-18)


def test_slice_callback_failure_aborts_the_call(cosmo):
    ra, dec, M, z = syn.catalog(300, seed=5)
    zax, Max, rax, T = syn.pressure_table()
    R = bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo),
                               bfg.LightconeShell(map=np.zeros(12 * 64 * 64), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False)

    def boom(k, n, lo, hi):
        raise KeyError("from the callback")
    with pytest.raises(KeyError, match="from the callback"):
        R.process_device(slices=4, on_slice=boom)
    got = R.process()                                         # the context is usable afterwards
    ref, _ = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 64, 10)
    assert_maps_close(got, ref, RTOL, what="after a failed callback")


def test_catalog_device_copy_is_locked_not_stale(cosmo):
    """the device copy of a catalog is reused between calls; numpy refuses in-place edits while it exists, unlock() drops it"""
    ra, dec, M, z = syn.catalog(400, seed=6)
    zax, Max, rax, T = syn.pressure_table()
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * 64 * 64), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False)
    a = R.process()
    with pytest.raises(ValueError):
        Cat.cat["M"][0] = 1e15
    assert np.array_equal(R.process(), a)
    Cat.unlock()
    Cat.cat["M"] *= 2.0
    ref, _ = oracle_paint(cosmo, ra, dec, 2 * M, z, (zax, Max, rax), T, 64, 10)
    assert_maps_close(R.process(), ref, RTOL, what="after unlock + edit")
    Cat.cat = Cat.cat.copy()                                  # another array object: noticed by identity
    Cat.cat["M"] /= 2.0
    ref, _ = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, 64, 10)
    assert_maps_close(R.process(), ref, RTOL, what="after replacing cat")
    # a bulk edit through a view that existed BEFORE the lock (numpy cannot revoke it): the sampled stamp notices
    Cat2 = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    m_view, z_view = Cat2.cat["M"], Cat2.cat["z"]
    R2 = bfg.PaintProfilesShell(Cat2, bfg.LightconeShell(map=np.zeros(12 * 64 * 64), cosmo=cosmo), 10,
                                _paint_model(zax, Max, rax, T), verbose=False)
    R2.process()
    assert m_view.flags.writeable and not Cat2.cat.flags.writeable
    m_view *= 3.0
    z_view += 0.02
    assert Cat2.z_max() == pytest.approx(float(np.max(z + 0.02)), rel=1e-15)
    ref, _ = oracle_paint(cosmo, ra, dec, 3 * M, z + 0.02, (zax, Max, rax), T, 64, 10)
    assert_maps_close(R2.process(), ref, RTOL, what="after a bulk edit through an older view")
    # a SINGLE element poked through such a view: invisible to the sampled stamp (documented) -- invalidate() is the way to say so,
    # and BFG_CATALOG_CACHE=full (a checksum of every byte per call) notices by itself
    m_view[1] *= 2.0                                          # (record 1 is not one of the ~256 sampled records of 400? it may be: use both paths)
    M2 = 3 * M
    M2[1] *= 2.0
    ref, _ = oracle_paint(cosmo, ra, dec, M2, z + 0.02, (zax, Max, rax), T, 64, 10)
    Cat2.invalidate()
    assert Cat2.cat.flags.writeable
    assert_maps_close(R2.process(), ref, RTOL, what="after invalidate()")


def test_catalog_full_checksum_notices_a_single_element_edit(cosmo, monkeypatch):
    """BFG_CATALOG_CACHE=full: the content stamp covers every byte of the catalog, so one element poked through a view taken before
    the lock refreshes the device copy (the default stamp samples ~256 records: documented, utils/io.py)"""
    monkeypatch.setenv("BFG_CATALOG_CACHE", "full")
    ra, dec, M, z = syn.catalog(3000, seed=6)                 # 3000 records: the sampled stamp would look at every 11th
    zax, Max, rax, T = syn.pressure_table()
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    m_view = Cat.cat["M"]
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * 64 * 64), cosmo=cosmo), 10,
                               _paint_model(zax, Max, rax, T), verbose=False)
    R.process()
    m_view[5] *= 4.0                                          # record 5 is not sampled (step 11)
    M2 = M.copy()
    M2[5] *= 4.0
    ref, _ = oracle_paint(cosmo, ra, dec, M2, z, (zax, Max, rax), T, 64, 10)
    assert_maps_close(R.process(), ref, RTOL, what="single-element edit under BFG_CATALOG_CACHE=full")


# --------------------------------------------------------------------------- a8 pin at the BASELINE resolutions
@pytest.mark.parametrize("nside", [1024, 2048])
def test_painted_pixel_sets_equal_brute_force_over_all_pixels(cosmo, nside):
    """Disc membership of the HIP kernels against an INDEPENDENT criterion at the workloads' resolutions:
    a table of T = 1 makes every (halo, pixel) pair paint exactly 1, so the painted map is the number of discs covering each
    pixel; the reference count is sum_j [ n_p . n_j > cos(theta_j) ] over ALL 12 nside^2 pixel centres, computed with torch
    from the closed-form ring formulae (tests/closed_form.py -- neither csrc/bfg_device.hpp nor the oracle).  ~500 halos incl.
    both poles and the phi = 0 seam; pixels within 1e-12 rad of a disc's rim are exempt."""
    import torch
    from closed_form import ring_pixel_vectors
    n, eps = 500, 10.0
    rng = np.random.default_rng(nside)
    ra, dec, M, z = syn.catalog(n, seed=nside)
    res_deg = np.degrees(np.sqrt(4 * np.pi / (12 * nside * nside)))
    dec[:30] = 90 - np.abs(rng.normal(0, 4 * res_deg, 30))                # north pole (the container clips |dec| = 90)
    dec[30:60] = -90 + np.abs(rng.normal(0, 4 * res_deg, 30))             # south pole
    ra[60:110] = rng.normal(0, 3 * res_deg, 50) % 360                      # the seam
    M[110:130] = 10 ** rng.uniform(11.0, 11.6, 20)                         # discs of a pixel or less (some empty)
    zax = np.log(1 + np.array([0.01, 1.0]))
    Max = np.log(np.array([1e10, 1e17]))
    rax = np.log(np.geomspace(1e-6, 1e4, 100))
    T = np.ones((2, 2, 100))
    model = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False)
    d_map = R.process_device()
    assert R.last_stats["fallback_halos"] == 0 and R.last_stats["pixels_out_of_table"] == 0
    a, Rr, D = orc.halo_scalars(cosmo, M, Cat.cat["z"])
    theta = torch.tensor(Rr * eps / D, dtype=torch.float64, device=d_map.device)
    dr = np.radians(Cat.cat["dec"]); rr = np.radians(Cat.cat["ra"])
    c = torch.tensor(np.stack([np.cos(dr) * np.cos(rr), np.cos(dr) * np.sin(rr), np.sin(dr)], 1), dtype=torch.float64,
                     device=d_map.device)
    v = ring_pixel_vectors(nside, xp=torch)
    count = torch.zeros(v.shape[0], dtype=torch.float64, device=d_map.device)
    rim = torch.zeros(v.shape[0], dtype=torch.bool, device=d_map.device)
    cos_t, sin_t = torch.cos(theta), torch.sin(theta)
    for j0 in range(0, n, 4):
        dot = v @ c[j0:j0 + 4].T                                           # [npix, 4]
        count += (dot > cos_t[j0:j0 + 4]).sum(dim=1).to(torch.float64)
        # |angle - theta| < 1e-12  <=>  |dot - cos(theta)| < 1e-12 sin(theta) (to first order)
        rim |= ((dot - cos_t[j0:j0 + 4]).abs() < 1e-12 * sin_t[j0:j0 + 4] + 4e-16).any(dim=1)
    ok = ~rim
    assert int(rim.sum()) < 50                                             # the exemption is a handful of pixels
    assert torch.equal(d_map[ok], count[ok]), int((d_map[ok] != count[ok]).sum())
    assert float(count.sum()) > 1e4 * n / 500 and abs(R.last_stats["pixel_updates"] - float(count.sum())) <= int(rim.sum())


@pytest.mark.parametrize("rows", ["fused", "separate"])
def test_paint_three_extra_table_axes(cosmo, rows, monkeypatch):
    """a 6-D table (three p_keys axes: BFG_MAX_EXTRA, 32 corners per halo): the prep kernel's fused row phase would need more
    than 64 KB of LDS there and hands the rows to halo_row4_kernel; against the oracle's N-linear read-out"""
    if rows == "separate":
        monkeypatch.setenv("BFG_ROWS", "separate")
    nside, n, eps = 256, 3000, 10.0
    ra, dec, M, z = syn.catalog(n, seed=271)
    rng = np.random.default_rng(6)
    p1, p2, p3 = rng.uniform(0.7, 1.4, n), rng.uniform(-1.0, 2.0, n), rng.uniform(10.0, 20.0, n)
    zax, Max, rax, T = syn.pressure_table(3, 8, 100)
    a1, a2, a3 = np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0])
    T6 = (T[..., None, None, None] * (1.0 + 0.3 * (a1 - 1.0))[None, None, None, :, None, None]
          * (1.0 + 0.05 * a2 ** 2)[None, None, None, None, :, None] * (a3 / 10.0)[None, None, None, None, None, :])
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax, a1, a2, a3), T6, nside, eps, extra=np.stack([p1, p2, p3], 1))
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, pa=p1, pb=p2, pc=p3)
    model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, T6, other_params={"pa": a1, "pb": a2, "pc": a3})
    for variant in ("tile_lds", "scatter_quarter"):
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model,
                                   verbose=False, variant=variant)
        got = R.process()
        assert R.last_stats["pixel_updates"] == ptot
        assert np.array_equal(got != 0, ref != 0)
        assert_maps_close(got, ref, RTOL, what=f"three extra axes ({rows}, {variant})")


def test_list_of_shells_on_one_gpu_pipelined(cosmo):
    """SplitJoinParallel over a list of shell runners without a process group: rotating map buffers, copies to the host on a copy
    stream (a buffer is repainted only after its copy has finished), stats collected once -- every map equals the runner's own
    process(), and SimpleParallel(split=True) is the same call"""
    nside = 256
    zax, Max, rax, T = syn.pressure_table()
    model = _paint_model(zax, Max, rax, T)
    runners, refs, ptots = [], [], 0
    for k, n in enumerate((3000, 10, 5000, 1, 2000)):
        ra, dec, M, z = syn.catalog(n, seed=500 + k)
        ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, 10)
        refs.append(ref); ptots += ptot
        runners.append(bfg.PaintProfilesShell(bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo),
                                              bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10, model, verbose=False))
    SJ = bfg.SplitJoinParallel(runners)
    outs = SJ.process()
    assert len(outs) == 5 and SJ.Runner_list[0].last_stats["pixel_updates"] == ptots
    for o, ref in zip(outs, refs):
        assert np.array_equal(o != 0, ref != 0)
        assert_maps_close(o, ref, RTOL, what="list of shells")
    outs2 = bfg.SimpleParallel(runners, split=True).process()
    assert all(np.allclose(a, b, rtol=1e-12, atol=0) for a, b in zip(outs, outs2))
    seen = []
    SJ.process_device(consume=lambda k, d: seen.append((k, d.cpu().numpy())))
    assert [k for k, _ in seen] == [0, 1, 2, 3, 4]
    assert all(np.allclose(m, o, rtol=1e-12, atol=0) for (_, m), o in zip(seen, outs))
    devs = SJ.process_device()
    assert len(devs) == 5 and all(np.allclose(d.cpu().numpy(), o, rtol=1e-12, atol=0) for d, o in zip(devs, outs))


def test_baryonify_list_on_one_gpu_overlaps_transfers_and_equals_the_plain_path(cosmo):
    """SimpleParallel / SplitJoinParallel over a list of BaryonifyShell runners on one GPU run the shells through
    _baryonify_pipelined (uploads, kernels and downloads of consecutive shells overlap; nothing is read back in between): the
    maps equal those of the step-by-step path (_baryonify_process) and the oracle, an all-zero shell comes back as the input
    array itself (HealpixRunner.py:293-294), the mass sums are checked, the counters are those of the whole list"""
    import warnings
    from baryonforge_amd.Runners.HealpixRunner import _BaryonifyDeviceOps, _baryonify_process
    nside, eps = 256, 10.0
    zd, Md, rd, d = syn.displacement_table()
    bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    runners, refs, ptots = [], [], 0
    for k, n in enumerate([3000, 0, 5000, 800, 2500]):
        ra, dec, M, z = syn.catalog(max(n, 1), seed=300 + k)
        m_in = syn.mass_map(nside) * (1.0 + 0.1 * k) if n else np.zeros(12 * nside * nside)
        Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
        runners.append(bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, bm, verbose=False))
        refs.append(oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, m_in) if n else None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        outs = bfg.SimpleParallel(runners).process()
        outs2 = bfg.SplitJoinParallel(runners).process()
        plain = [_baryonify_process(R, _BaryonifyDeviceOps(R), None) for R in runners]
        single = runners[2].process()
    assert outs[1] is runners[1].LightconeShell.map and outs2[1] is runners[1].LightconeShell.map      # the all-zero shell
    for k in (0, 2, 3, 4):
        assert_maps_close(outs[k], refs[k], RTOL, floor=BFLOOR, what=f"pipelined list, shell {k}")
        assert_maps_close(outs[k], plain[k], 1e-9, floor=BFLOOR, what=f"pipelined vs step by step, shell {k}")
        assert_maps_close(outs2[k], plain[k], 1e-9, floor=BFLOOR, what=f"SplitJoinParallel list, shell {k}")
        assert np.isclose(outs[k].sum(), runners[k].LightconeShell.map.sum(), rtol=1e-12)
    assert_maps_close(single, plain[2], 1e-9, floor=BFLOOR, what="single process()")
    assert runners[0].last_stats["pixel_updates"] > 0


def test_page_locked_shell_maps_give_the_same_result(cosmo):
    """LightconeShell(pinned=True) (= "copy") replaces the map by a page-locked copy, pinned="inplace" page-locks the caller's array
    itself (engine.pin -> hipHostRegister): BaryonifyShell.process() then moves the map asynchronously in slices
    -- same map as from pageable memory (to the rounding of the regrid's atomics), mass conserved, and unpin() gives the pages back."""
    from baryonforge_amd import engine
    nside = 128
    ra, dec, M, z = syn.catalog(3000, seed=12, logM=(13.0, 15.3))
    zd, Md, rd, d = syn.displacement_table()
    bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, d, cosmo, epsilon_max=20)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    m_in = syn.mass_map(nside)
    ref = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, bm, verbose=False).process()
    for mode in (True, "copy", "inplace", "inplace-heap"):
        if mode == "inplace":                                             # a page-aligned buffer that owns its pages: registered in place
            src = engine.aligned_empty(m_in.size)
            src[:] = m_in
            assert engine.pin_ok(src)
            shell = bfg.LightconeShell(map=src, cosmo=cosmo, pinned="inplace")
        elif mode == "inplace-heap":                                      # an ordinary numpy array: refused by pin(), copied with a warning
            src = m_in.copy()
            assert not engine.pin_ok(src)
            with pytest.raises(ValueError):
                engine.pin(src)
            with pytest.warns(UserWarning, match="page-locked copy"):
                shell = bfg.LightconeShell(map=src, cosmo=cosmo, pinned="inplace")
        else:
            src = m_in.copy()
            shell = bfg.LightconeShell(map=src, cosmo=cosmo, pinned=mode)
        assert engine.is_pinned(shell.map) and (shell.map is src) == (mode == "inplace")
        got = bfg.BaryonifyShell(Cat, shell, 10, bm, verbose=False).process()
        assert np.isclose(got.sum(), m_in.sum())
        assert_maps_close(got, ref, 1e-9, floor=1e-12, what=f"pinned={mode!r} vs pageable")
        assert np.array_equal(shell.map, m_in)                          # the input map is untouched
        if mode == "inplace":
            engine.unpin(src)
            assert not engine.is_pinned(src)
    # paint into a caller's page-locked output
    zax, Max, rax, T = syn.pressure_table()
    Rp = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10, _paint_model(zax, Max, rax, T),
                                verbose=False)
    out = engine.pinned_empty(12 * nside * nside)
    a, b = Rp.process(), Rp.process(out=out)
    assert b.base is not None and np.shares_memory(b, out) and np.array_equal(a != 0, b != 0)
    assert_maps_close(b, a, 1e-12, what="paint into a page-locked out= array")      # (LDS atomics: the order of additions is not fixed)
