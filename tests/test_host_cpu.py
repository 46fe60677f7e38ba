"""
CPU-only tests (no GPU): the host logic that mirrors the reference API, the C-ABI library
(loads, exports every symbol of include/bfg_mi355.h, refuses to compute without a GPU),
the host-side table builder against the reference's golden output, sky-patch sharding.
"""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest

import baryonforge_amd as bfg
from baryonforge_amd import _lib, sharding, synthetic as syn
from baryonforge_amd.background import Background, MassDef
from oracle import oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI library
def test_library_exports_every_declared_symbol():
    L = _lib.load()
    header = open(os.path.join(REPO, "include", "bfg_mi355.h")).read()
    declared = set(re.findall(r"\b(bfg_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.bfg_abi_version() == _lib.ABI_VERSION == 6
    assert L.bfg_status_string(0) == b"ok" and b"invalid" in L.bfg_status_string(-1)


def test_header_is_plain_c(tmp_path):
    """include/bfg_mi355.h must compile as C (the drop-in boundary is a C-ABI) and agree with the ctypes structs"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "bfg_mi355.h"\n#include <stdio.h>\n'
                   'int main(void){printf("%zu %zu %zu %d %zu %zu\\n", sizeof(bfg_massdef), sizeof(bfg_shell_args), '
                   'sizeof(bfg_stats), BFG_ABI_VERSION, sizeof(bfg_snapshot_args), sizeof(bfg_grid_args)); return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert [int(x) for x in out] == [ctypes.sizeof(_lib.MassDefStruct), ctypes.sizeof(_lib.ShellArgs),
                                     ctypes.sizeof(_lib.Stats), _lib.ABI_VERSION, ctypes.sizeof(_lib.SnapshotArgs),
                                     ctypes.sizeof(_lib.GridArgs)]


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.MassDefStruct) == 7 * 8 + 8
    assert ctypes.sizeof(_lib.Stats) == 4 * 8 + 8
    assert ctypes.sizeof(_lib.ShellArgs) == 8 + 8 + 8 + 4 + 4 + 8 + 2 * 64 + 8 + 4 * 4


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.load()
    n = ctypes.c_int(-1)
    assert L.bfg_device_count(ctypes.byref(n)) == 0 and n.value == 0
    h = ctypes.c_void_p()
    assert L.bfg_ctx_create(0, None, ctypes.byref(h)) == -3          # BFG_ERR_NO_DEVICE
    zax, Max, rax, T = syn.pressure_table(4, 5, 20)
    one = np.array([1.0])
    Cat = bfg.HaloLightConeCatalog(one, one, one * 1e14, one * 0.3, syn.COSMO)
    Shell = bfg.LightconeShell(map=np.zeros(12 * 4 * 4), cosmo=syn.COSMO)
    with pytest.raises(_lib.BFGError):                                 # no silent CPU fallback
        bfg.PaintProfilesShell(Cat, Shell, 10, bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False).process()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "baryonforge_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "libbfg_oracle" not in src, f


# ------------------------------------------------------------------ containers (io.py)
def test_catalog_and_shell_containers():
    c = dict(syn.COSMO)
    ra, dec, M, z = syn.catalog(50)
    cat = bfg.HaloLightConeCatalog(ra, dec, M, z, c, cdelta=np.arange(50.0))
    assert cat.cat.dtype.names == ("M", "z", "ra", "dec", "cdelta") and cat.cat.dtype[0] == np.float64
    assert cat.cosmology is c and cat.data is cat.cat
    sub = cat[10:20]
    assert isinstance(sub, bfg.HaloLightConeCatalog) and sub.cat.size == 10 and sub.cat["cdelta"][0] == 10
    sub = cat[np.array([3, 1])]
    assert sub.cat["M"][0] == M[3]
    rec = cat.records(["cdelta"])
    assert rec.shape == (50, 5) and rec[7, 0] == M[7] and rec[7, 2] == ra[7] and rec[7, 4] == 7
    with pytest.raises(ValueError):
        bfg.HaloLightConeCatalog(ra, dec, M, z, {"Omega_m": 0.3})
    with pytest.warns(UserWarning):
        p = bfg.HaloLightConeCatalog(np.zeros(2), np.array([90.0, -90.0]), M[:2], z[:2], c)
    assert p.cat["dec"][0] == 90 - 1e-8 and p.cat["dec"][1] == -90 + 1e-8
    sh = bfg.LightconeShell(map=np.zeros(12 * 8 * 8), cosmo=c, redshift=0.3)
    assert sh.NSIDE == 8 and sh.redshift == 0.3 and sh.data is sh.map
    with pytest.raises(ValueError):
        bfg.LightconeShell(map=np.zeros(100), cosmo=c)
    with pytest.raises(ValueError):
        bfg.LightconeShell(cosmo=c)
    with pytest.raises(ValueError):
        bfg.LightconeShell(map=np.zeros(12), cosmo={"h": 0.7})


def test_runner_attributes_and_errors():
    c = dict(syn.COSMO)
    cat = bfg.HaloLightConeCatalog(*syn.catalog(5), c)
    sh = bfg.LightconeShell(map=np.zeros(12), cosmo=c)
    R = bfg.PaintProfilesShell(cat, sh, 10, None)
    for k in ("HaloLightConeCatalog", "LightconeShell", "cosmo", "model", "epsilon_max", "mass_def", "verbose",
              "use_ellipticity", "include_pixel_size"):
        assert hasattr(R, k)                                            # read by Parallelize.py:237-243
    assert R.mass_def.Delta == 200 and R.mass_def.rho_type == "critical"
    with pytest.raises(NotImplementedError):
        bfg.BaryonifyShell(cat, sh, 10, None, use_ellipticity=True)
    with pytest.raises(AssertionError):
        R.process()                                                     # "You must provide a model"
    # zero map: early return of the very same array object, before any GPU work (HealpixRunner.py:293-294)
    assert bfg.BaryonifyShell(cat, sh, 10, None).process() is sh.map
    hm = bfg.regrid_pixels_hpix(np.zeros(10), np.ones(5), np.ones([5, 4], dtype=int), np.ones([5, 4]) * 0.25)
    assert hm[1] == 5.0 and hm.sum() == 5.0


# ------------------------------------------------------------------ background vs the oracle's independent one
def test_background_matches_oracle_background():
    for c in (dict(syn.COSMO), dict(syn.COSMO, w0=-0.8, Omega_m=0.25, h=0.68)):
        bg = Background(c)
        a = 1 / (1 + np.array([1e-3, 0.05, 0.45, 1.0, 3.0, 30.0]))
        np.testing.assert_allclose(bg.angular_diameter_distance(a), orc.angular_diameter_distance(c, a), rtol=1e-12)
        for md in (MassDef(200, "critical"), MassDef(500, "matter")):
            np.testing.assert_allclose(md.get_radius(c, 3e14, a), orc.get_radius(c, 3e14, a, md.Delta, md.rho_type),
                                       rtol=1e-14)
    # textbook anchor: flat LCDM Om = 0.3, h = 0.7 -> D_C(z = 0.5) ~ 1888.6 Mpc (radiation shifts it by ~1e-4)
    assert Background(syn.COSMO).comoving_radial_distance(1 / 1.5) == pytest.approx(1888.6, rel=3e-4)
    assert abs(Background(syn.COSMO).E2(1.0) - 1.0) < 1e-14


# ------------------------------------------------------------------ table construction (host)
class _Sigma(object):
    """analytic projected profile, same stand-in as tests/golden/make_golden.py"""

    def __init__(self, core, slope):
        self.core, self.slope, self.cutoff = core, slope, None

    def set_parameter(self, k, v):
        setattr(self, k, v)

    def projected(self, cosmo, r, M, a):
        M = np.atleast_1d(M)
        R = (orc.get_radius(syn.COSMO, M, a) / a)[:, None]
        x = np.atleast_1d(r)[None, :] / (self.core * R)
        S = M[:, None] / (2 * np.pi * (self.core * R) ** 2) * (1 + x * x) ** (-self.slope)
        return S * np.exp(-np.atleast_1d(r)[None, :] / (30 * R))

    real = projected


def test_baryonification2d_table_builder_matches_reference(golden):
    g = golden("table_builder.npz")
    DMO, DMB = _Sigma(0.25, 1.6), _Sigma(0.45, 1.6)
    B2 = bfg.Baryonification2D(DMO, DMB, dict(syn.COSMO), epsilon_max=20, N_int=500)
    assert DMO.cutoff == 1000 and DMB.cutoff == 1000                    # BaryonCorrection.py:99-100
    a = float(g["tb_a"])
    np.testing.assert_allclose(DMO.projected(None, g["tb_rint"], g["tb_M"], a) * a, g["tb_Sigma_DMO"], rtol=1e-13)
    np.testing.assert_allclose(B2.get_masses(DMO, g["tb_r"], g["tb_M"], a), g["tb_M_DMO"], rtol=1e-12)
    np.testing.assert_allclose(B2.get_masses(DMB, g["tb_r"], g["tb_M"], a), g["tb_M_DMB"], rtol=1e-12)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        B2.setup_interpolator(z_min=0.1, z_max=0.5, N_samples_z=3, M_min=1e12, M_max=1e16, N_samples_Mass=5,
                              R_min=1e-3, R_max=1e2, N_samples_R=50, verbose=False)
    assert B2.raw_input_d.shape == g["tb_d_interp"].shape and B2.p_keys == [] and B2.Rdelta_sampling is False
    np.testing.assert_allclose(B2.raw_input_d, g["tb_d_interp"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.exp(B2.raw_input_z_range) - 1, g["tb_z_tab"], rtol=1e-14)
    with pytest.raises(NotImplementedError):
        bfg.BaryonificationClass(DMO, DMB, syn.COSMO).get_masses(DMO, g["tb_r"], 1e14, a)
    with pytest.raises(NameError):
        bfg.Baryonification2D(DMO, DMB, syn.COSMO).displacement(1.0, 1e14, 0.8)


def test_tabulated_profile_setup_fills_reference_layout():
    class Model(_Sigma):
        mass_def = None
    m = Model(0.3, 1.5)
    prof = bfg.TabulatedProfile(m, syn.COSMO)
    with pytest.raises(NameError):
        prof.projected(None, 1.0, 1e14, 0.8)
    prof.setup_interpolator(z_min=0.01, z_max=1, N_samples_z=4, N_samples_Mass=5, N_samples_R=12, verbose=False)
    assert prof.raw_input_2D.shape == (4, 5, 12)
    z = np.geomspace(0.01, 1, 4)
    np.testing.assert_allclose(prof.raw_input_z_range, np.log(1 + z))
    np.testing.assert_allclose(prof.raw_input_M_range, np.log(np.geomspace(1e12, 1e16, 5)))
    np.testing.assert_allclose(prof.raw_input_r_range, np.log(np.geomspace(1e-3, 1e2, 12)))
    a1 = 1 / (1 + z[1])                                                  # the factor a of Tabulate.py:259
    np.testing.assert_allclose(prof.raw_input_2D[1], m.projected(None, np.geomspace(1e-3, 1e2, 12),
                                                                np.geomspace(1e12, 1e16, 5), a1) * a1)
    np.testing.assert_allclose(prof.raw_input_3D[1], m.real(None, np.geomspace(1e-3, 1e2, 12),
                                                          np.geomspace(1e12, 1e16, 5), a1))
    pp = bfg.ParamTabulatedProfile(m, syn.COSMO)
    pp.setup_interpolator(N_samples_z=2, N_samples_Mass=3, N_samples_R=6, other_params={"core": np.array([0.2, 0.4])},
                          verbose=False)
    assert pp.p_keys == ["core"] and pp.raw_input_2D.shape == (2, 3, 6, 2)
    assert np.all(pp.raw_input_2D[..., 0] != pp.raw_input_2D[..., 1])
    np.testing.assert_array_equal(pp.raw_input_core_range, [0.2, 0.4])
    with pytest.raises(AssertionError):
        bfg.ParamTabulatedProfile(prof, syn.COSMO)
    with pytest.raises(ValueError):
        bfg.TabulatedProfile.from_arrays(np.zeros(3), np.zeros(4), np.zeros(5), np.zeros((3, 4, 6)))


# ------------------------------------------------------------------ sharding
def test_ang2pix_nest_is_a_consistent_spatial_key():
    rng = np.random.default_rng(0)
    ra = np.degrees(rng.uniform(0, 2 * np.pi, 20000))
    dec = np.degrees(np.arcsin(rng.uniform(-1, 1, 20000)))
    for nside in (1, 8, 1024):
        p = sharding.ang2pix_nest(nside, ra, dec)
        assert p.min() >= 0 and p.max() < 12 * nside * nside
    # NEST hierarchy: parent pixel = child >> 2
    assert np.array_equal(sharding.ang2pix_nest(8, ra, dec), sharding.ang2pix_nest(16, ra, dec) >> 2)
    # equal-area: uniform points populate the 768 nside-8 pixels evenly
    cnt = np.bincount(sharding.ang2pix_nest(8, ra, dec), minlength=768)
    assert cnt.min() > 5 and cnt.max() < 60
    # base pixel of a few known directions (north cap faces 0..3, equatorial 4..7, south 8..11)
    assert list(sharding.ang2pix_nest(1, [45.0, 135.0, 0.0, 90.0, 45.0], [60.0, 60.0, 0.0, 0.0, -60.0])) == [0, 1, 4, 5, 8]
    # agrees with the oracle's RING geometry: points of one nside-4 NEST pixel are close on the sky
    v = orc.ang2vec(ra, dec, lonlat=True)
    p = sharding.ang2pix_nest(4, ra, dec)
    for k in rng.integers(0, 192, 10):
        vv = v[p == k]
        assert np.all(vv @ vv.mean(0) / np.linalg.norm(vv.mean(0)) > np.cos(0.45))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_by_sky_patch_contiguous_partitions_and_balances(world):
    ra, dec, M, z = syn.catalog(20000, seed=3)
    w = sharding.estimate_disc_pixels(syn.COSMO, M, z, 10, 1024)
    assert np.all(w > 16) and np.isfinite(w).all()
    shards = sharding.shard_by_sky_patch(ra, dec, w, world, nside_patch=8, layout="contiguous")
    assert len(shards) == world
    allidx = np.concatenate(shards)
    assert np.array_equal(np.sort(allidx), np.arange(20000))            # disjoint and complete
    loads = np.array([w[s].sum() for s in shards])
    assert loads.max() / loads.mean() < 1.25                            # balanced by pixel work, not by count
    if world > 1:                                                       # contiguous NEST ranges
        hi = [sharding.ang2pix_nest(8, ra[s], dec[s]).max() for s in shards if s.size]
        lo = [sharding.ang2pix_nest(8, ra[s], dec[s]).min() for s in shards if s.size]
        assert all(lo[i + 1] > hi[i] for i in range(len(hi) - 1))
    fine = sharding.ang2pix_nest(1024, ra[shards[0]], dec[shards[0]])
    assert np.all(np.diff(fine) >= 0)                                   # sorted for locality inside a shard


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_by_sky_patch_interleaved_covers_the_sky_on_every_rank(world):
    ra, dec, M, z = syn.catalog(40000, seed=4)
    w = sharding.estimate_disc_pixels(syn.COSMO, M, z, 10, 1024)
    shards = sharding.shard_by_sky_patch(ra, dec, w, world)               # the default layout: NSIDE-64 patches, interleaved
    assert len(shards) == world
    assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(40000))      # disjoint and complete
    loads = np.array([w[s].sum() for s in shards])
    assert loads.max() / loads.mean() < 1.1
    patch = sharding.ang2pix_nest(64, ra, dec)
    for r, s in enumerate(shards):
        assert np.all(np.diff(sharding.ang2pix_nest(1024, ra[s], dec[s])) >= 0)   # sorted by position inside a shard
        if world > 1:
            assert np.all(patch[s] % world == r)                        # whole patches, dealt round-robin
            assert np.unique(sharding.ang2pix_nest(2, ra[s], dec[s])).size == 48   # every rank sees the whole sky
    with pytest.raises(ValueError):
        sharding.shard_by_sky_patch(ra, dec, w, world, layout="random")


def test_estimate_disc_pixels_tracks_the_oracle_count():
    ra, dec, M, z = syn.catalog(300, seed=5)
    est = sharding.estimate_disc_pixels(syn.COSMO, M, z, 10, 256, overhead=0.0)
    a, R, D = orc.halo_scalars(syn.COSMO, M, z)
    cnt = np.array([orc.query_disc(256, orc.ang2vec(ra[i], dec[i], lonlat=True), R[i] * 10 / D[i]).size
                    for i in range(300)])
    assert abs(est.sum() / cnt.sum() - 1) < 0.05


def test_anis_runner_conventions(cosmo):
    """PaintProfilesAnisShell mirrors the reference constructor (HealpixRunner.py:501-511) and fails loudly off-GPU"""
    ra, dec, M, z = syn.catalog(10, seed=3)
    zax, Max, rax, T = syn.pressure_table()
    t = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.ones(12 * 8 * 8), cosmo=cosmo, redshift=0.2)
    R = bfg.PaintProfilesAnisShell(Cat, Shell, 10, t, t, t, 0.5, 0.1, verbose=False)
    assert (R.Tracer_model, R.Mtot_model, R.background_val, R.global_tracer_fraction) == (t, t, 0.5, 0.1)
    with pytest.raises(NotImplementedError):
        bfg.PaintProfilesAnisShell(Cat, Shell, 10, t, t, t, 0.5, 0.1, use_ellipticity=True)
    from baryonforge_amd.Runners.HealpixRunner import _ProductTable
    prod = _ProductTable(t, t, [])
    np.testing.assert_allclose(prod.ln_product, 2 * np.log(T))
    t2 = bfg.TabulatedProfile.from_arrays(zax, Max, rax + 0.1, T)
    with pytest.raises(ValueError):
        _ProductTable(t, t2, [])
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            R.process()


def test_snapshot_containers_and_runner_conventions(cosmo):
    """HaloNDCatalog keeps float32 big-endian columns (io.py:204), ParticleSnapshot float64 + NGP make_map (io.py:629-677)"""
    rng = np.random.default_rng(2)
    H = rng.uniform(0, 50, (7, 3))
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], 10 ** rng.uniform(13, 15, 7), 0.3, cosmo, z=H[:, 2], cdelta=np.arange(7.0))
    assert Cat.cat.dtype["M"] == np.dtype(">f4") and Cat.cat.dtype["x"] == np.dtype(">f4") and "cdelta" in Cat.cat.dtype.names
    assert Cat.cosmology is cosmo and Cat.redshift == 0.3 and Cat.data is Cat.cat
    Cat2 = bfg.HaloNDCatalog(H[:, 0], H[:, 1], np.ones(7), 0.3, cosmo)
    assert np.all(Cat2.cat["z"] == 0)
    with pytest.raises(ValueError):
        bfg.HaloNDCatalog(H[:, 0], H[:, 1], np.ones(7), 0.3, {"Omega_m": 0.3})
    P = rng.uniform(0, 50, (1000, 3))
    S = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2], M=np.full(1000, 2.0), L=50.0, redshift=0.3, cosmo=cosmo)
    assert not S.is2D and S.cat.dtype["x"] == np.float64
    m = S.make_map(8)
    assert m.shape == (8, 8, 8) and np.isclose(m.sum(), 2000.0)
    S2 = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], M=np.ones(1000), L=50.0, redshift=0.3, cosmo=cosmo)
    assert S2.is2D and S2.make_map(4).shape == (4, 4)
    zax, Max, rax, d = syn.displacement_table()
    model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
    R = bfg.BaryonifySnapshot(Cat, S, epsilon_max=10, model=model, verbose=False)
    assert R.cosmo is cosmo and R.epsilon_max == 10 and R.model is model
    np.testing.assert_allclose(R.compute_distance(np.array([49.0]), np.array([0.0])), [1.0])
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            R.process()


def test_grid_containers_and_host_regrid(golden, cosmo):
    """GriddedMap mirrors io.py:450-470; the host regrid_pixels_2D/_3D utilities reproduce the reference's overlap rule"""
    bins = (np.arange(16) + 0.5) * 2.0
    G = bfg.GriddedMap(map=np.zeros((16, 16)), redshift=0.1, bins=bins, cosmo=cosmo)
    assert G.is2D and G.Npix == 16 and G.res == 2.0 and G.L == 32.0 and G.inds.shape == (16, 16) and G.inds[2, 3] == 35
    G3 = bfg.GriddedMap(map=np.zeros((8, 8, 8)), redshift=0.1, bins=bins[:8], cosmo=cosmo)
    assert not G3.is2D and G3.inds.shape == (8, 8, 8)
    with pytest.raises(AssertionError):
        bfg.GriddedMap(map=np.zeros((16, 16)), redshift=0.1, bins=bins[:8], cosmo=cosmo)
    from oracle import oracle as orc_
    rng = np.random.default_rng(0)
    for nd, N in ((2, 12), (3, 7)):
        pos = rng.uniform(-3, N + 3, (200, nd))
        val = rng.uniform(0, 2, 200)
        a = np.zeros((N,) * nd)
        (bfg.regrid_pixels_2D if nd == 2 else bfg.regrid_pixels_3D)(a, pos, val)
        np.testing.assert_allclose(a, orc_.regrid_pixels_grid(N, pos, val, nd), rtol=1e-13, atol=1e-13)
        assert np.isclose(a.sum(), val.sum())
    Cat = bfg.HaloNDCatalog([1.0], [2.0], [1e14], 0.1, cosmo)
    with pytest.raises(AssertionError):                       # no q_ell / A_ell columns (Map2DRunner.py:272-278)
        bfg.PaintProfilesGrid(Cat, G, 4, None, use_ellipticity=True)
    R = bfg.BaryonifyGrid(Cat, G, 4, None, verbose=False)
    from oracle import oracle as orc2
    Rm = R.build_Rmat(np.array([0.6, 0.8], dtype=np.float32), np.float32(0.7))
    np.testing.assert_allclose(Rm, orc2.build_Rmat(np.array([0.6, 0.8], dtype=np.float32), np.float32(0.7)), rtol=1e-12)
    assert np.isclose(np.linalg.det(Rm), 1.0)
    np.testing.assert_array_equal(R.pick_indices(1, 2, 16), [15, 0, 1, 2])



def test_tabulated_correlation3d_host_side():
    X = bfg.utils.TabulatedCorrelation3D(dict(syn.COSMO))
    with pytest.raises(NameError):
        X(1.0, 0.5)
    with pytest.raises(ImportError):                       # no pyccl here and no callable given
        X.setup_interpolator()
    with pytest.raises(ValueError):
        bfg.utils.TabulatedCorrelation3D.from_arrays(np.zeros(3), np.zeros(5), np.ones((3, 4)))


def test_table_cache_never_hands_out_a_stale_table(monkeypatch):
    """engine.Context.table: entries are reused only for the very same model / array objects with unchanged contents.
    (A cache keyed by bare id() values returned the table of a collected model to a new one allocated at the same
    address: intermittently wrong maps.)"""
    from baryonforge_amd import engine

    class FakeTable(object):
        made = 0

        def __init__(self, ctx, axes, values, log_values):
            FakeTable.made += 1
            self.values = np.array(values, copy=True)

    monkeypatch.setattr(engine, "Table", FakeTable)
    ctx = engine.Context.__new__(engine.Context)          # no GPU needed for the cache logic
    ctx._table_cache = {}
    axes = [np.arange(3.0), np.arange(4.0), np.arange(5.0)]

    class Model(object):
        pass
    m, raw = Model(), np.ones((3, 4, 5))
    t1 = ctx.table(axes, raw, True, cache_key=(m, "2D", raw))
    assert ctx.table(axes, raw, True, cache_key=(m, "2D", raw)) is t1 and FakeTable.made == 1
    raw[1, 2, 3] = 7.0                                    # same objects, contents changed in place
    t2 = ctx.table(axes, raw, True, cache_key=(m, "2D", raw))
    assert t2 is not t1 and t2.values[1, 2, 3] == 7.0
    raw2 = raw.copy()                                     # equal contents, another array object
    assert ctx.table(axes, raw2, True, cache_key=(m, "2D", raw2)) is not t2
    # entries keep their key objects alive, so an id can never be recycled while its entry exists
    key_ids = {k[0] for k in ctx._table_cache}
    assert id(m) in key_ids and all(v[1][0] is m for v in ctx._table_cache.values())
    assert ctx.table(axes, raw, True, cache_key=(m, "3D", raw)) is not t2      # the tag is part of the key
    assert ctx.table(axes, lambda: raw, False, cache_key=None) is not None     # uncached
    # a large table edited in ONE place (an element a strided sample of the array would not see) misses the cache too
    big_axes = [np.arange(30.0), np.arange(30.0), np.arange(100.0)]
    big = np.ones((30, 30, 100))
    tb = ctx.table(big_axes, big, True, cache_key=(m, "big", big))
    assert ctx.table(big_axes, big, True, cache_key=(m, "big", big)) is tb
    big[17, 3, 41] = 2.0
    tb2 = ctx.table(big_axes, big, True, cache_key=(m, "big", big))
    assert tb2 is not tb and tb2.values[17, 3, 41] == 2.0
    # the stamp itself (engine._fingerprint, a 128-bit hash): EVERY byte counts, whatever the size -- any single element, the last
    # odd bytes, a swap of two values (sampling large tables let a single-cell edit through); BFG_TABLE_CACHE=sampled is
    # the documented opt-out for arrays over 4 MiB, BFG_TABLE_CACHE=0 keeps no tables at all
    fp = engine._fingerprint
    a = np.arange(100_000, dtype=np.float64)
    f0 = fp(a)
    for edit in (lambda x: x.__setitem__(54321, -1.0), lambda x: x.__setitem__(0, 0.5), lambda x: x.__setitem__(-1, 0.5)):
        b = a.copy(); edit(b)
        assert fp(b) != f0
    b = a.copy(); b[[10, 20]] = b[[20, 10]]
    assert fp(b) != f0                                    # (a swap of two values: a plain xor / sum of words would not see it)
    assert fp(np.frombuffer(b"abcdefghijk", dtype=np.uint8)) != fp(np.frombuffer(b"abcdefghijK", dtype=np.uint8))
    huge = np.zeros(1_000_000)                             # 8 MB
    h0 = fp(huge)
    huge[123_457] = 1.0                                    # one cell, on none of the pages a sampled stamp would read
    assert fp(huge) != h0
    monkeypatch.setenv("BFG_TABLE_CACHE", "sampled")
    assert fp(huge) == fp(np.zeros(1_000_000))             # the opt-out misses it (documented)
    huge[0] = 1.0
    assert fp(huge) != fp(np.zeros(1_000_000))             # first page: in the sample
    monkeypatch.setenv("BFG_TABLE_CACHE", "0")
    assert ctx.table(axes, raw, True, cache_key=(m, "3D", raw)) is not ctx.table(axes, raw, True, cache_key=(m, "3D", raw))
    monkeypatch.delenv("BFG_TABLE_CACHE")
    t3 = ctx.table(axes, raw, True, cache_key=(m, "3D", raw))
    assert ctx.table(axes, raw, True, cache_key=(m, "3D", raw)) is t3
    ctx.invalidate_tables()
    assert ctx.table(axes, raw, True, cache_key=(m, "3D", raw)) is not t3


def test_grid_runner_mirrors_the_halo_offset_assertion():
    """Map2DRunner.py:522 / :761 / :936 (2D branches): a halo left of the grid by more than a pixel, or at a NaN position,
    fails 'Halo offsets ... are larger than res'; +infinity and positions inside or right of the grid pass"""
    import baryonforge_amd as bfg
    from baryonforge_amd import synthetic as syn
    from baryonforge_amd.Runners.Map2DRunner import DefaultRunnerGrid

    class FakeCtx(object):
        def to_device(self, a):
            return a
    cosmo = dict(syn.COSMO)
    N, L = 64, 100.0
    bins = (np.arange(N) + 0.5) * (L / N)
    Map = bfg.GriddedMap(map=np.zeros((N, N)), redshift=0.2, bins=bins, cosmo=cosmo)

    def inputs(x, y):
        Cat = bfg.HaloNDCatalog(np.array(x, float), np.array(y, float), np.full(len(x), 1e14), 0.2, cosmo)
        return DefaultRunnerGrid(Cat, Map, 6, None, verbose=False)._device_inputs(FakeCtx(), [])
    halos, b = inputs([10.0, 99.9, 150.0, np.inf, bins[0] - 0.9 * L / N], [5.0, 5.0, 5.0, 5.0, 5.0])
    assert halos.shape == (5, 5) and np.array_equal(b, bins)
    for bad in (np.nan, -np.inf, bins[0] - 1.5 * L / N):
        with pytest.raises(AssertionError, match="larger than res"):
            inputs([10.0, bad], [5.0, 5.0])
        with pytest.raises(AssertionError, match="larger than res"):
            inputs([10.0, 20.0], [bad, 5.0])


def test_snapshot_runner_constructor_order_matches_reference():
    """SnapshotRunner.py:84-85: (HaloNDCatalog, ParticleSnapshot, epsilon_max, model, mass_def, verbose, KDTree_kwargs):
    a positional mass_def must land in mass_def (it sets R_j and the query radius, :222-225), not in KDTree_kwargs"""
    import inspect
    from baryonforge_amd.Runners.SnapshotRunner import DefaultRunnerSnapshot
    names = list(inspect.signature(DefaultRunnerSnapshot.__init__).parameters)
    assert names == ["self", "HaloNDCatalog", "ParticleSnapshot", "epsilon_max", "model", "mass_def", "verbose",
                     "KDTree_kwargs"]
    c = dict(syn.COSMO)
    x = np.array([1.0, 2.0])
    Cat = bfg.HaloNDCatalog(x, x, np.array([1e14, 2e14]), 0.3, c, z=x)
    Part = bfg.ParticleSnapshot(x=x, y=x, z=x, M=np.ones(2), L=10.0, redshift=0.3, cosmo=c)
    R = bfg.BaryonifySnapshot(Cat, Part, 5, None, MassDef(500, "critical"), False)
    assert R.mass_def.Delta == 500 and R.verbose is False
    assert bfg.BaryonifySnapshot(Cat, Part, 5, None).mass_def.Delta == 200


def test_grid_and_snapshot_runners_use_lcdm_like_the_reference():
    """Map2DRunner.py:462-465 / SnapshotRunner.py:197-200 build ccl.Cosmology WITHOUT w0; the shell runners pass it
    (HealpixRunner.py:283)"""
    from baryonforge_amd.background import lcdm
    c = dict(syn.COSMO, w0=-0.8)
    assert lcdm(c)["w0"] == -1.0 and c["w0"] == -0.8 and lcdm(c)["Omega_m"] == c["Omega_m"]
    a = 1 / 1.5
    assert Background(lcdm(c)).E2(a) != Background(c).E2(a)
    assert Background(lcdm(c)).E2(a) == Background(dict(c, w0=-1.0)).E2(a)


def test_missing_rccl_is_an_error_code_not_a_crash(tmp_path):
    """with no loadable librccl the communicator entry points must return BFG_ERR_COMM (-6) and leave a message
    in bfg_last_error() -- not crash on a null dlerror().  A fresh process: the load is attempted once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"L = C.CDLL({_lib.so_path()!r})\n"
        "L.bfg_last_error.restype = C.c_char_p\n"
        "buf = C.create_string_buffer(128)\n"
        "L.bfg_comm_unique_id.argtypes = [C.c_char_p, C.c_size_t]\n"
        "rc = [L.bfg_comm_unique_id(buf, 128) for _ in range(3)]\n"
        "print(rc, L.bfg_last_error().decode())\n")
    env = dict(os.environ, BFG_RCCL_SO=str(tmp_path / "no_such_librccl.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("[-6, -6, -6]") and "no_such_librccl" in out.stdout, out.stdout


def test_background_against_independent_quadrature_and_closed_form():
    """a9 pins: D_A(z) = a c/H0 int_a^1 da' / (a'^2 E(a')) against a high-order Gauss-Legendre quadrature
    written here from the CCL parameter definitions, and R_200c against its closed form, at 50 redshifts, to 1e-12 -- for the
    product's Background, the oracle's background and the device-side spline knots' source alike"""
    from numpy.polynomial.legendre import leggauss
    cosmo = {"Omega_m": 0.31, "Omega_b": 0.045, "h": 0.68, "sigma8": 0.8, "n_s": 0.96, "w0": -0.9}
    bg = Background(cosmo)
    h = cosmo["h"]
    # photons + 3.044 massless neutrinos (CCL defaults: T_CMB 2.7255 K, T_ncdm 0.71611), from first principles
    sigma_sb, c, G_Msun, Mpc = 5.670374419e-8, 299792458.0, 1.3271244e20, 3.085677581491367e22
    rho_crit_si = 3 * (100e3 * h / Mpc) ** 2 / (8 * np.pi) * (1.98840987e30 / 1.3271244e20) ** 0 / (G_Msun / 1.98840987e30)
    rho_g = 4 * sigma_sb / c ** 3 * 2.7255 ** 4
    Og = rho_g / rho_crit_si
    Onu = 3.044 * 7.0 / 8.0 * (0.71611 ** 4) * Og
    Or = Og + Onu
    Ol = 1 - cosmo["Omega_m"] - Or

    def E(a):
        return np.sqrt(cosmo["Omega_m"] / a ** 3 + Ol * a ** (-3 * (1 + cosmo["w0"])) + Or / a ** 4)
    np.testing.assert_allclose(bg.Omega_r, Or, rtol=1e-9)                    # same radiation content from independent constants
    x, w = leggauss(96)
    zs = np.linspace(0.01, 3.0, 50)
    chi = np.empty_like(zs)
    for k, zz in enumerate(zs):
        a0 = 1 / (1 + zz)
        aa = 0.5 * (1 - a0) * x + 0.5 * (1 + a0)
        chi[k] = 0.5 * (1 - a0) * np.sum(w / (aa ** 2 * np.sqrt(cosmo["Omega_m"] / aa ** 3 + bg.Omega_l * aa ** (-3 * (1 + cosmo["w0"])) + bg.Omega_r / aa ** 4)))
    DA_ref = chi * (c / 1e3 / (100 * h)) / (1 + zs)
    np.testing.assert_allclose(bg.angular_diameter_distance(1 / (1 + zs)), DA_ref, rtol=1e-12)
    np.testing.assert_allclose(orc.angular_diameter_distance(cosmo, 1 / (1 + zs)), DA_ref, rtol=1e-12)
    M = 10 ** np.linspace(12, 15.5, 50)
    a = 1 / (1 + zs)
    E2 = cosmo["Omega_m"] / a ** 3 + bg.Omega_l * a ** (-3 * (1 + cosmo["w0"])) + bg.Omega_r / a ** 4
    R_ref = (M / (4.18879020479 * 200 * 2.775366e11 * h * h * E2)) ** (1 / 3)
    np.testing.assert_allclose(MassDef(200, "critical").get_radius(cosmo, M, a), R_ref, rtol=2e-7)   # CCL's rounded RHO_CRITICAL
    from baryonforge_amd.background import RHO_CRITICAL
    R_exact = (M / (4.18879020479 * 200 * RHO_CRITICAL * h * h * E2)) ** (1 / 3)
    np.testing.assert_allclose(MassDef(200, "critical").get_radius(cosmo, M, a), R_exact, rtol=1e-12)
    np.testing.assert_allclose(orc.get_radius(cosmo, M, a), R_exact, rtol=1e-12)


def test_massless_neutrino_convention_is_a_parameter_and_moves_distances_by_parts_in_1e7():
    """The one constants-level ambiguity of the background: Omega_nu,rel = N_eff 7/8 x^4 Omega_gamma with x = T_ncdm =
    0.71611 (the default, pyccl's T_nu = T_CMB T_ncdm as recalled) or x = (4/11)^(1/3).  Both are selectable in the product's
    Background and in the oracle (cosmo["nu_rel"]); they agree with each other convention by convention, and the choice moves D_A by
    2.5e-7 at z = 0.5 and ~1.2e-6 at z = 3 -- far inside the 1e-5 map tolerance, recorded so that a run against live pyccl can settle it."""
    cosmo = dict(syn.COSMO)
    zs = np.array([0.1, 0.5, 1.0, 3.0])
    a = 1 / (1 + zs)
    d = {}
    for conv in ("T_ncdm", "4/11"):
        bg = Background(cosmo, nu_rel=conv)
        d[conv] = bg.angular_diameter_distance(a)
        np.testing.assert_allclose(orc.angular_diameter_distance(dict(cosmo, nu_rel=conv), a), d[conv], rtol=1e-12)
        x = 0.71611 if conv == "T_ncdm" else (4 / 11) ** (1 / 3)
        Og = bg.Omega_r / (1 + 3.044 * 7 / 8 * x ** 4)
        np.testing.assert_allclose(bg.Omega_r - Og, 3.044 * 7 / 8 * x ** 4 * Og, rtol=1e-12)
    np.testing.assert_array_equal(Background(cosmo).angular_diameter_distance(a), d["T_ncdm"])       # the default
    rel = np.abs(d["4/11"] / d["T_ncdm"] - 1)
    assert 1.5e-7 < rel[1] < 3.5e-7 and 0.8e-6 < rel[3] < 1.6e-6 and np.all(rel < 2e-6), rel
    with pytest.raises(ValueError):
        Background(cosmo, nu_rel="other")


def test_product_background_reproduces_what_live_pyccl_printed_in_the_reference_notebooks():
    """the product's Background (host side of every D_A spline and R_200c the GPU path uses) against the two numbers live pyccl
    printed in the reference's example notebooks (tests/golden/pyccl_notebook_outputs.json): chi(z_max) - chi(z_min) to 2e-7"""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pyccl_notebook_outputs.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 2
    for case in cases:
        c = case["cosmology"]
        cosmo = {"Omega_m": c["Omega_c"] + c["Omega_b"], "Omega_b": c["Omega_b"], "h": c["h"], "sigma8": c["sigma8"], "n_s": c["n_s"], "w0": -1.0}
        bg = Background(cosmo)
        chi = bg.comoving_radial_distance(np.array([1 / (1 + case["max_z"]), 1 / (1 + case["min_z"])]))
        assert abs((chi[0] - chi[1]) / case["shell_thickness_mpc"] - 1) < 2e-7, case["source"]
        # and through the angular diameter distance, which is what the runners spline (HealpixRunner.py:297-299)
        zz = np.array([case["max_z"], case["min_z"]])
        chi2 = bg.angular_diameter_distance(1 / (1 + zz)) * (1 + zz)
        assert abs((chi2[0] - chi2[1]) / case["shell_thickness_mpc"] - 1) < 2e-7


@pytest.mark.parametrize("world", [2, 4, 8])
def test_stripe_shards_partition_and_extents_cover_the_painted_pixels(world):
    """sharding.shard_by_stripes: every halo exactly once, equal-area stripes get ~equal counts; stripe_extent: a RING pixel
    range that contains every pixel the oracle paints for the shard (incl. shards that reach a pole), tight to a few rings"""
    from util import oracle_paint
    cosmo = dict(syn.COSMO)
    nside, n = 128, 8000
    ra, dec, M, z = syn.catalog(n, seed=77)
    dec[:3] = [89.5, -89.7, 0.01]
    shards = sharding.shard_by_stripes(ra, dec, world)
    assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(n))
    assert min(s.size for s in shards) > 0.7 * n / world
    th = sharding.disc_radius(cosmo, M, z, 10.0)
    zax, Max, rax, T = syn.pressure_table(4, 8, 40)
    npix = 12 * nside * nside
    for r, idx in enumerate(shards):
        e0, e1 = sharding.stripe_extent(nside, dec[idx], th[idx])
        m, _ = oracle_paint(cosmo, ra[idx], dec[idx], M[idx], z[idx], (zax, Max, rax), T, nside, 10.0)
        nz = np.flatnonzero(m)
        assert 0 <= e0 <= nz.min() and nz.max() < e1 <= npix, (r, e0, e1, nz.min(), nz.max())
        assert (e1 - e0) < (1.0 / world + 0.25) * npix                     # own part + borders (NSIDE 128: large discs)
    assert sharding.stripe_extent(nside, np.array([]), np.array([])) == (0, 0)
    assert sharding.stripe_extent(nside, np.array([10.0]), np.array([np.pi])) == (0, npix)     # a disc over the whole sky


def test_slice_cuts_depend_on_nside_and_slice_count_only():
    """bfg_shell_slice_cuts needs no GPU: the slices of a sliced call are whole bands of the loop's tile geometry (32 rings for paint,
    16 for the offset field, 3 elements per pixel), at most 16, covering the output once -- what lets every rank of a process group
    issue the same collectives whatever its shard holds"""
    from baryonforge_amd import _lib
    for nside in (1, 2, 8, 64, 1024, 2048):
        npix = 12 * nside * nside
        nrings = 4 * nside - 1
        for offsets, tr, per in ((False, 32, 1), (True, 16, 3)):
            nbands = (nrings + tr - 1) // tr
            for slices in (1, 2, 3, 4, 7, 16, 17, 100):
                cuts = _lib.shell_slice_cuts(nside, offsets, slices)
                K = len(cuts) - 1
                assert K == max(1, min(slices, 16, nbands))
                assert cuts[0] == 0 and cuts[-1] == per * npix and all(a < b for a, b in zip(cuts, cuts[1:]))
                assert all(c % per == 0 for c in cuts)
                assert cuts == _lib.shell_slice_cuts(nside, offsets, slices)
    # power-of-two NSIDE: band boundaries fall on multiples of 32 pixels, so the slices of the offset field split evenly over 2 / 4 / 8 ranks
    for nside in (64, 1024, 2048):
        cuts = _lib.shell_slice_cuts(nside, True, 4)
        assert all(((b - a) // 3) % 8 == 0 for a, b in zip(cuts, cuts[1:]))
    L = _lib.load()
    import ctypes
    buf, n = (ctypes.c_int64 * 17)(), ctypes.c_int(0)
    assert L.bfg_shell_slice_cuts(0, 0, 4, buf, ctypes.byref(n)) == -1
    assert L.bfg_shell_slice_cuts(64, 0, 0, buf, ctypes.byref(n)) == -1
    assert L.bfg_shell_slice_cuts(64, 0, 4, None, ctypes.byref(n)) == -1


def test_full_catalog_stamp_sees_permutations(monkeypatch):
    """BFG_CATALOG_CACHE=full stamps every byte of the catalog with a position-dependent 128-bit hash: a swap of two halos' masses
    or a shuffle through a view taken before the lock changes it (a wrapping sum / xor of the words would not: ADVICE r5)"""
    import baryonforge_amd as bfg
    from baryonforge_amd import synthetic as syn
    ra, dec, M, z = syn.catalog(1000, seed=5)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, dict(syn.COSMO))
    monkeypatch.setenv("BFG_CATALOG_CACHE", "full")
    s0 = Cat._sample_stamp()
    assert Cat._sample_stamp() == s0
    m = Cat.cat["M"]
    m[[3, 7]] = m[[7, 3]]                                   # a swap: same multiset of words
    s1 = Cat._sample_stamp()
    assert s1 != s0
    np.random.default_rng(0).shuffle(m)                     # a permutation of one column
    assert Cat._sample_stamp() not in (s0, s1)
    m[5] = np.nextafter(m[5], np.inf)                       # one ulp of one element
    s3 = Cat._sample_stamp()
    m[5] = np.nextafter(m[5], -np.inf)
    assert Cat._sample_stamp() != s3
