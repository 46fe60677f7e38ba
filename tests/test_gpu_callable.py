"""
GPU tests of the shell runners with a model that is NOT tabulated: any object with .projected(cosmo, r, M, a) /
.displacement(r, M, a), which the reference calls once per halo (Runners/HealpixRunner.py:472, :345).  The callable
stays on the host; the disc enumeration, the distances, the scatter-add, the offset geometry and the regrid are HIP
kernels (csrc/bfg_enum.hpp) -- compared here with the oracle's line-by-line restatement of the reference loops.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from oracle import oracle as orc
from util import assert_maps_close


class AnalyticPressure(object):
    """a projected profile in closed form, with the awkward values a real model produces: inf at r = 0 is impossible here
    (pixel centres never coincide with a halo centre) but NaN beyond a cut and negative values are"""

    def __init__(self, nan_beyond=None):
        self.nan_beyond = nan_beyond
        self.calls = 0

    def projected(self, cosmo, r, M, a):
        self.calls += 1
        r = np.asarray(r, dtype=np.float64)
        rc = 0.3 * (M / 1e14) ** (1 / 3)
        out = 1e-6 * (M / 1e14) ** (5 / 3) / a ** 2 * (1 + (r / rc) ** 2) ** -1.5 * np.cos(r / (4 * rc))
        if self.nan_beyond is not None:
            out = np.where(r > self.nan_beyond * rc, np.nan, out)
        return out


class AnalyticDisplacement(object):
    def displacement(self, r, M, a):
        r = np.asarray(r, dtype=np.float64)
        x = r / (0.8 * (M / 1e14) ** (1 / 3) / a)
        return 0.1 * (M / 1e14) ** (1 / 3) * x * (1 - x / 4) * np.exp(-x)


@pytest.mark.parametrize("include_pixel_size", [False, True])
@pytest.mark.parametrize("batch", [None, 3000])
def test_paint_callable_model_vs_oracle(cosmo, include_pixel_size, batch, monkeypatch):
    nside, n, eps = 128, 400, 6.0
    if batch:
        monkeypatch.setenv("BFG_CALLABLE_BATCH", str(batch))               # several batches of halos
    ra, dec, M, z = syn.catalog(n, seed=5, logM=(13.0, 15.5))
    ra[0], dec[0] = 10.0, 89.9                                             # one disc over the north pole
    ra[1], dec[1] = 359.99, -0.01                                          # one across phi = 0
    model = AnalyticPressure(nan_beyond=8.0)
    ref, ptot = orc.paint_shell_callable(cosmo, nside, ra, dec, M, z, eps,
                                         lambda r, Mj, aj: model.projected(None, r, Mj, aj), include_pixel_size)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model,
                               include_pixel_size=include_pixel_size, verbose=False)
    model.calls = 0
    got = R.process()
    assert model.calls == n                                                # once per halo, as the reference
    assert R.last_stats["pixel_updates"] == ptot
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, 1e-10, what="paint with a callable model")
    # the device-map entry point accumulates INTO a given map
    d = R.process_device()
    d2 = R.process_device(d_map=d)
    assert_maps_close(d2.cpu().numpy(), 2 * ref, 1e-10, what="accumulate into a given map")


def test_baryonify_callable_model_vs_oracle(cosmo):
    nside, n, eps = 128, 300, 4.0
    ra, dec, M, z = syn.catalog(n, seed=6, logM=(12.0, 15.3))              # the small halos have < 4 pixels at NSIDE 128
    model = AnalyticDisplacement()
    off, ptot = orc.baryonify_offsets_callable(cosmo, nside, ra, dec, M, z, eps, model.displacement)
    m_in = syn.mass_map(nside)
    ref = orc.regrid_shell(nside, off, m_in)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, model, verbose=False)
    d_off = R.offsets_device()
    assert R.last_stats["pixel_updates"] == ptot
    got_off = d_off.cpu().numpy()
    # (unit-vector differences: a far pixel's offset of ~1e-17 may round to exactly 0 on one side and not on the other)
    touched, touched_ref = np.any(np.abs(got_off) > 1e-14, axis=1), np.any(np.abs(off) > 1e-14, axis=1)
    assert touched_ref.sum() > 500 and np.sum(touched != touched_ref) <= 2
    np.testing.assert_allclose(got_off, off, rtol=1e-7, atol=1e-15 + 1e-9 * np.abs(off).max())
    got = R.process()
    assert np.isclose(got.sum(), m_in.sum())
    assert_maps_close(got, ref, 1e-5, floor=1e-9, what="baryonify with a callable model")


def test_callable_model_argument_checks(cosmo):
    ra, dec, M, z = syn.catalog(10, seed=1)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Shell = bfg.LightconeShell(map=np.ones(12 * 16 * 16), cosmo=cosmo)
    with pytest.raises(TypeError):
        bfg.PaintProfilesShell(Cat, Shell, 5, object(), verbose=False).process()
    with pytest.raises(TypeError):
        bfg.BaryonifyShell(Cat, Shell, 5, AnalyticPressure(), verbose=False).process()
    with pytest.raises(AssertionError):
        bfg.PaintProfilesShell(Cat, Shell, 5, None, verbose=False).process()


@pytest.mark.parametrize("n_extra", [4, 5])
@pytest.mark.parametrize("batched", [False, True])
def test_tables_with_more_than_three_extra_axes_on_the_device(cosmo, n_extra, batched, monkeypatch):
    """ParamTabulatedProfile / BaryonificationClass are N-dimensional in the reference (utils/Tabulate.py:497-650,
    BaryonCorrection.py:211-227, :404-408); the shell kernels read at most three p_keys axes (BFG_MAX_EXTRA).  Tables with four and
    five are the same bfg_paint_shell / bfg_baryonify_offsets calls: the library blends the 2^(n + 2) corners once per halo into the
    halo's radial row (csrc/bfg_ndtable.hpp) and runs the tile path on the rows, nothing on the host -- the models' own scipy
    read-out must not be called.  Against the oracle's N-linear loops, incl. halos outside the hull of a parameter axis (NaN ->
    nothing painted / no displacement); `batched`: rows of 37 halos at a time (each batch accumulates into the one before)."""
    from util import oracle_baryonify, oracle_paint
    nside, n, eps = 128, 300, 6.0
    if batched:
        monkeypatch.setenv("BFG_ND_ROW_BYTES", str(8 * 60 * 37))          # rows of 37 halos per batch
    ra, dec, M, z = syn.catalog(n, seed=77, logM=(13.0, 15.3))
    rng = np.random.default_rng(8)
    p = [rng.uniform(0.7, 1.4, n), rng.uniform(-1.0, 2.0, n), rng.uniform(10.0, 20.0, n), rng.uniform(0.0, 1.0, n),
         rng.uniform(2.0, 3.0, n)][:n_extra]
    p[1][:7] = 2.6                                                        # outside the second parameter axis: NaN rows
    ax = [np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0]), np.array([-0.5, 0.5, 1.5]),
          np.array([1.5, 2.5, 3.5])][:n_extra]
    fac = np.ones([a.size for a in ax])
    for k, (a, f) in enumerate(zip(ax, [lambda x: 1.0 + 0.3 * (x - 1.0), lambda x: 1.0 + 0.05 * x ** 2, lambda x: x / 10.0,
                                        lambda x: 1.0 + 0.2 * x, lambda x: 0.5 + 0.2 * x])):
        shape = [1] * n_extra
        shape[k] = a.size
        fac = fac * f(a).reshape(shape)
    zax, Max, rax, T = syn.pressure_table(3, 8, 60)
    TN = T.reshape(T.shape + (1,) * n_extra) * fac[None, None, None]
    keys = ["pa", "pb", "pc", "pd", "pe"][:n_extra]
    extra = np.stack(p, 1)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax, *ax), TN, nside, eps, extra=extra)
    model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax)))

    def forbidden(*a, **k):
        raise AssertionError("the host read-out of the table was called")
    monkeypatch.setattr(model, "projected", forbidden, raising=False)
    for pix_size in (False, True):
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model,
                                   include_pixel_size=pix_size, verbose=False)
        got = R.process()
        if pix_size:
            ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax, *ax), TN, nside, eps, extra=extra, include_pixel_size=True)
        assert R.last_stats["pixel_updates"] == ptot and np.array_equal(got != 0, ref != 0)
        assert_maps_close(got, ref, 1e-9, what=f"paint, {n_extra} extra table axes (pixel size {pix_size})")
    d2 = R.process_device(d_map=R.process_device())                       # accumulates INTO a given map
    assert_maps_close(d2.cpu().numpy(), 2 * ref, 1e-9, what="accumulate into a given map")

    for rdelta in (False, True):
        zd, Md, rd, d = syn.displacement_table(3, 8, 60, rdelta=rdelta)
        dN = d.reshape(d.shape + (1,) * n_extra) * fac[None, None, None]
        m_in = syn.mass_map(nside)
        refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd, *ax), dN, nside, eps, 20, m_in, extra=extra, rdelta=rdelta)
        bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, dN, cosmo, epsilon_max=20, other_params=dict(zip(keys, ax)),
                                               **({"Rdelta_sampling": True} if rdelta else {}))
        monkeypatch.setattr(bm, "displacement", forbidden, raising=False)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, bm, verbose=False).process()
        assert not np.allclose(gotb, m_in)
        assert np.isclose(gotb.sum(), m_in.sum())
        assert_maps_close(gotb, refb, 1e-5, floor=1e-9, what=f"baryonify, {n_extra} extra table axes (Rdelta_sampling {rdelta})")


def test_nd_table_on_the_tile_path_equals_the_3d_table_at_full_size(cosmo):
    """A table with four p_keys axes whose values do not depend on them IS its 3-D table: at BASELINE configs[1]'s size (1e5 halos,
    NSIDE 1024) the N-dimensional call -- rows blended per halo, then the tile kernels on the rows -- must give the 3-D call's map
    (the blend's weights sum to 1: rounding only), the same P_tot, and must not fall back to the global-atomic kernel."""
    nside, n, eps = 1024, 100_000, 10.0
    ra, dec, M, z = syn.catalog(n, seed=42)
    zax, Max, rax, T = syn.pressure_table()
    ax = [np.array([0.0, 1.0, 2.0]), np.array([-1.0, 1.0]), np.array([0.0, 0.5, 1.0]), np.array([10.0, 20.0])]
    rng = np.random.default_rng(5)
    p = [rng.uniform(a[0], a[-1], n) for a in ax]
    TN = np.ascontiguousarray(np.broadcast_to(T.reshape(T.shape + (1,) * 4), T.shape + tuple(a.size for a in ax)))
    keys = ["pa", "pb", "pc", "pd"]
    Cat3 = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    CatN = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))
    shell = lambda: bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo)
    R3 = bfg.PaintProfilesShell(Cat3, shell(), eps, bfg.TabulatedProfile.from_arrays(zax, Max, rax, T), verbose=False)
    RN = bfg.PaintProfilesShell(CatN, shell(), eps, bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax))),
                                verbose=False)
    ref, got = R3.process(), RN.process()
    assert RN.last_stats["pixel_updates"] == R3.last_stats["pixel_updates"] > 2e7
    assert RN.last_stats["fallback_halos"] == 0
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, 1e-11, what="4 degenerate p_keys axes vs the 3-D table")


@pytest.mark.parametrize("batched", [False, True])
def test_nd_table_sliced_calls_report_final_slices(cosmo, batched, monkeypatch):
    """bfg_paint_shell_sliced / bfg_baryonify_offsets_sliced with a table of four p_keys axes: in one batch the slices are reported
    as the tile kernel finishes them (each final when reported: it is copied out at once and must equal the unsliced result there); in
    several batches of halos (rows capped by BFG_ND_ROW_BYTES) the same K ranges are reported after the last batch.  The slice
    ranges are those of the 3-D call (a function of NSIDE and the slice count only), so ranks with different tables stay in step."""
    import torch
    nside, n, eps = 64, 500, 8.0
    if batched:
        monkeypatch.setenv("BFG_ND_ROW_BYTES", str(8 * 60 * 90))
    ra, dec, M, z = syn.catalog(n, seed=21, logM=(13.0, 15.3))
    rng = np.random.default_rng(3)
    ax = [np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0]), np.array([-0.5, 0.5, 1.5])]
    p = [rng.uniform(a[0], a[-1], n) for a in ax]
    keys = ["pa", "pb", "pc", "pd"]
    zax, Max, rax, T = syn.pressure_table(3, 8, 60)
    fac = (1.0 + 0.3 * (ax[0] - 1.0))[:, None, None, None] * (1.0 + 0.05 * ax[1] ** 2)[None, :, None, None] * \
        (ax[2] / 10.0)[None, None, :, None] * (1.0 + 0.2 * ax[3])[None, None, None, :]
    TN = T[..., None, None, None, None] * fac[None, None, None]
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))
    model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax)))
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False)
    whole = R.process_device().clone()
    got, ranges = torch.full_like(whole, float("nan")), []

    def on_slice(k, K, lo, hi):
        ranges.append((k, K, lo, hi))
        got[lo:hi] = d_map[lo:hi]                                         # enqueued behind the kernels of this slice
    d_map = torch.full_like(whole, float("nan"))
    R.process_device(d_map=d_map, overwrite=True, slices=4, on_slice=on_slice)
    torch.cuda.synchronize()
    assert [r[0] for r in ranges] == list(range(len(ranges))) and len(ranges) == ranges[0][1] == 4
    assert ranges[0][2] == 0 and ranges[-1][3] == whole.numel() and all(a[3] == b[2] for a, b in zip(ranges, ranges[1:]))
    assert torch.allclose(got, whole, rtol=1e-12, atol=0.0) and torch.equal(got != 0, whole != 0)
    ref, ptot = oracle_paint_nd(cosmo, ra, dec, M, z, (zax, Max, rax, *ax), TN, nside, eps, np.stack(p, 1))
    assert_maps_close(whole.cpu().numpy(), ref, 1e-9, what="4 extra axes, unsliced")


def oracle_paint_nd(cosmo, ra, dec, M, z, axes, TN, nside, eps, extra):
    from util import oracle_paint
    return oracle_paint(cosmo, ra, dec, M, z, axes, TN, nside, eps, extra=extra)


@pytest.mark.parametrize("variant,nr", [("auto", 60), ("scatter_quarter", 60), ("scatter_wave", 60), ("auto", 700)])
def test_nd_table_rows_through_every_kernel_variant(cosmo, variant, nr):
    """the per-halo rows of a table with four p_keys axes (DevTable::hstride) are read by the tile kernel's window path, by the
    scatter kernels (left-overs / scatter variants) and by the table-direct read-out of finely sampled radial axes (nr = 700: no row
    windows, the pixel stage reads the halo's row itself): each against the oracle's N-linear loops, paint and offsets."""
    nside, n, eps = 128, 400, 6.0
    ra, dec, M, z = syn.catalog(n, seed=31, logM=(13.0, 15.3))
    ra[:3], dec[:3] = [10.0, 200.0, 359.9], [89.7, -89.8, 0.0]           # both poles and the seam: left-overs of the tile variant
    rng = np.random.default_rng(4)
    ax = [np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0]), np.array([-0.5, 0.5, 1.5])]
    p = [rng.uniform(a[0], a[-1], n) for a in ax]
    p[2][:5] = 30.0                                                      # outside the third parameter axis: NaN rows
    keys = ["pa", "pb", "pc", "pd"]
    fac = (1.0 + 0.3 * (ax[0] - 1.0))[:, None, None, None] * (1.0 + 0.05 * ax[1] ** 2)[None, :, None, None] * \
        (ax[2] / 10.0)[None, None, :, None] * (1.0 + 0.2 * ax[3])[None, None, None, :]
    extra = np.stack(p, 1)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))
    zax, Max, rax, T = syn.pressure_table(3, 8, nr)
    TN = T[..., None, None, None, None] * fac[None, None, None]
    ref, ptot = oracle_paint_nd(cosmo, ra, dec, M, z, (zax, Max, rax, *ax), TN, nside, eps, extra)
    model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax)))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False,
                                   variant=variant)
        got = R.process()
    assert R.last_stats["pixel_updates"] == ptot and np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, 1e-9, what=f"paint, 4 extra axes, variant {variant}, nr {nr}")
    if nr == 60:
        from util import oracle_baryonify
        zd, Md, rd, d = syn.displacement_table(3, 8, nr)
        dN = d[..., None, None, None, None] * fac[None, None, None]
        m_in = syn.mass_map(nside)
        refb = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd, *ax), dN, nside, eps, 20, m_in, extra=extra)
        bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, dN, cosmo, epsilon_max=20, other_params=dict(zip(keys, ax)))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gotb = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), eps, bm, verbose=False, variant=variant).process()
        assert np.isclose(gotb.sum(), m_in.sum())
        assert_maps_close(gotb, refb, 1e-5, floor=1e-9, what=f"baryonify, 4 extra axes, variant {variant}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["usual", "batched", "all_outside", "one_cell"])
def test_nd_rows_grouped_by_cell_equal_the_rows_blended_per_halo(cosmo, case, monkeypatch):
    """run_shell_nd groups the halos by table cell and blends eight halos of a cell per wavefront (nd_rows_blocked_kernel; the same
    corner rows and weights as nd_rows_kernel, summed in another order): the painted map must be the map of BFG_ND_ROWS=plain -- every halo's rows blended by
    itself -- up to the order of the tile kernel's additions, with the same P_tot, non-zero set and warnings; also in batches of
    halos, with every halo outside the hull of a parameter axis (nothing to sort), and with all halos in ONE cell (one hot counter)."""
    import warnings
    nside, n, eps = 256, 20_000, 8.0
    ra, dec, M, z = syn.catalog(n, seed=77, logM=(13.0, 15.3))
    rng = np.random.default_rng(8)
    ax = [np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0]), np.array([-0.5, 0.5, 1.5]), np.array([1.0, 2.0, 4.0])]
    p = [rng.uniform(a[0], a[-1], n) for a in ax]
    if case == "all_outside":
        p[1][:] = 3.0
    if case == "one_cell":
        p = [rng.uniform(a[0], a[1], n) for a in ax]
        M[:] = 10 ** rng.uniform(14.0, 14.01, n); z[:] = rng.uniform(0.30, 0.301, n)
    if case == "batched":
        monkeypatch.setenv("BFG_ND_ROW_BYTES", str(8 * 60 * 777))
    keys = ["pa", "pb", "pc", "pd", "pe"]
    fac = 1.0
    for k, a in enumerate(ax):
        sh = [1] * len(ax); sh[k] = a.size
        fac = fac * (1.0 + 0.1 * (k + 1) * (a - a[0]) / (a[-1] - a[0])).reshape(sh)
    zax, Max, rax, T = syn.pressure_table(3, 8, 60)
    TN = T.reshape(T.shape + (1,) * len(ax)) * fac[None, None, None]
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))
    model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax)))

    def paint():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False)
            out = R.process()
        return out, R.last_stats, sorted(str(x.message)[:40] for x in w)
    got, st, wg = paint()
    monkeypatch.setenv("BFG_ND_ROWS", "plain")
    ref, st_ref, wr = paint()
    assert st["pixel_updates"] == st_ref["pixel_updates"] and st["halos_out_of_table"] == st_ref["halos_out_of_table"]
    assert wg == wr
    assert np.array_equal(got != 0, ref != 0)
    if case == "all_outside":
        assert st["halos_out_of_table"] == n and not got.any()
    else:
        assert got.any()
        assert_maps_close(got, ref, 1e-12, what=f"rows grouped by cell vs per halo ({case})")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["paint3", "offsets1"])
def test_tables_with_both_forms_give_the_same_map_on_rows_and_direct(cosmo, kind, monkeypatch):
    """A paint table with three p_keys axes / a displacement table with one keeps both forms (bfg_table_create) and a shell call takes
    the per-halo rows where a table cell holds eight halos or more on average (nd_rows_pay), else the kernels' own corner blend:
    the same catalog through both (BFG_ND_FROM_DIM=7 at table creation: no N-dimensional form, so direct) must give the same map."""
    import warnings
    nside, n, eps = 128, 3000, 8.0
    ra, dec, M, z = syn.catalog(n, seed=5, logM=(13.0, 15.3))
    rng = np.random.default_rng(2)
    if kind == "paint3":
        ax = [np.array([0.6, 1.0, 1.5]), np.array([-1.5, 0.0, 2.5]), np.array([5.0, 25.0])]
    else:
        ax = [np.array([0.6, 1.0, 1.5])]
    keys = ["pa", "pb", "pc"][: len(ax)]
    p = [rng.uniform(a[0], a[-1], n) for a in ax]
    fac = 1.0
    for k, a in enumerate(ax):
        sh = [1] * len(ax); sh[k] = a.size
        fac = fac * (1.0 + 0.1 * (k + 1) * (a - a[0]) / (a[-1] - a[0])).reshape(sh)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo, **dict(zip(keys, p)))

    def run():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == "paint3":
                zax, Max, rax, T = syn.pressure_table(3, 8, 60)
                TN = T.reshape(T.shape + (1,) * len(ax)) * fac[None, None, None]
                model = bfg.ParamTabulatedProfile.from_arrays(zax, Max, rax, TN, other_params=dict(zip(keys, ax)))
                R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), eps, model, verbose=False)
            else:
                zd, Md, rd, d = syn.displacement_table(3, 8, 60)
                dN = d.reshape(d.shape + (1,) * len(ax)) * fac[None, None, None]
                bm = bfg.Baryonification2D.from_arrays(zd, Md, rd, dN, cosmo, epsilon_max=20, other_params=dict(zip(keys, ax)))
                R = bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=syn.mass_map(nside), cosmo=cosmo), eps, bm, verbose=False)
            return R.process(), R.last_stats
    got, st = run()                                   # 3000 halos over 2 * 7 * (2 * 2 * 1 | 2) cells: rows
    monkeypatch.setenv("BFG_ND_FROM_DIM", "7")
    ref, st_ref = run()                               # a new table object without the N-dimensional form: direct
    assert st["pixel_updates"] == st_ref["pixel_updates"] > 0
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, 1e-11, what=f"{kind}: rows vs direct")
