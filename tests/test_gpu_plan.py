"""
One plan, K models (BFG_SHELL_REUSE_PLAN, include/bfg_mi355.h): the reference's workflow paints several models over ONE catalog
(examples/05_Paint_tSZ_shell.ipynb:303-324, utils/Parallelize.py:92-113).  A shell call whose catalog tensor, geometry and table
AXES are those of the context's previous call reuses its per-halo records and pair lists and runs the tile kernels only.  Checked
here: the short cut is taken when it may be and only then, and a call that takes it gives the map of a call that does not (same
kernels on the same records: equal to the rounding of the LDS atomics' order, identical non-zero sets, identical counters).
"""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.engine import get_context
from util import assert_maps_close

TIGHT = 1e-11


def _models(n, kind="paint", cosmo=None, shape=(10, 30, 100)):
    out = []
    for k in range(n):
        if kind == "paint":
            zax, Max, rax, T = syn.pressure_table(*shape)
            out.append(bfg.TabulatedProfile.from_arrays(zax, Max, rax, T * (1.0 + 0.5 * k) * (1.0 + 0.1 * k * np.tanh(np.exp(rax)))[None, None, :]))
        else:
            zax, Max, rax, d = syn.displacement_table(*shape)
            out.append(bfg.Baryonification2D.from_arrays(zax, Max, rax, d * (1.0 + 0.3 * k), cosmo, epsilon_max=20))
    return out


def _separately(monkeypatch, fn):
    """fn() with the short cut switched off; afterwards the context has no catalog on record, so that the next call plans afresh
    (the last call of fn() left a perfectly good plan behind)"""
    monkeypatch.setenv("BFG_PLAN_REUSE", "0")
    try:
        return fn()
    finally:
        monkeypatch.delenv("BFG_PLAN_REUSE")
        get_context()._plan_cat = None


@pytest.mark.parametrize("case", ["sky", "crowded", "steep"])
def test_five_paint_models_over_one_catalog_share_one_plan(cosmo, case, monkeypatch):
    nside = 256
    if case == "crowded":                 # a compact patch: tiles beyond their fixed slots, shared work items, overflow lists
        rng = np.random.default_rng(3)
        n = 30000
        ra, dec = rng.uniform(10.0, 14.0, n), rng.uniform(-2.0, 2.0, n)
        M, z = 10 ** rng.uniform(13.0, 14.5, n), rng.uniform(0.3, 0.5, n)
    else:
        ra, dec, M, z = syn.catalog(20000, seed=77, steep=(case == "steep"))
        M[5], z[7] = 5e16, 0.001            # outside the table hull: counted, left to the scatter kernel
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    models = _models(5)
    ctx = get_context()

    def run():
        maps, stats = [], []
        for m in models:
            R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10, m, verbose=False)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                maps.append(R.process())
            stats.append(dict(R.last_stats))
        return maps, stats
    ref, ref_stats = _separately(monkeypatch, run)
    n0 = ctx.plan_reuses()
    got, got_stats = run()
    assert ctx.plan_reuses() - n0 == 4                     # the first model plans, the other four ride on it
    for k in range(5):
        assert np.array_equal(got[k] != 0, ref[k] != 0)
        assert_maps_close(got[k], ref[k], TIGHT, what=f"{case}: model {k} on a reused plan")
        assert got_stats[k] == ref_stats[k], (k, got_stats[k], ref_stats[k])
    assert not np.allclose(got[0], got[1])                 # (the models do differ)


def test_baryonify_models_share_one_plan_and_windows_are_rebuilt_per_table(cosmo, monkeypatch):
    nside = 128
    ra, dec, M, z = syn.catalog(4000, seed=12, logM=(13.0, 15.3))
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    m_in = syn.mass_map(nside)
    models = _models(3, "bary", cosmo)
    ctx = get_context()

    def run():
        out = []
        for m in models:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out.append(bfg.BaryonifyShell(Cat, bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo), 10, m, verbose=False).process())
        return out
    ref = _separately(monkeypatch, run)
    n0 = ctx.plan_reuses()
    got = run()
    assert ctx.plan_reuses() - n0 == 2
    for k in range(3):
        assert np.isclose(got[k].sum(), m_in.sum())
        assert_maps_close(got[k], ref[k], 1e-9, floor=1e-12, what=f"baryonify model {k} on a reused plan")
    assert not np.allclose(got[0], got[2])


def test_plan_is_not_reused_when_anything_it_was_built_from_differs(cosmo):
    nside = 128
    ra, dec, M, z = syn.catalog(3000, seed=5)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    Cat2 = bfg.HaloLightConeCatalog(ra, dec, M * 1.5, z, cosmo)
    base, other = _models(2)
    zax, Max, rax, T = syn.pressure_table(8, 30, 100)       # another redshift grid
    regrid = bfg.TabulatedProfile.from_arrays(zax, Max, rax, T)
    ctx = get_context()

    def paint(cat, model, eps=10, ns=nside, ips=False):
        R = bfg.PaintProfilesShell(cat, bfg.LightconeShell(map=np.zeros(12 * ns * ns), cosmo=cosmo), eps, model, verbose=False,
                                   include_pixel_size=ips)
        return R.process()

    def reused(fn):
        n0 = ctx.plan_reuses()
        fn()
        return ctx.plan_reuses() - n0
    ctx._plan_cat = None
    assert reused(lambda: paint(Cat, base)) == 0
    assert reused(lambda: paint(Cat, other)) == 1           # same grid, other values
    assert reused(lambda: paint(Cat, regrid)) == 0          # other axes
    assert reused(lambda: paint(Cat, regrid)) == 1          # ... which now are the plan's
    assert reused(lambda: paint(Cat2, regrid)) == 0         # another catalog
    assert reused(lambda: paint(Cat2, regrid, eps=8)) == 0  # another cut-out radius
    assert reused(lambda: paint(Cat2, regrid, eps=8)) == 1
    assert reused(lambda: paint(Cat2, regrid, eps=8, ns=64)) == 0
    assert reused(lambda: paint(Cat2, regrid, eps=8, ns=64, ips=True)) == 0
    # an in-place edit of the catalog drops its device copy (the array is read-only while one exists): a new tensor, a new plan
    Cat2.invalidate()
    Cat2.cat["M"][:] *= 2.0
    ref = paint(bfg.HaloLightConeCatalog(ra, dec, M * 3.0, z, cosmo), regrid, eps=8, ns=64, ips=True)
    n0 = ctx.plan_reuses()
    got = paint(Cat2, regrid, eps=8, ns=64, ips=True)
    assert ctx.plan_reuses() == n0
    assert_maps_close(got, ref, TIGHT, what="edited catalog")


def test_list_of_runners_through_the_parallel_wrappers_reuses_the_plan(cosmo, monkeypatch):
    """SimpleParallel / SplitJoinParallel over the notebooks' list of runners (one catalog, five models): sliced calls included"""
    nside = 128
    ra, dec, M, z = syn.catalog(5000, seed=9)
    Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
    models = _models(4)
    ctx = get_context()

    def runners():
        return [bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), 10, m, verbose=False)
                for m in models]
    for wrap in (lambda rs: bfg.SimpleParallel(rs), lambda rs: bfg.SplitJoinParallel(rs, slices=3)):
        ref = _separately(monkeypatch, lambda: wrap(runners()).process())
        n0 = ctx.plan_reuses()
        got = wrap(runners()).process()
        assert ctx.plan_reuses() - n0 == 3
        for k in range(4):
            assert np.array_equal(got[k] != 0, ref[k] != 0)
            assert_maps_close(got[k], ref[k], TIGHT, what=f"list of runners, model {k}")
