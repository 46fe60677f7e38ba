"""
GPU parity of tables with SEVERAL extra (p_keys) axes against fixtures produced by the reference's own glue
(tests/golden/make_golden.py pkeys -> pkeys.npz; VERDICT r5 "missing 2"):

  ParamTabulatedProfile._readout with two and four keys          Tabulate.py:598-650
  PaintProfilesShell.process with the keys as catalog columns   HealpixRunner.py:436, :456, :472
  BaryonifyShell.process, one / two keys, +- Rdelta_sampling      HealpixRunner.py:304, :322, :345; BaryonCorrection.py:374, :404-408
  BaryonifySnapshot.process, one key                            SnapshotRunner.py:208-258
  PaintProfilesGrid.process, one key                            Map2DRunner.py:716-812

Every shell case runs on BOTH device routes: the kernels reading the table's corners themselves (BFG_ND_FROM_DIM=7 keeps
tables of up to 6 dimensions in that form) and the per-halo-row route (nd_cell_kernel -> nd_rows_blocked_kernel -> tile
kernels; BFG_ND_FROM_DIM=4 forces it from one extra axis on; BFG_ND_ROWS=plain: nd_rows_kernel instead of the grouped rows).
Tolerance: the north star's 1e-5 relative on non-zero pixels (observed ~1e-12).
"""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import baryonforge_amd as bfg
from util import assert_maps_close

RTOL = 1e-5
BFLOOR = 1e-9
# (name, environment): which form of the table the shell kernels read
ROUTES = [("corners", {"BFG_ND_FROM_DIM": "7"}), ("rows-grouped", {"BFG_ND_FROM_DIM": "4"}),
          ("rows-plain", {"BFG_ND_FROM_DIM": "4", "BFG_ND_ROWS": "plain"}), ("default", {})]


def _keys(g, tag):
    return [str(k) for k in g[f"{tag}_keys"]]


def _route(monkeypatch, env, n_keys):
    if env.get("BFG_ND_FROM_DIM") == "7" and n_keys > 3:
        pytest.skip("the kernels read at most three extra axes themselves (BFG_MAX_DIM = 6): wider tables always take the rows")
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def test_multi_pkeys_readout_matches_reference(golden):
    """ParamTabulatedProfile.projected(cosmo, r, M, a, **{key: value}) with two and four keys -- bfg_table_eval (two keys: the
    6-D corner read-out on the device) or, past BFG_MAX_DIM, the host read-out the class falls back to -- against the reference's"""
    g = golden("pkeys.npz")
    for tag in ("k2", "k4"):
        keys = _keys(g, tag)
        prof = bfg.ParamTabulatedProfile.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_T2D"],
                                                     other_params={k: g[f"{tag}_ax_{k}"] for k in keys})
        for i in range(g[f"{tag}_ro_M"].size):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got = prof.projected(None, g[f"{tag}_ro_r"], g[f"{tag}_ro_M"][i], g[f"{tag}_ro_a"][i],
                                     **{k: g[f"{tag}_ro_{k}"][i] for k in keys})
            ref = g[f"{tag}_ro_projected"][i]
            assert np.array_equal(np.isnan(got), np.isnan(ref)), (tag, i)
            m = ~np.isnan(ref)
            np.testing.assert_allclose(got[m], ref[m], rtol=1e-11, atol=0)


@pytest.mark.parametrize("route", ROUTES, ids=[r[0] for r in ROUTES])
@pytest.mark.parametrize("tag", ["k2", "k4"])
def test_paint_shell_with_multi_pkeys_matches_reference(golden, cosmo, tag, route, monkeypatch):
    g = golden("pkeys.npz")
    keys = _keys(g, tag)
    _route(monkeypatch, route[1], len(keys))
    nside = int(g[f"{tag}_nside"])
    Cat = bfg.HaloLightConeCatalog(g[f"{tag}_ra"], g[f"{tag}_dec"], g[f"{tag}_M"], g[f"{tag}_z"], cosmo,
                                   **{k: g[f"{tag}_cat_{k}"] for k in keys})
    model = bfg.ParamTabulatedProfile.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_T2D"],
                                                  other_params={k: g[f"{tag}_ax_{k}"] for k in keys})
    R = bfg.PaintProfilesShell(Cat, bfg.LightconeShell(map=np.zeros(12 * nside * nside), cosmo=cosmo), float(g[f"{tag}_eps"]), model,
                               verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = R.process()
    ref = g[f"{tag}_map"]
    assert np.array_equal(got != 0, ref != 0)
    assert_maps_close(got, ref, RTOL, what=f"paint {tag} ({route[0]})")
    assert R.last_stats["halos_out_of_table"] >= 2            # the two halos put outside an extra axis, plus the (z, M) specials


@pytest.mark.parametrize("route", ROUTES, ids=[r[0] for r in ROUTES])
@pytest.mark.parametrize("tag", ["b1", "b1r", "b2", "b2r"])
def test_baryonify_shell_with_pkeys_matches_reference(golden, cosmo, tag, route, monkeypatch):
    g = golden("pkeys.npz")
    keys = _keys(g, tag)
    _route(monkeypatch, route[1], len(keys))
    Cat = bfg.HaloLightConeCatalog(g[f"{tag}_ra"], g[f"{tag}_dec"], g[f"{tag}_M"], g[f"{tag}_z"], cosmo,
                                   **{k: g[f"{tag}_cat_{k}"] for k in keys})
    model = bfg.Baryonification2D.from_arrays(g[f"{tag}_zax"], g[f"{tag}_Max"], g[f"{tag}_rax"], g[f"{tag}_d"], cosmo,
                                              epsilon_max=float(g[f"{tag}_eps_model"]), Rdelta_sampling=bool(g[f"{tag}_rdelta"]),
                                              other_params={k: g[f"{tag}_ax_{k}"] for k in keys})
    Shell = bfg.LightconeShell(map=g[f"{tag}_map_in"].copy(), cosmo=cosmo)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = bfg.BaryonifyShell(Cat, Shell, float(g[f"{tag}_eps"]), model, verbose=False).process()
    assert np.isclose(got.sum(), g[f"{tag}_map_in"].sum())
    assert_maps_close(got, g[f"{tag}_map_out"], RTOL, floor=BFLOOR, what=f"baryonify {tag} ({route[0]})")


@pytest.mark.parametrize("path", ["direct", "cell"])
def test_baryonify_snapshot_with_one_pkey_matches_reference(golden, cosmo, path, monkeypatch):
    monkeypatch.setenv("BFG_SNAPSHOT", path)
    g = golden("pkeys.npz")
    P, H, L = g["s1_P"], g["s1_H"], float(g["s1_L"])
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g["s1_hM"], float(g["s1_redshift"]), cosmo, z=H[:, 2], cdelta=g["s1_cat_cdelta"])
    Part = bfg.ParticleSnapshot(x=P[:, 0], y=P[:, 1], z=P[:, 2], M=np.ones(P.shape[0]), L=L, redshift=float(g["s1_redshift"]), cosmo=cosmo)
    model = bfg.Baryonification2D.from_arrays(g["s1_zax"], g["s1_Max"], g["s1_rax"], g["s1_d"], cosmo,
                                              epsilon_max=float(g["s1_eps_model"]), other_params={"cdelta": g["s1_ax_cdelta"]})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        new = bfg.BaryonifySnapshot(Cat, Part, epsilon_max=float(g["s1_eps"]), model=model, verbose=False).process()
    got = np.stack([new["x"], new["y"], new["z"]], axis=1)
    d = np.abs(got - g["s1_P_new"])
    assert np.minimum(d, L - d).max() <= 1e-9
    assert np.array_equal(np.any(g["s1_P_new"] != P, axis=1), np.any(got != P, axis=1))      # exactly the same particles move


@pytest.mark.parametrize("path", ["direct", "small"])
def test_grid_paint_with_one_pkey_matches_reference(golden, cosmo, path, monkeypatch):
    monkeypatch.setenv("BFG_GRID", path)
    g = golden("pkeys.npz")
    N, bins, H = int(g["g1_Npix"]), g["g1_bins"], g["g1_H"]
    Cat = bfg.HaloNDCatalog(H[:, 0], H[:, 1], g["g1_hM"], float(g["g1_redshift"]), cosmo, cdelta=g["g1_cat_cdelta"])
    model = bfg.ParamTabulatedProfile.from_arrays(g["g1_zax"], g["g1_Max"], g["g1_rax"], g["g1_T2D"],
                                                  other_params={"cdelta": g["g1_ax_cdelta"]})
    Map = bfg.GriddedMap(map=np.zeros((N, N)), redshift=float(g["g1_redshift"]), bins=bins, cosmo=cosmo)
    got = bfg.PaintProfilesGrid(Cat, Map, float(g["g1_eps"]), model, include_pixel_size=True, verbose=False).process()
    assert np.array_equal(got != 0, g["g1_map"] != 0)
    assert_maps_close(got, g["g1_map"], RTOL, what="grid paint with one p_keys axis")
    # the reference's BaryonifyGrid asserts isinstance(model, ParamTabulatedProfile) for a model with p_keys (Map2DRunner.py:477-480):
    # no displacement model passes -- same exception here
    assert str(g["g1_baryonify_grid_with_pkeys_raises"]) == "AssertionError"
    disp = bfg.Baryonification2D.from_arrays(g["b1_zax"], g["b1_Max"], g["b1_rax"], g["b1_d"], cosmo, epsilon_max=20,
                                             other_params={"cdelta": g["b1_ax_cdelta"]})
    with pytest.raises(AssertionError):
        bfg.BaryonifyGrid(Cat, bfg.GriddedMap(map=np.ones((N, N)), redshift=0.3, bins=bins, cosmo=cosmo), 4, disp, verbose=False).process()
