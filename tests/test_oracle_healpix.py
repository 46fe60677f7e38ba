"""
Pins the oracle's HEALPix RING restatement (oracle/bfg_oracle.c) without healpy:
  * healpy doc-string examples (healpy.pixelfunc.pix2vec / pix2ang / get_interp_weights /
    nside2pixarea / nside2resol doc-strings, recalled -- healpy is not installed here);
  * analytic known answers: Npix, base-pixel centres, ring z values,
    sum of bilinear weights, equal-area, brute-force disc membership, round trips.
"""
import numpy as np
import pytest

from oracle import oracle as o


def test_healpy_docstring_examples():
    x, y, z = o.pix2vec(16, 1504)
    assert (x, y, z) == (0.99879545620517241, 0.049067674327418015, 0.0)
    x, y, z = o.pix2vec(16, np.array([1440, 427]))
    np.testing.assert_allclose(x, [0.99913157, 0.5000534], atol=5e-9)
    np.testing.assert_allclose(y, [0.0, 0.5000534], atol=5e-9)
    np.testing.assert_allclose(z, [0.04166667, 0.70703125], atol=5e-9)
    # pix2ang(16, [1440, 427, 1520, 0, 3068]) (the recalled phi of the last entry is not trusted; theta is)
    th, ph = o.vec2ang(np.stack(o.pix2vec(16, np.array([1440, 427, 1520, 0, 3068])), 1))
    np.testing.assert_allclose(th, [1.52911759, 0.78550497, 1.57079633, 0.05103658, 3.09055608], atol=5e-9)
    np.testing.assert_allclose(ph[:4], [0.0, 0.78539816, 1.61988371, 0.78539816], atol=5e-9)
    p, w = o.get_interp_weights(1, 0.0, 0.0)
    assert list(p) == [1, 2, 3, 0] and list(w) == [0.25] * 4
    p, w = o.get_interp_weights(1, np.array([0, np.pi / 2]), np.array([0.0, 0.0]))
    assert p.tolist() == [[1, 4], [2, 5], [3, 11], [0, 8]]
    assert w.tolist() == [[0.25, 1.0], [0.25, 0.0], [0.25, 0.0], [0.25, 0.0]]
    assert o.nside2pixarea(128, degrees=True) == 0.2098234113027917
    assert o.nside2pixarea(256) == 1.5978966540475428e-05
    assert o.nside2resol(128, arcmin=True) == pytest.approx(27.483891294539248, rel=1e-14)


def test_base_pixels_and_rings():
    assert o.nside2npix(1024) == 12582912 and o.npix2nside(12582912) == 1024
    with pytest.raises(ValueError):
        o.npix2nside(100)
    x, y, z = o.pix2vec(1, np.arange(12))
    np.testing.assert_allclose(z, [2 / 3] * 4 + [0] * 4 + [-2 / 3] * 4, atol=1e-15)
    phi = np.mod(np.arctan2(y, x), 2 * np.pi)
    np.testing.assert_allclose(phi[:4], np.pi / 4 + np.arange(4) * np.pi / 2, atol=1e-15)
    np.testing.assert_allclose(phi[4:8], np.arange(4) * np.pi / 2, atol=1e-15)
    for nside in (2, 8, 64):
        npix = o.nside2npix(nside)
        x, y, z = o.pix2vec(nside, np.arange(npix))
        np.testing.assert_allclose(x * x + y * y + z * z, 1, atol=1e-14)
        zr = np.unique(np.round(z, 13))
        assert zr.size == 4 * nside - 1
        i = np.arange(1, 4 * nside)
        zt = np.where(i < nside, 1 - i ** 2 / (3 * nside ** 2),
                      np.where(i <= 3 * nside, (2 * nside - i) * 2 / (3 * nside),
                               (4 * nside - i) ** 2 / (3 * nside ** 2) - 1))
        np.testing.assert_allclose(np.sort(zt), zr, atol=1e-12)
        # ascending pixel index = descending z, ascending phi within a ring
        assert np.all(np.diff(z) <= 1e-15)
        # equal-area: mean vector ~ 0
        assert abs(x.sum()) < 1e-9 and abs(y.sum()) < 1e-9 and abs(z.sum()) < 1e-9


def test_ang2vec_vec2ang_roundtrip():
    rng = np.random.default_rng(0)
    lon = rng.uniform(0, 360, 1000)
    lat = np.degrees(np.arcsin(rng.uniform(-1, 1, 1000)))
    v = o.ang2vec(lon, lat, lonlat=True)
    np.testing.assert_allclose(v[:, 2], np.sin(np.radians(lat)), atol=1e-15)
    lon2, lat2 = o.vec2ang(v * 3.7, lonlat=True)
    np.testing.assert_allclose(lon2, lon, atol=1e-10)
    np.testing.assert_allclose(lat2, lat, atol=1e-7)  # acos near the poles
    for nside in (4, 32):
        p = np.arange(o.nside2npix(nside))
        v = np.stack(o.pix2vec(nside, p), 1)
        lon, lat = o.vec2ang(v, lonlat=True)
        pp, ww = o.get_interp_weights(nside, lon, lat, lonlat=True)
        np.testing.assert_allclose(ww.sum(0), 1, atol=1e-12)
        # a pixel centre interpolates (almost) entirely onto itself
        self_w = np.where(pp == p[None, :], ww, 0).sum(0)
        np.testing.assert_allclose(self_w, 1, atol=1e-9)


@pytest.mark.parametrize("nside", [1, 2, 4, 16, 64])
def test_query_disc_bruteforce(nside):
    rng = np.random.default_rng(nside)
    npix = o.nside2npix(nside)
    V = np.stack(o.pix2vec(nside, np.arange(npix)), 1)
    for t in range(150):
        lon = rng.uniform(0, 360)
        lat = np.degrees(np.arcsin(rng.uniform(-1, 1)))
        if t < 6:
            lat = (89.9 - 0.01 * t) * (1 if t % 2 else -1)
        if 6 <= t < 10:
            lon = [0.0, 1e-9, 359.999999, 180.0][t - 6]
        r = rng.uniform(0, 1.0 if t % 2 else 0.1) * (3.3 if t % 17 == 0 else 1)
        v = o.ang2vec(lon, lat, lonlat=True)
        q = o.query_disc(nside, v, r)
        assert np.all(np.diff(q) > 0)
        cosang = V @ v
        bf = np.where(cosang > np.cos(r))[0] if r < np.pi else np.arange(npix)
        diff = np.setxor1d(q, bf)
        # only pixels within rounding of the rim may differ
        assert np.all(np.abs(np.arccos(np.clip(cosang[diff], -1, 1)) - r) < 1e-9)


def test_interp_weights_properties():
    rng = np.random.default_rng(3)
    for nside in (1, 2, 16, 128):
        lon = np.concatenate([rng.uniform(0, 360, 2000), [0, 0, 359.9999999, 45, 45]])
        lat = np.concatenate([np.degrees(np.arcsin(rng.uniform(-1, 1, 2000))), [90, -90, 0, 89.99999, -89.99999]])
        p, w = o.get_interp_weights(nside, lon, lat, lonlat=True)
        assert p.shape == (4, lon.size) and p.min() >= 0 and p.max() < o.nside2npix(nside)
        np.testing.assert_allclose(w.sum(0), 1, atol=1e-12)
        assert w.min() > -1e-12
        # interpolating the smooth field f = z reproduces z to O(pixel^2)
        z = np.stack(o.pix2vec(nside, p.ravel()), 1)[:, 2].reshape(4, -1)
        if nside >= 16:
            np.testing.assert_allclose((w * z).sum(0), np.sin(np.radians(lat)), atol=3.0 / nside ** 2 + 1e-12)


# ---------------------------------------------------------------- pins at the resolutions the workloads run at
@pytest.mark.parametrize("nside", [128, 1024, 2048])
def test_pix2vec_equals_the_closed_form_ring_formulae(nside):
    """every pixel centre of the oracle == the closed-form ring formulae evaluated independently (tests/closed_form.py)"""
    from closed_form import ring_pixel_vectors
    v = ring_pixel_vectors(nside)
    step = 1 if nside <= 1024 else 3
    idx = np.arange(0, 12 * nside * nside, step)
    ref = np.stack(o.pix2vec(nside, idx), axis=1)
    np.testing.assert_allclose(ref, v[idx], rtol=0, atol=2e-14)


@pytest.mark.parametrize("nside", [128, 512, 1024, 2048])
def test_interp_weights_reproduce_linear_functions(nside):
    """get_interp_weights is the bilinear ring interpolation: between two rings it reproduces the colatitude exactly
    (sum_k w_k theta_k = theta) and inside each ring the longitude (sum over the ring's two pixels of w_k phi_k / their weight
    = phi, away from the phi = 0 seam), to 1e-12 -- at the resolutions of the BASELINE workloads"""
    from closed_form import ring_pixel_z_phi, ring_theta
    rng = np.random.default_rng(nside)
    n = 20000
    th_r = ring_theta(nside)
    theta = rng.uniform(th_r[0] * 1.0001, th_r[-1] * 0.9999, n)            # between the first and the last ring
    phi = rng.uniform(0.0, 2 * np.pi, n)
    pix, w = o.get_interp_weights(nside, theta, phi)
    np.testing.assert_allclose(w.sum(axis=0), 1.0, rtol=0, atol=1e-13)
    assert np.all(w >= -1e-15)
    z, ph = ring_pixel_z_phi(nside)
    tk = np.arccos(z[pix])                                                  # colatitudes of the four pixels
    np.testing.assert_allclose((w * tk).sum(axis=0), theta, rtol=0, atol=1e-12)
    # the two pixels of each ring: pix[0:2] upper ring, pix[2:4] lower ring
    for a, b in ((0, 1), (2, 3)):
        pa, pb = ph[pix[a]], ph[pix[b]]
        pb = np.where(pb < pa - np.pi, pb + 2 * np.pi, pb)                  # the pair straddles phi = 2 pi
        pa2 = np.where(pa > pb + np.pi, pa - 2 * np.pi, pa)
        wsum = w[a] + w[b]
        ok = wsum > 1e-9
        got = (w[a] * pa2 + w[b] * pb)[ok] / wsum[ok]
        want = phi[ok]
        d = np.mod(got - want + np.pi, 2 * np.pi) - np.pi                    # compare on the circle
        assert np.max(np.abs(d)) < 1e-12, np.max(np.abs(d))


@pytest.mark.parametrize("nside", [1024, 2048])
def test_query_disc_equals_brute_force_at_workload_resolution(nside):
    """disc membership at NSIDE 1024 / 2048 (the existing brute-force test stops at 64): for 60 discs incl. the poles and the
    seam, query_disc == { p : n_p . n_j > cos(theta) } over the pixels of a generous ring band around the disc, centres from
    the closed-form formulae; pixels within 1e-12 rad of the rim are exempt"""
    from closed_form import ring_pixel_vectors
    v = ring_pixel_vectors(nside)
    rng = np.random.default_rng(5 + nside)
    res = np.sqrt(4 * np.pi / (12 * nside * nside))
    cases = []
    for k in range(60):
        th = np.arccos(rng.uniform(-1, 1))
        ph = rng.uniform(0, 2 * np.pi)
        if k < 6:
            th = rng.uniform(0, 3 * res)                 # around the north pole
        elif k < 12:
            th = np.pi - rng.uniform(0, 3 * res)         # south pole
        elif k < 20:
            ph = rng.normal(0, 2 * res) % (2 * np.pi)    # the phi = 0 seam
        cases.append((th, ph, res * 10 ** rng.uniform(-0.3, 1.7)))            # radii 0.5 .. 50 pixels
    for th, ph, rad in cases:
        c = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        got = o.query_disc(nside, c, rad)
        zlo, zhi = np.cos(min(np.pi, th + rad + 3 * res)), np.cos(max(0.0, th - rad - 3 * res))
        band = np.flatnonzero((v[:, 2] >= zlo) & (v[:, 2] <= zhi))
        dot = v[band] @ c
        margin = np.abs(np.arccos(np.clip(dot, -1, 1)) - rad)
        inside = band[dot > np.cos(rad)]
        rim = band[margin < 1e-12]
        assert np.array_equal(np.setdiff1d(got, rim), np.setdiff1d(inside, rim)), (th, ph, rad)
        assert np.all(np.diff(got) > 0)


def test_callable_model_loops_equal_the_table_loops(cosmo):
    """the python restatement of the reference loops for models that are callables (oracle.paint_shell_callable /
    baryonify_offsets_callable -- what tests/test_gpu_callable.py checks the GPU against) run with a callable that IS the
    tabulated read-out must reproduce the C oracle's table loops, which the reference's golden vectors pin"""
    from baryonforge_amd import synthetic as syn
    from oracle import oracle as orc
    from util import oracle_baryonify, oracle_paint
    nside, n, eps = 64, 150, 8.0
    ra, dec, M, z = syn.catalog(n, seed=12, logM=(12.5, 15.3))
    zax, Max, rax, T = syn.pressure_table()
    ref, ptot = oracle_paint(cosmo, ra, dec, M, z, (zax, Max, rax), T, nside, eps, include_pixel_size=True)

    def projected(r, Mj, aj):                               # Tabulate.py:305-316: exp of the read-out on (ln(1+z), ln M, ln r)
        with np.errstate(all="ignore"):
            pts = np.stack([np.full(r.size, np.log(1 / aj)), np.full(r.size, np.log(Mj)), np.log(r)], axis=1)
            return np.exp(orc.interp_linear((zax, Max, rax), np.log(T), pts))
    got, p2 = orc.paint_shell_callable(cosmo, nside, ra, dec, M, z, eps, projected, include_pixel_size=True)
    assert p2 == ptot and np.array_equal(got != 0, ref != 0)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=0)

    zd, Md, rd, d = syn.displacement_table()
    a_all, R_all, _ = orc.halo_scalars(cosmo, M, z)

    def displacement(r, Mj, aj):                            # BaryonCorrection.py:396-411: linear table, zero beyond eps_model R
        Rm = float(orc.get_radius(cosmo, Mj, aj)) / aj
        pts = np.stack([np.full(r.size, np.log(1 / aj)), np.full(r.size, np.log(Mj)), np.log(r)], axis=1)
        with np.errstate(all="ignore"):
            out = orc.interp_linear((zd, Md, rd), d, pts)
        return np.where(r < 20.0 * Rm, out, 0.0)
    refo, ptot_b = oracle_baryonify(cosmo, ra, dec, M, z, (zd, Md, rd), d, nside, eps, 20, None, offsets_only=True)
    goto, p3 = orc.baryonify_offsets_callable(cosmo, nside, ra, dec, M, z, eps, displacement)
    assert p3 == ptot_b
    np.testing.assert_allclose(goto, np.asarray(refo).reshape(-1, 3), rtol=1e-9, atol=1e-18)
