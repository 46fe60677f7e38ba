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
