"""
LightconeShell(path="*.fits") (BaryonForge/utils/io.py:346-347 hp.read_map(path)) without healpy: the numpy reader of
baryonforge_amd/utils/fits.py against files written here byte by byte (healpy's own layout: 1024 values per row; one value per
row; two columns; a partial-sky file with explicit indices; TSCAL / TZERO; BAD_DATA), NESTED -> RING reordering pinned by an
independent route (pixel-centre angles from the closed-form ring formulae -> ang2pix_nest), and the container end to end.
"""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import baryonforge_amd as bfg                                   # noqa: E402
from baryonforge_amd import sharding, synthetic as syn          # noqa: E402
from baryonforge_amd.utils import fits                          # noqa: E402
from closed_form import ring_pixel_z_phi                        # noqa: E402


def _hdu(cards):
    txt = "".join(f"{c:<80}"[:80] for c in cards) + f"{'END':<80}"
    return (txt + " " * (-len(txt) % 2880)).encode("ascii")


def _file(path, ext_cards, data):
    with open(path, "wb") as f:
        f.write(_hdu(["SIMPLE  =                    T", "BITPIX  =                    8", "NAXIS   =                    0",
                      "EXTEND  =                    T"]))
        f.write(_hdu(ext_cards))
        f.write(data + b"\0" * (-len(data) % 2880))


@pytest.mark.parametrize("nside", [1, 2, 4, 16, 64])
def test_ring2nest_equals_the_angle_route(nside):
    """integer ring -> nest (healpix_cxx's ring2xyf / xyf2nest, restated) == ang2pix_nest of the closed-form pixel centres: every
    pixel, a bijection"""
    npix = 12 * nside * nside
    z, phi = ring_pixel_z_phi(nside)
    via_angles = sharding.ang2pix_nest(nside, np.degrees(phi), np.degrees(np.arcsin(z)))
    got = fits.ring2nest(nside, np.arange(npix))
    assert np.array_equal(got, via_angles)
    assert np.array_equal(np.sort(got), np.arange(npix))
    if nside == 1:
        assert np.array_equal(got, np.arange(12))                # base pixels: the two schemes coincide
    with pytest.raises(ValueError):
        fits.ring2nest(3, np.arange(4))


@pytest.mark.parametrize("dtype,nest", [(np.float64, False), (np.float32, False), (np.float64, True), (np.float32, True)])
def test_write_read_round_trip_and_lightcone_shell(tmp_path, dtype, nest):
    nside = 32
    npix = 12 * nside * nside
    ring_map = np.random.default_rng(5).uniform(0, 10, npix).astype(dtype)
    stored = ring_map
    if nest:                                                     # the file holds the map in NESTED order
        stored = np.empty_like(ring_map)
        stored[fits.ring2nest(nside, np.arange(npix))] = ring_map
    path = str(tmp_path / "map.fits")
    fits.write_healpix_map(path, stored, nest=nest)
    got = fits.read_healpix_map(path, use_healpy=False)
    assert got.dtype == np.dtype(dtype) and got.dtype.isnative and got.flags["C_CONTIGUOUS"]
    assert np.array_equal(got, ring_map)                         # RING order out, whatever the file's ordering (read_map's default)
    assert np.array_equal(fits.read_healpix_map(path, nest=True, use_healpy=False), stored if nest else
                          ring_map[np.argsort(fits.ring2nest(nside, np.arange(npix)))])
    Shell = bfg.LightconeShell(path=path, cosmo=dict(syn.COSMO), redshift=0.3)      # io.py:346-347
    assert Shell.NSIDE == nside and np.array_equal(Shell.map, ring_map) and Shell.redshift == 0.3
    assert Shell.data is Shell.map


def test_reader_handles_the_layouts_found_in_the_wild(tmp_path):
    nside, npix = 4, 192
    m = np.arange(npix, dtype=np.float64) * 0.5 - 3
    # (1) one value per row, TFORM 'D' without a repeat count, an extra column before the map is NOT there: plain single column
    p1 = str(tmp_path / "one_per_row.fits")
    _file(p1, ["XTENSION= 'BINTABLE'", "BITPIX  =                    8", "NAXIS   =                    2", "NAXIS1  =                    8",
               f"NAXIS2  = {npix:>20d}", "PCOUNT  =                    0", "GCOUNT  =                    1",
               "TFIELDS =                    1", "TTYPE1  = 'SIGNAL  '", "TFORM1  = 'D       '", "PIXTYPE = 'HEALPIX '",
               "ORDERING= 'RING    '           / Pixel ordering scheme, either RING or NESTED", f"NSIDE   = {nside:>20d}"],
          m.astype(">f8").tobytes())
    assert np.array_equal(fits.read_healpix_map(p1, use_healpy=False), m)
    # (2) two map columns (I, Q) of 64 float32 per row + TSCAL / TZERO on the second
    q = (np.arange(npix) % 7).astype(np.float32)
    rows = np.zeros(npix // 64, dtype=[("a", ">f4", (64,)), ("b", ">f4", (64,))])
    rows["a"] = m.astype(np.float32).reshape(-1, 64)
    rows["b"] = q.reshape(-1, 64)
    p2 = str(tmp_path / "two_columns.fits")
    _file(p2, ["XTENSION= 'BINTABLE'", "BITPIX  =                    8", "NAXIS   =                    2", "NAXIS1  =                  512",
               f"NAXIS2  = {npix // 64:>20d}", "PCOUNT  =                    0", "GCOUNT  =                    1",
               "TFIELDS =                    2", "TFORM1  = '64E     '", "TFORM2  = '64E     '", "TSCAL2  =                  2.0",
               "TZERO2  =                  1.5", "ORDERING= 'RING    '", f"NSIDE   = {nside:>20d}", "INDXSCHM= 'IMPLICIT'"],
          rows.tobytes())
    assert np.array_equal(fits.read_healpix_map(p2, use_healpy=False), m.astype(np.float32))
    assert np.allclose(fits.read_healpix_map(p2, field=1, use_healpy=False), q * 2.0 + 1.5)
    with pytest.raises(ValueError):
        fits.read_healpix_map(p2, field=2, use_healpy=False)
    # (3) partial sky: explicit pixel indices (J) + values (E); the rest is UNSEEN; BAD_DATA marks one value
    pix = np.array([5, 17, 100, 191], dtype=np.int64)
    val = np.array([1.5, -2.0, 7.0, -99.0], dtype=np.float32)
    rows = np.zeros(4, dtype=[("p", ">i4", (1,)), ("v", ">f4", (1,))])
    rows["p"][:, 0] = pix
    rows["v"][:, 0] = val
    p3 = str(tmp_path / "partial.fits")
    _file(p3, ["XTENSION= 'BINTABLE'", "BITPIX  =                    8", "NAXIS   =                    2", "NAXIS1  =                    8",
               "NAXIS2  =                    4", "PCOUNT  =                    0", "GCOUNT  =                    1",
               "TFIELDS =                    2", "TTYPE1  = 'PIXEL   '", "TFORM1  = '1J      '", "TFORM2  = '1E      '",
               "ORDERING= 'RING    '", f"NSIDE   = {nside:>20d}", "INDXSCHM= 'EXPLICIT'", "BAD_DATA=                -99.0"],
          rows.tobytes())
    got = fits.read_healpix_map(p3, use_healpy=False)
    assert got.size == npix and got.dtype == np.float32
    assert got[5] == 1.5 and got[17] == -2.0 and got[100] == 7.0 and got[191] == np.float32(fits.UNSEEN)
    assert np.count_nonzero(got == np.float32(fits.UNSEEN)) == npix - 3
    # errors: not a table, a missing HDU, a truncated file
    with pytest.raises(ValueError):
        fits.read_healpix_map(p1, hdu=0, use_healpy=False)
    with pytest.raises(ValueError):
        fits.read_healpix_map(p1, hdu=3, use_healpy=False)
    raw = open(p1, "rb").read()
    open(str(tmp_path / "cut.fits"), "wb").write(raw[:2880 * 2 + 100])
    with pytest.raises(ValueError):
        fits.read_healpix_map(str(tmp_path / "cut.fits"), use_healpy=False)


def test_lightcone_shell_still_reads_npy_and_validates(tmp_path):
    m = np.zeros(12 * 8 * 8)
    np.save(tmp_path / "m.npy", m)
    assert bfg.LightconeShell(path=str(tmp_path / "m.npy"), cosmo=dict(syn.COSMO)).NSIDE == 8
    with pytest.raises(ValueError):
        bfg.LightconeShell(cosmo=dict(syn.COSMO))
    with pytest.raises(ValueError):
        bfg.LightconeShell(map=m, cosmo={"Omega_m": 0.3})
