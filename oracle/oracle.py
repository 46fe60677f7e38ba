"""
CPU ORACLE -- python side (ctypes binding of oracle/bfg_oracle.c + numpy glue).

THIS MODULE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it, and only as the checker.  The product
package (baryonforge_amd) never imports it.

It restates, for the shell paint / baryonify hot path:
  * the reference runner loops      /root/reference/BaryonForge/Runners/HealpixRunner.py:252-373, :390-483
  * the table read-outs              utils/Tabulate.py:279-327, :598-650 ; Profiles/BaryonCorrection.py:331-419
  * the per-halo scalars             HealpixRunner.py:297-299 (D_A spline), :320 (R_delta)
and the third-party arithmetic underneath them (healpy -> HEALPix RING,
pyccl -> flat wCDM background), neither of which is installed here:
see the header of bfg_oracle.c for the parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbfg_oracle.so")

_i64 = C.c_int64
_dbl = C.c_double
_pd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_pi = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """gcc-build the C restatement next to its source."""
    src = os.path.join(_HERE, "bfg_oracle.c")
    if force or (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbfg_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_ring_above.restype = _i64
        L.orc_ring_above.argtypes = [_i64, _dbl]
        L.orc_ang2vec_lonlat.argtypes = [_i64, _pd, _pd, _pd]
        L.orc_pix2vec_ring.argtypes = [_i64, _i64, _pi, _pd]
        L.orc_query_disc_ring.restype = _i64
        L.orc_query_disc_ring.argtypes = [_i64, _pd, _dbl, C.c_void_p, _i64]
        L.orc_get_interp_weights_lonlat.argtypes = [_i64, _i64, _pd, _pd, _pi, _pd]
        L.orc_get_interp_weights_thetaphi.argtypes = [_i64, _i64, _pd, _pd, _pi, _pd]
        L.orc_vec2ang_lonlat.argtypes = [_i64, _pd, _pd, _pd]
        L.orc_interp_linear.argtypes = [C.c_int, _pi, _pd, _pd, _i64, _pd, _pd]
        L.orc_paint_shell.restype = _i64
        L.orc_paint_shell.argtypes = [_i64, _i64, _pd, _pd, _pd, _pd, _pd, _pd, C.c_void_p,
                                      C.c_int, _dbl, C.c_int, C.c_int, _pi, _pd, _pd, _pd]
        L.orc_paint_shell_splitjoin.restype = _i64
        L.orc_paint_shell_splitjoin.argtypes = [C.c_int, _i64, _i64, _pd, _pd, _pd, _pd, _pd, _pd,
                                                C.c_void_p, C.c_int, _dbl, C.c_int, _pi, _pd,
                                                _pd, _pd]
        L.orc_baryonify_offsets.restype = _i64
        L.orc_baryonify_offsets.argtypes = [_i64, _i64, _pd, _pd, _pd, _pd, _pd, _pd, _pd,
                                            C.c_void_p, C.c_int, _dbl, _dbl, C.c_int, C.c_int,
                                            _pi, _pd, _pd, _pd]
        L.orc_regrid_shell.argtypes = [_i64, _pd, _pd, _pd]
        L.orc_regrid_pixels_hpix.argtypes = [_pd, _i64, _pd, _pi, _pd]
        _lib = L
    return _lib


def _f(x):
    return np.ascontiguousarray(x, dtype=np.float64)


# --------------------------------------------------------------------------
# healpy-shaped wrappers (same call signatures as the healpy functions the
# reference calls, HealpixRunner.py:327,330,334,336,357,358,361,426; io.py:353)
# --------------------------------------------------------------------------
def nside2npix(nside):
    return 12 * int(nside) ** 2


def npix2nside(npix):
    nside = int(round(np.sqrt(npix / 12.0)))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


def nside2pixarea(nside, degrees=False):
    area = 4.0 * np.pi / nside2npix(nside)
    return area * (180.0 / np.pi) ** 2 if degrees else area


def nside2resol(nside, arcmin=False):
    r = np.sqrt(nside2pixarea(nside))
    return np.degrees(r) * 60 if arcmin else r


def ang2vec(theta, phi, lonlat=False):
    scalar = np.ndim(theta) == 0 and np.ndim(phi) == 0
    theta, phi = np.broadcast_arrays(_f(np.atleast_1d(theta)), _f(np.atleast_1d(phi)))
    if not lonlat:  # convert to the lon/lat the C routine expects, exactly invertible enough for tests
        lon, lat = np.degrees(phi), 90.0 - np.degrees(theta)
    else:
        lon, lat = theta, phi
    lon, lat = _f(lon), _f(lat)
    out = np.empty((lon.size, 3))
    lib().orc_ang2vec_lonlat(lon.size, lon, lat, out)
    return out[0] if scalar else out


def pix2vec(nside, ipix, nest=False):
    assert not nest
    scalar = np.ndim(ipix) == 0
    ipix = np.ascontiguousarray(np.atleast_1d(ipix), dtype=np.int64)
    out = np.empty((ipix.size, 3))
    lib().orc_pix2vec_ring(int(nside), ipix.size, ipix, out)
    if scalar:
        return out[0, 0], out[0, 1], out[0, 2]
    return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy()


def query_disc(nside, vec, radius, inclusive=False, fact=4, nest=False):
    assert (not inclusive) and (not nest)
    vec = _f(vec)
    n = lib().orc_query_disc_ring(int(nside), vec, float(radius), None, 0)
    out = np.empty(n, dtype=np.int64)
    if n:
        lib().orc_query_disc_ring(int(nside), vec, float(radius), out.ctypes.data, n)
    return out


def get_interp_weights(nside, theta, phi=None, nest=False, lonlat=False):
    assert not nest and phi is not None
    scalar = np.ndim(theta) == 0 and np.ndim(phi) == 0
    theta, phi = np.broadcast_arrays(_f(np.atleast_1d(theta)), _f(np.atleast_1d(phi)))
    theta, phi = _f(theta), _f(phi)
    pix = np.empty((theta.size, 4), dtype=np.int64)
    wgt = np.empty((theta.size, 4))
    if lonlat:
        lib().orc_get_interp_weights_lonlat(int(nside), theta.size, theta, phi, pix, wgt)
    else:
        lib().orc_get_interp_weights_thetaphi(int(nside), theta.size, theta, phi, pix, wgt)
    if scalar:
        return pix[0].copy(), wgt[0].copy()
    return pix.T.copy(), wgt.T.copy()  # healpy returns shape (4, n)


def vec2ang(vectors, lonlat=False):
    v = _f(vectors).reshape(-1, 3)
    lon = np.empty(v.shape[0])
    lat = np.empty(v.shape[0])
    lib().orc_vec2ang_lonlat(v.shape[0], v, lon, lat)
    if lonlat:
        return lon, lat
    return np.radians(90.0 - lat), np.radians(lon)


# --------------------------------------------------------------------------
# N-linear read-out (scipy RegularGridInterpolator, linear, nan fill)
# --------------------------------------------------------------------------
def _table_args(axes, values):
    values = _f(values)
    shape = np.ascontiguousarray(values.shape, dtype=np.int64)
    assert len(axes) == values.ndim and all(len(a) == s for a, s in zip(axes, values.shape))
    axes_concat = _f(np.concatenate([np.asarray(a, dtype=np.float64) for a in axes]))
    return values.ndim, shape, axes_concat, values


def interp_linear(axes, values, coords):
    """coords: [npts, ndim] -> [npts]"""
    ndim, shape, axes_concat, values = _table_args(axes, values)
    coords = _f(coords).reshape(-1, ndim)
    out = np.empty(coords.shape[0])
    lib().orc_interp_linear(ndim, shape, axes_concat, values, coords.shape[0], coords, out)
    return out


# --------------------------------------------------------------------------
# Background cosmology: what ccl.Cosmology(Omega_c, Omega_b, h, sigma8, n_s, w0)
# gives with its defaults (flat, T_CMB = 2.7255 K, N_eff = 3.044 massless,
# T_ncdm = 0.71611) for the three calls on the path:
#   ccl.angular_diameter_distance   HealpixRunner.py:299, :431
#   mass_def.get_radius             HealpixRunner.py:320, :454 ; BaryonCorrection.py:399
# --------------------------------------------------------------------------
CLIGHT = 299792458.0
GNEWT = 6.67430e-11
MPC_TO_METER = 3.085677581491367e22
GM_SUN = 1.3271244e20
SOLAR_MASS = GM_SUN / GNEWT
STBOLTZ = 5.670374419e-8
T_CMB = 2.7255
N_EFF = 3.044
T_NCDM = 0.71611
# 3 (100 km/s/Mpc)^2 / (8 pi G) in Msun / Mpc^3  (= 2.775366e11)
RHO_CRITICAL = (3.0 * 100.0 * 100.0) / (8.0 * np.pi * GNEWT) * (1000.0 * 1000.0 * MPC_TO_METER / SOLAR_MASS)


# Omega_nu,rel of the massless neutrinos = N_eff 7/8 x^4 Omega_gamma: "T_ncdm" x = 0.71611 (pyccl's T_nu = T_CMB T_ncdm, as recalled
# from its source; the default) or "4/11" x = (4/11)^(1/3) (instantaneous decoupling).  The choice moves D_A by 2.5e-7 at z = 0.5;
# a cosmology dict may carry it as cosmo["nu_rel"] (what a run against live pyccl must settle).
# The two chi differences live pyccl printed in the reference's example notebooks (tests/golden/pyccl_notebook_outputs.json) lie
# BETWEEN the two conventions: -1.4e-7 with "T_ncdm", +1.0e-7 with "4/11" (tests/test_oracle_golden.py).
NU_REL = "T_ncdm"


def _omegas(cosmo):
    h = cosmo["h"]
    rho_crit_si = RHO_CRITICAL * SOLAR_MASS / MPC_TO_METER ** 3 * h * h
    rho_g = 4.0 * STBOLTZ / CLIGHT ** 3 * T_CMB ** 4
    Om_g = rho_g / rho_crit_si
    conv = cosmo.get("nu_rel", NU_REL)
    assert conv in ("T_ncdm", "4/11")
    x = T_NCDM if conv == "T_ncdm" else (4.0 / 11.0) ** (1.0 / 3.0)
    Om_nu = N_EFF * 7.0 / 8.0 * x ** 4 * Om_g
    Om_m = cosmo["Omega_m"]
    Om_l = 1.0 - Om_m - Om_g - Om_nu
    return Om_m, Om_l, Om_g + Om_nu


def E2(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    Om_m, Om_l, Om_r = _omegas(cosmo)
    w0 = cosmo.get("w0", -1.0)
    return Om_m / a ** 3 + Om_l * a ** (-3.0 * (1.0 + w0)) + Om_r / a ** 4


_GL_X, _GL_W = np.polynomial.legendre.leggauss(96)


def comoving_radial_distance(cosmo, a):
    """chi(a) = c/H0 int_a^1 da' / (a'^2 E(a'))   [Mpc]"""
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    half = 0.5 * (1.0 - a)[:, None]
    mid = 0.5 * (1.0 + a)[:, None]
    x = mid + half * _GL_X[None, :]
    integrand = 1.0 / (x * x * np.sqrt(E2(cosmo, x)))
    return (CLIGHT / 1000.0 / 100.0 / cosmo["h"]) * np.sum(integrand * _GL_W[None, :], axis=1) * half[:, 0]


def angular_diameter_distance(cosmo, a):
    a = np.asarray(a, dtype=np.float64)
    return (np.atleast_1d(a) * comoving_radial_distance(cosmo, a)).reshape(a.shape)


def rho_x(cosmo, a, rho_type="critical"):
    """physical density in Msun/Mpc^3"""
    a = np.asarray(a, dtype=np.float64)
    h = cosmo["h"]
    if rho_type == "critical":
        return RHO_CRITICAL * h * h * E2(cosmo, a)
    if rho_type == "matter":
        return RHO_CRITICAL * h * h * cosmo["Omega_m"] / a ** 3
    raise ValueError(rho_type)


def get_radius(cosmo, M, a, Delta=200, rho_type="critical"):
    """ccl MassDef(Delta, rho_type).get_radius: physical Mpc"""
    return (np.asarray(M, dtype=np.float64) / (4.18879020479 * Delta * rho_x(cosmo, a, rho_type))) ** (1.0 / 3.0)


def halo_scalars(cosmo, M, z, Delta=200, rho_type="critical"):
    """a_j, R_j, D_j exactly as the runner loops form them
    (HealpixRunner.py:297-299, :317-321)."""
    from scipy import interpolate
    M = _f(M)
    z = _f(z)
    z_m = np.max(z)
    z_t = np.linspace(0, z_m + 0.1, 1000)
    D_a = interpolate.CubicSpline(z_t, angular_diameter_distance(cosmo, 1 / (1 + z_t)))
    a = 1 / (1 + z)
    R = get_radius(cosmo, M, a, Delta, rho_type)
    D = D_a(z)
    return _f(a), _f(R), _f(D)


# --------------------------------------------------------------------------
# Runner loops
# --------------------------------------------------------------------------
def _extra_ptr(extra, n):
    if extra is None or np.size(extra) == 0:
        return None, 0, None
    extra = _f(extra).reshape(n, -1)
    return extra.ctypes.data, extra.shape[1], extra


def paint_shell(nside, ra, dec, M, a, D, R, axes, values_log, eps_run,
                include_pixel_size=False, extra=None, njobs=None, out=None):
    """PaintProfilesShell.process loop (HealpixRunner.py:449-481) on a zero map.
    values_log is the table the interpolator holds, i.e. np.log(raw_input_2D)
    (Tabulate.py:271).  Returns (map, P_tot).  njobs: SplitJoinParallel analogue.
    out: a float64[npix] array to accumulate into instead of a fresh zero map (bench.py's timing loop: a fresh 101 MB
    allocation per call is page faults, not painting)."""
    ra, dec, M, a, D, R = map(_f, (ra, dec, M, a, D, R))
    n = ra.size
    ndim, shape, axes_concat, values = _table_args(axes, values_log)
    eptr, n_extra, _keep = _extra_ptr(extra, n)
    assert ndim == 3 + n_extra
    if out is None:
        out = np.zeros(nside2npix(nside))
    assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.size == nside2npix(nside)
    if njobs is None:
        ptot = lib().orc_paint_shell(nside, n, ra, dec, M, a, D, R, eptr, n_extra, float(eps_run),
                                     int(bool(include_pixel_size)), ndim, shape, axes_concat,
                                     values, out)
    else:
        assert not include_pixel_size  # Parallelize.py:271 does not forward it
        ptot = lib().orc_paint_shell_splitjoin(int(njobs), nside, n, ra, dec, M, a, D, R, eptr,
                                               n_extra, float(eps_run), ndim, shape, axes_concat,
                                               values, out)
    return out, int(ptot)


def baryonify_offsets(nside, ra, dec, M, a, D, R, R_model_com, axes, values, eps_run,
                      eps_model, rdelta_sampling=False, extra=None, out=None):
    """BaryonifyShell.process halo loop (HealpixRunner.py:315-355). Returns (offsets[npix,3], P_tot).
    out: a float64[npix, 3] array to accumulate into instead of a fresh zero field."""
    ra, dec, M, a, D, R, R_model_com = map(_f, (ra, dec, M, a, D, R, R_model_com))
    n = ra.size
    ndim, shape, axes_concat, values = _table_args(axes, values)
    eptr, n_extra, _keep = _extra_ptr(extra, n)
    assert ndim == 3 + n_extra
    off = np.zeros((nside2npix(nside), 3)) if out is None else out
    assert off.dtype == np.float64 and off.flags["C_CONTIGUOUS"] and off.shape == (nside2npix(nside), 3)
    ptot = lib().orc_baryonify_offsets(nside, n, ra, dec, M, a, D, R, R_model_com, eptr, n_extra,
                                       float(eps_run), float(eps_model), int(bool(rdelta_sampling)),
                                       ndim, shape, axes_concat, values, off)
    return off, int(ptot)


def paint_shell_callable(cosmo, nside, ra, dec, M, z, eps_run, projected, include_pixel_size=False, Delta=200,
                         rho_type="critical"):
    """HealpixRunner.py:449-481 line by line for a model that is a Python callable (small cases: a python loop).
    projected(r_com, M_j, a_j) -> values; returns (map, P_tot)."""
    a, R, D = halo_scalars(cosmo, M, z, Delta, rho_type)
    new_map = np.zeros(nside2npix(nside))
    pixarea = nside2pixarea(nside)
    ptot = 0
    for j in range(len(M)):
        vec_j = ang2vec(ra[j], dec[j], lonlat=True)                         # :460
        radius = R[j] * eps_run / D[j]                                      # :462
        pixind = query_disc(nside, vec_j, radius)                           # :463
        vec = np.stack(pix2vec(nside, pixind), axis=1) if pixind.size else np.zeros((0, 3))
        diff = vec * D[j] - vec_j * D[j]                                    # :465-467
        r_sep = np.sqrt(np.sum(diff ** 2, axis=1))
        Paint = np.asarray(projected(r_sep / a[j], M[j], a[j]), dtype=np.float64).reshape(-1)   # :472
        Paint = np.where(np.isfinite(Paint), Paint, 0)                      # :473
        if include_pixel_size:
            Paint = Paint * (pixarea * D[j] ** 2)                           # :478
        new_map[pixind] += Paint                                            # :481
        ptot += pixind.size
    return new_map, ptot


def baryonify_offsets_callable(cosmo, nside, ra, dec, M, z, eps_run, displacement, Delta=200, rho_type="critical"):
    """HealpixRunner.py:313-355 line by line for a model that is a Python callable.
    displacement(r_com, M_j, a_j) -> comoving displacement; returns (pix_offsets[Npix, 3], P_tot)."""
    a, R, D = halo_scalars(cosmo, M, z, Delta, rho_type)
    pix_offsets = np.zeros((nside2npix(nside), 3))
    ptot = 0
    for j in range(len(M)):
        vec_j = ang2vec(ra[j], dec[j], lonlat=True)
        radius = R[j] * eps_run / D[j]
        pixind = query_disc(nside, vec_j, radius)
        if pixind.size < 4:                                                 # :333-334
            pixind = get_interp_weights(nside, ra[j], dec[j], lonlat=True)[0]
        vec = np.stack(pix2vec(nside, pixind), axis=1)
        pos_j, pos = vec_j * D[j], vec * D[j]
        diff = pos - pos_j
        r_sep = np.sqrt(np.sum(diff ** 2, axis=1))
        with np.errstate(all="ignore"):
            offset = np.asarray(displacement(r_sep / a[j], M[j], a[j]), dtype=np.float64).reshape(-1) * a[j]      # :345
            offset = offset[:, None] * (diff / r_sep[:, None])              # :346
            offset = np.where(np.isfinite(offset), offset, 0)               # :347
        nw_pos = pos + offset
        nw_vec = nw_pos / np.sqrt(np.sum(nw_pos ** 2, axis=1))[:, None]
        np.add.at(pix_offsets, pixind, nw_vec - vec)                        # :355 (fancy += : last write wins for duplicates; none here)
        ptot += pixind.size
    return pix_offsets, ptot


def regrid_shell(nside, pix_offsets, orig_map):
    """HealpixRunner.py:357-365"""
    out = np.zeros(nside2npix(nside))
    lib().orc_regrid_shell(nside, _f(pix_offsets), _f(orig_map), out)
    return out


def regrid_pixels_hpix(hmap, parent_pix_vals, child_pix, child_weights):
    """HealpixRunner.py:17-71 (child arrays [N,4])"""
    hmap = _f(hmap)
    lib().orc_regrid_pixels_hpix(hmap, len(parent_pix_vals), _f(parent_pix_vals),
                                 np.ascontiguousarray(child_pix, dtype=np.int64), _f(child_weights))
    return hmap


def baryonify_shell(nside, orig_map, ra, dec, M, a, D, R, R_model_com, axes, values, eps_run,
                    eps_model, rdelta_sampling=False, extra=None):
    """BaryonifyShell.process (HealpixRunner.py:252-373) given per-halo scalars."""
    orig_map = _f(orig_map)
    if np.allclose(orig_map, 0):  # :293-294
        return orig_map
    off, _ = baryonify_offsets(nside, ra, dec, M, a, D, R, R_model_com, axes, values, eps_run,
                               eps_model, rdelta_sampling, extra)
    new_map = regrid_shell(nside, off, orig_map)
    assert np.isclose(np.sum(new_map), np.sum(orig_map))  # :368-370
    return new_map


def paint_anis_shell(cosmo, nside, orig_map, shell_redshift, ra, dec, M, z, axes, T_paint, T_tracer, T_mtot,
                     proj_cutoff, background_val, global_tracer_fraction, eps_run, include_pixel_size=False):
    """PaintProfilesAnisShell.process (HealpixRunner.py:513-640), restated halo by halo in numpy (small cases only).
    axes = (ln(1+z), ln M, ln r) shared by the three tables; T_* are the raw (not log) projected tables."""
    from scipy import interpolate
    ra, dec, M, z = map(_f, (ra, dec, M, z))
    orig_map = _f(orig_map)
    npix = nside2npix(nside)
    pixarea = nside2pixarea(nside)
    a, R, D = halo_scalars(cosmo, M, z)
    z_t = np.linspace(0, np.max(z) + 0.1, 1000)
    D_a = interpolate.CubicSpline(z_t, angular_diameter_distance(cosmo, 1 / (1 + z_t)))       # :553-555
    with np.errstate(all="ignore"):
        lnP, lnT, lnM_ = np.log(_f(T_paint)), np.log(_f(T_tracer)), np.log(_f(T_mtot))
    Mtot_map, _ = paint_shell(nside, ra, dec, M, a, D, R, axes, lnM_, eps_run, include_pixel_size=True)   # :565-571
    dL = 2 * proj_cutoff                                                                       # :573
    dD = D_a(shell_redshift)
    dV = pixarea * ((dD + dL) ** 3 - dD ** 3)
    rho_halos = np.sum(Mtot_map) / (dV * Mtot_map.size)
    rho_m = rho_x(cosmo, 1 / (shell_redshift + 1), "matter")                                   # :580
    drho_m = np.clip(rho_m - rho_halos, 0, None)
    Mtot_map = Mtot_map + dV * drho_m
    new_map = np.zeros(npix)
    for j in range(ra.size):                                                                   # :595-625
        vec_j = ang2vec(ra[j], dec[j], lonlat=True)
        pixind = query_disc(nside, vec_j, R[j] * eps_run / D[j], inclusive=False, nest=False)
        if pixind.size == 0:
            continue
        vec = np.stack(pix2vec(nside, pixind), axis=1)
        r_sep = np.sqrt(np.sum((vec * D[j] - vec_j * D[j]) ** 2, axis=1))
        with np.errstate(all="ignore"):
            pts = np.stack([np.full(pixind.size, np.log(1 / a[j])), np.full(pixind.size, np.log(M[j])),
                            np.log(r_sep / a[j])], axis=1)
            Painting = np.exp(interp_linear(axes, lnP, pts))
            Canvas = np.exp(interp_linear(axes, lnT, pts))
        Painting = np.where(np.isfinite(Painting), Painting, 0)
        Canvas = np.where(np.isfinite(Canvas), Canvas, 0)
        Mfrac = np.divide(Canvas, Mtot_map[pixind], out=np.zeros_like(Canvas), where=Mtot_map[pixind] > 0)
        Mfrac = Mfrac * orig_map[pixind]
        if include_pixel_size:
            Painting = Painting * (pixarea * D[j] ** 2)
        new_map[pixind] += Painting * Mfrac
    Mfrac = np.divide(dV * drho_m, Mtot_map, out=np.zeros_like(Mtot_map), where=Mtot_map > 0) * orig_map   # :628-630
    return new_map + background_val * global_tracer_fraction * Mfrac


def displacement_readout(cosmo, axes, values, r, M, a, eps_model, rdelta_sampling=False, Delta=200, rho_type="critical",
                         lnM=None, extra=()):
    """BaryonificationClass._readout for scalar M, a (BaryonCorrection.py:331-419): linear table of comoving
    displacement on (ln(1+z), ln M, ln r [- ln R_com]), NaN outside the hull, 0 where r >= eps_model * R_com.
    lnM: the table coordinate if it is not ln(M) in float64 (a float32 catalog column gives a float32 logarithm, :397)."""
    r = _f(np.atleast_1d(r))
    R = float(get_radius(cosmo, M, a, Delta, rho_type)) / a                 # comoving Mpc (:399)
    with np.errstate(all="ignore"):
        r_in = np.log(r) - (np.log(R) if rdelta_sampling else 0.0)
    pts = np.stack([np.full(r.size, np.log(1 / a)), np.full(r.size, np.log(M) if lnM is None else lnM), r_in] +
                   [np.full(r.size, float(e)) for e in extra], axis=1)       # **kwargs in p_keys order (BaryonCorrection.py:374, :404-408)
    d = interp_linear(axes, values, pts)
    return np.where(r < eps_model * R, d, 0.0)                              # :410-411


def baryonify_snapshot(cosmo, L, redshift, px, py, pz, hM, hx, hy, hz, axes, values, eps_run, eps_model,
                       rdelta_sampling=False, Delta=200, rho_type="critical", extra=None):
    """BaryonifySnapshot.process (SnapshotRunner.py:176-275) restated with scipy's periodic KDTree (:99).
    pz / hz = None: 2D snapshot.  The reference keeps the halo columns in float32 (io.py:204): positions and masses are
    widened exactly, but the table coordinate ln M is a float32 logarithm (BaryonCorrection.py:397 on a float32 M).
    extra: [n_halo, n_extra] p_keys columns of the halo catalogue (float32 there, io.py:204; SnapshotRunner.py:223 o_j).
    Returns the displaced, box-wrapped particle coordinates [n, ndim]."""
    from scipy.spatial import KDTree
    ex = None if extra is None else np.asarray(extra, dtype=np.float32).astype(np.float64).reshape(len(hM), -1)
    is2D = pz is None
    P = np.stack([_f(px), _f(py)] + ([] if is2D else [_f(pz)]), axis=1)
    H = np.stack([_f(np.asarray(c, dtype=np.float32)) for c in ([hx, hy] + ([] if is2D else [hz]))], axis=1)
    tree = KDTree(P, boxsize=L)                                              # :99
    tot = np.zeros_like(P)
    a = 1 / (1 + redshift)
    for j in range(H.shape[0]):                                              # :212-258
        M32 = np.float32(hM[j])
        M_j = float(M32)
        lnM_j = float(np.log(M32))
        R_j = float(get_radius(cosmo, M_j, a, Delta, rho_type))
        R_q = np.clip(eps_run * R_j / a, 0, L / 2)
        inds = tree.query_ball_point(H[j], R_q)
        if len(inds) == 0:
            continue
        dd = P[inds] - H[j]
        dd = np.where(dd > L / 2, dd - L, dd)                                # compute_distance / enforce_periodicity
        dd = np.where(dd < -L / 2, dd + L, dd)
        d = np.sqrt(np.sum(dd ** 2, axis=1))
        with np.errstate(all="ignore"):
            off = displacement_readout(cosmo, axes, values, d, M_j, a, eps_model, rdelta_sampling, Delta, rho_type,
                                       lnM=lnM_j, extra=() if ex is None else ex[j])
            off = np.where(np.isfinite(off), off, 0)                         # :231 / :248
            tot[inds] += off[:, None] * (dd / d[:, None])
    new = P + tot
    new = np.where(new > L, new - L, new)                                    # :268-273
    new = np.where(new < 0, new + L, new)
    return new


def make_map(pos, mass, L, n_grid, mode="ngp"):
    """Mass map of particles pos[n, ndim] on a periodic n_grid^ndim grid.  'ngp' = ParticleSnapshot.make_map
    (io.py:629-677: numpy.histogramdd on linspace(0, L, n_grid + 1)); 'cic' = cloud-in-cell on the cell centres
    (i + 1/2) L / n_grid with periodic wrap (the BASELINE config asks for CIC; the reference itself only has NGP)."""
    pos = _f(pos)
    ndim = pos.shape[1]
    mass = np.ones(pos.shape[0]) if mass is None else _f(mass)
    if mode == "ngp":
        bins = np.linspace(0, L, n_grid + 1)
        return np.histogramdd(pos, bins=(bins,) * ndim, weights=mass)[0]
    u = pos / (L / n_grid) - 0.5
    f = np.floor(u)
    w1 = u - f
    i0 = f.astype(np.int64) % n_grid
    out = np.zeros((n_grid,) * ndim)
    for corner in range(1 << ndim):
        w = mass.copy()
        idx = []
        for k in range(ndim):
            bit = (corner >> k) & 1
            w = w * (w1[:, k] if bit else 1.0 - w1[:, k])
            idx.append((i0[:, k] + bit) % n_grid)
        np.add.at(out, tuple(idx), w)
    return out


# --------------------------------------------------------------------------
# Periodic Cartesian grids (BaryonForge/Runners/Map2DRunner.py)
# --------------------------------------------------------------------------
def _grid_cutout(bins, Npix, res, hpos, R_cut):
    """the cut-out of one halo as Map2DRunner builds it (:485-505 / :720-742): even size, stretched linspace offsets,
    nearest-bin centre, wrapped index ranges"""
    Nsize = 2 * R_cut / res
    Nsize = int(Nsize // 2) * 2
    Nsize = int(np.clip(Nsize, 2, bins.size // 2))
    x = np.linspace(-Nsize / 2, Nsize / 2, Nsize) * res
    w = Nsize // 2
    cen = [int(np.argmin(np.abs(bins - h))) for h in hpos]
    inds = []
    for c in cen:                                                            # pick_indices (:400-428)
        i = np.arange(c - w, c + w)
        i = np.where(i < 0, i + Npix, i)
        i = np.where(i >= Npix, i - Npix, i)
        inds.append(i)
    d = [bins[c] - h for c, h in zip(cen, hpos)]
    return x, inds, d


def build_Rmat(A, q):
    """DefaultRunnerGrid.build_Rmat (Map2DRunner.py:281-350), 2D branch, with the dtypes the runner hands it: A is the
    float32 'A_ell' row already divided by its norm (:478-480), q the float32 'q_ell'."""
    A = np.array(A)
    A /= np.linalg.norm(A)
    ref = np.array([1., 0.])
    beta = np.arccos(np.dot(A, ref))
    eta = -np.log(q)
    if eta > 1e-4:
        eta2g = np.tanh(0.5 * eta) / eta
    else:
        etasq = eta * eta
        eta2g = 0.5 + etasq * ((-1 / 24) + etasq * (1 / 240))
    g = eta2g * eta * np.exp(2j * beta)
    g1, g2 = g.real, g.imag
    det = np.sqrt(1 - np.abs(g) ** 2)
    return np.array([[1 + g1, g2], [g2, 1 - g1]]) / det


def _grid_r(comps, Rmat=None):
    """radius handed to the model: circular, or after the shear of the 2D ellipticity option (:520-524)"""
    if Rmat is None:
        return np.sqrt(sum(c * c for c in comps))
    xy = np.vstack([comps[0], comps[1]]).T @ Rmat
    return np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)


def _grid_flat_and_r(x, inds, d, Npix):
    """flat map indices of the cut-out and the offsets (x_grid + dx, y_grid + dy[, z_grid + dz]) paired with them, in the
    reference's pairing: inds[x_inds, :][:, y_inds].flatten() against np.meshgrid(x, x, indexing='xy') (:507-516)"""
    if len(inds) == 2:
        flat = (inds[0][:, None] * Npix + inds[1][None, :]).ravel()
        xg, yg = np.meshgrid(x, x, indexing="xy")
        comps = [(xg + d[0]).ravel(), (yg + d[1]).ravel()]
    else:
        flat = ((inds[0][:, None, None] * Npix + inds[1][None, :, None]) * Npix + inds[2][None, None, :]).ravel()
        xg, yg, zg = np.meshgrid(x, x, x, indexing="xy")
        comps = [(xg + d[0]).ravel(), (yg + d[1]).ravel(), (zg + d[2]).ravel()]
    return flat, comps


def paint_grid(cosmo, bins, shape, redshift, hpos, hM, axes, T, eps_run, include_pixel_size=True, Delta=200,
               rho_type="critical", q_ell=None, A_ell=None, extra=None):
    """PaintProfilesGrid.process (Map2DRunner.py:676-829) without ellipticity.  hpos [n, ndim], T = the raw table the model
    reads for this dimensionality (projected for 2D maps, real for 3D).  Halo columns are narrowed to float32 as in
    HaloNDCatalog (io.py:204); ln M is a float32 logarithm (Tabulate.py:316)."""
    bins = _f(bins)
    Npix, nd = bins.size, len(shape)
    res = bins[1] - bins[0]
    a = 1 / (1 + redshift)
    new_map = np.zeros(int(np.prod(shape)))
    with np.errstate(all="ignore"):
        lnT = np.log(_f(T))
    hpos32 = np.asarray(hpos, dtype=np.float32).astype(np.float64)
    ex = None if extra is None else np.asarray(extra, dtype=np.float32).astype(np.float64).reshape(hpos32.shape[0], -1)   # :731 o_j
    for j in range(hpos32.shape[0]):
        M32 = np.float32(hM[j])
        R_j = float(get_radius(cosmo, float(M32), a, Delta, rho_type)) / a                  # comoving (:708)
        x, inds, d = _grid_cutout(bins, Npix, res, hpos32[j, :nd], eps_run * R_j)
        flat, comps = _grid_flat_and_r(x, inds, d, Npix)
        Rmat = None
        if q_ell is not None:                                                               # :710-713, :753-757
            A = np.asarray(A_ell[j], dtype=np.float32)
            Rmat = build_Rmat(A / np.sqrt(np.sum(A ** 2)), np.float32(q_ell[j]))
        r = _grid_r(comps, Rmat)
        with np.errstate(all="ignore"):
            pts = np.stack([np.full(r.size, np.log(1 / a)), np.full(r.size, float(np.log(M32))), np.log(r)] +
                           ([] if ex is None else [np.full(r.size, e) for e in ex[j]]), axis=1)
            P = np.exp(interp_linear(axes, lnT, pts))
        mask = np.isfinite(P) & (r < R_j * eps_run)                                         # :812-815
        if mask.sum() == 0:
            continue
        new_map[flat] += np.where(mask, P, 0)                                                # :820-823
    if include_pixel_size:
        new_map *= res ** nd                                                                  # :826
    return new_map.reshape(shape)


def regrid_pixels_grid(N, positions, values, nd):
    """regrid_pixels_2D / regrid_pixels_3D (Map2DRunner.py:14-82, :86-162): every unit pixel displaced to `positions`
    (pixel units) deposits value x overlap onto the periodic N^nd grid; grid[i, j(, k)] with i from the y-range, j from
    the x-range (and k from z).  Vectorised over pixels, candidate cells as in the reference loops."""
    start = np.mod(_f(positions), N)                                                          # :52 / :127
    end = start + 1
    out = np.zeros((N,) * nd)
    offs = np.arange(-2, 4)                                                                    # int(start) - 2 ... int(end) + 1
    base = start.astype(np.int64)                                                              # int() truncation, start >= 0

    def overlaps(axis):
        c = base[:, axis][:, None] + offs[None, :]                                             # candidate cells
        valid = c < (end[:, axis].astype(np.int64) + 2)[:, None]                               # range(x_min, x_max)
        c = np.where(c < 0, c + N, c)
        c = np.where(c + 1 > N, c % N, c)
        s, e = start[:, axis][:, None], end[:, axis][:, None]
        dx = np.minimum(c + 1, e) - np.maximum(c, s)
        dx = np.where(dx < 0, np.minimum(c + 1, e + N) - np.maximum(c, s + N), dx)
        dx = np.where(dx < 0, np.minimum(c + 1, e - N) - np.maximum(c, s - N), dx)
        return c, np.where(valid, dx, -1.0)

    cx, dx = overlaps(0)
    cy, dy = overlaps(1)
    v = _f(values)
    if nd == 2:
        w = dy[:, :, None] * dx[:, None, :]
        ok = (dy[:, :, None] > 0) & (dx[:, None, :] > 0)
        ii = np.broadcast_to(cy[:, :, None], w.shape)
        jj = np.broadcast_to(cx[:, None, :], w.shape)
        np.add.at(out, (ii[ok], jj[ok]), (w * v[:, None, None])[ok])
    else:
        cz, dz = overlaps(2)
        w = dy[:, :, None, None] * dx[:, None, :, None] * dz[:, None, None, :]
        ok = (dy[:, :, None, None] > 0) & (dx[:, None, :, None] > 0) & (dz[:, None, None, :] > 0)
        ii = np.broadcast_to(cy[:, :, None, None], w.shape)
        jj = np.broadcast_to(cx[:, None, :, None], w.shape)
        kk = np.broadcast_to(cz[:, None, None, :], w.shape)
        np.add.at(out, (ii[ok], jj[ok], kk[ok]), (w * v[:, None, None, None])[ok])
    return out


def baryonify_grid(cosmo, bins, orig_map, redshift, hpos, hM, axes, d_table, eps_run, eps_model, rdelta_sampling=False,
                   Delta=200, rho_type="critical", q_ell=None, A_ell=None):
    """BaryonifyGrid.process (Map2DRunner.py:431-621) without ellipticity: per-pixel offsets (in pixel widths) summed over
    halos, then the overlap regrid; mass conserved (:617-619)."""
    bins = _f(bins)
    orig_map = _f(orig_map)
    Npix, nd = bins.size, orig_map.ndim
    res = bins[1] - bins[0]
    a = 1 / (1 + redshift)
    pix_offsets = np.zeros((orig_map.size, nd))
    hpos32 = np.asarray(hpos, dtype=np.float32).astype(np.float64)
    for j in range(hpos32.shape[0]):
        M32 = np.float32(hM[j])
        R_j = float(get_radius(cosmo, float(M32), a, Delta, rho_type))                     # physical (:471)
        R_q = np.clip(eps_run * R_j / a, 0, np.max(bins) / 2)                               # :473-474
        x, inds, d = _grid_cutout(bins, Npix, res, hpos32[j, :nd], R_q)
        flat, comps = _grid_flat_and_r(x, inds, d, Npix)
        r = np.sqrt(sum(c * c for c in comps))                                              # unit vectors use the circular radius
        r_mod = r
        if q_ell is not None:
            A = np.asarray(A_ell[j], dtype=np.float32)
            r_mod = _grid_r(comps, build_Rmat(A / np.sqrt(np.sum(A ** 2)), np.float32(q_ell[j])))
        with np.errstate(all="ignore"):
            off = displacement_readout(cosmo, axes, d_table, r_mod, float(M32), a, eps_model, rdelta_sampling, Delta, rho_type,
                                       lnM=float(np.log(M32))) / res                        # :530 / :570
            for k in range(nd):
                pix_offsets[flat, k] += off * (comps[k] / r)
    pix_offsets = np.where(np.isfinite(pix_offsets), pix_offsets, 0)                        # :591 / :603
    grids = np.meshgrid(*([np.arange(Npix)] * nd), indexing="xy")
    for k in range(nd):
        pix_offsets[:, k] += grids[k].ravel()
    return regrid_pixels_grid(Npix, pix_offsets, orig_map.ravel(), nd)


def paint_anis_grid(cosmo, bins, orig_map, redshift, hpos, hM, axes, T_paint, T_tracer, T_mtot, proj_cutoff, background_val,
                    global_tracer_fraction, eps_run, include_pixel_size=True, q_ell=None, A_ell=None):
    """PaintProfilesAnisGrid.process (Map2DRunner.py:847-1015), 2D maps only (:849)."""
    bins = _f(bins)
    orig = _f(orig_map)
    assert orig.ndim == 2
    Npix = bins.size
    res = bins[1] - bins[0]
    a = 1 / (1 + redshift)
    Mtot = paint_grid(cosmo, bins, orig.shape, redshift, hpos, hM, axes, T_mtot, eps_run, include_pixel_size=False,
                      q_ell=q_ell, A_ell=A_ell).ravel()                                       # :866-871
    dL = 2 * proj_cutoff                                                                      # :876-878
    dV = res ** 2 * dL
    rho_halos = np.average(Mtot) / dL
    rho_m = rho_x(cosmo, 1.0, "matter")                                                       # comoving matter density (:886)
    drho_m = np.clip(rho_m - rho_halos, 0, None)
    Mtot = Mtot + dV * drho_m
    with np.errstate(all="ignore"):
        lnP, lnC = np.log(_f(T_paint)), np.log(_f(T_tracer))
    new_map = np.zeros(orig.size)
    of = orig.ravel()
    hpos32 = np.asarray(hpos, dtype=np.float32).astype(np.float64)
    for j in range(hpos32.shape[0]):                                                          # :901-1000
        M32 = np.float32(hM[j])
        R_j = float(get_radius(cosmo, float(M32), a)) / a
        x, inds, d = _grid_cutout(bins, Npix, res, hpos32[j, :2], eps_run * R_j)
        flat, comps = _grid_flat_and_r(x, inds, d, Npix)
        Rmat = None
        if q_ell is not None:
            A = np.asarray(A_ell[j], dtype=np.float32)
            Rmat = build_Rmat(A / np.sqrt(np.sum(A ** 2)), np.float32(q_ell[j]))
        r = _grid_r(comps, Rmat)
        with np.errstate(all="ignore"):
            pts = np.stack([np.full(r.size, np.log(1 / a)), np.full(r.size, float(np.log(M32))), np.log(r)], axis=1)
            Painting = np.exp(interp_linear(axes, lnP, pts))
            Canvas = np.exp(interp_linear(axes, lnC, pts))
        Canvas = np.where(np.isfinite(Canvas), Canvas, 0)
        Mfrac = np.divide(Canvas, Mtot[flat], out=np.zeros_like(Canvas), where=Mtot[flat] > 0) * of[flat]
        mask = np.isfinite(Painting) & (r < R_j * eps_run)
        if mask.sum() == 0:
            continue
        new_map[flat] += np.where(mask, Painting, 0) * Mfrac
    Mfrac = np.divide(dV * drho_m, Mtot, out=np.zeros_like(Mtot), where=Mtot > 0) * of
    new_map += background_val * global_tracer_fraction * Mfrac
    new_map = new_map.reshape(orig.shape)
    if include_pixel_size:
        new_map = new_map * res ** 2
    return new_map

