"""
Seeded synthetic inputs for tests and bench.py (SURVEY.md 8d / BASELINE.md 2): there is no
network, pyccl or healpy here, so catalogs and tables are analytic and reproducible.
Shapes follow the reference: catalog of tests/test_healpix.py:27-32, table grids of
utils/Tabulate.py:193-195 and the notebooks' 2 x 30 x 2000 stress shape.
"""
import numpy as np

from .background import Background, MassDef

COSMO = {"Omega_m": 0.30, "Omega_b": 0.04, "h": 0.7, "sigma8": 0.8, "n_s": 0.96, "w0": -1.0}


def catalog(n, seed=42, logM=(12.0, 15.5), z=(0.4, 0.5), steep=False):
    """ra, dec [deg], M [Msun], z.  Uniform on the sphere; log10 M ~ U(12, 15.5) or dn/dlnM ~ M^-0.9 (steep)."""
    rng = np.random.default_rng(seed)
    ra = np.degrees(rng.uniform(0.0, 2 * np.pi, n))
    dec = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, n)))
    if steep:
        u = rng.uniform(0, 1, n)
        lo, hi, s = 10.0 ** logM[0], 10.0 ** logM[1], -0.9
        M = (lo ** s + u * (hi ** s - lo ** s)) ** (1.0 / s)
    else:
        M = 10.0 ** rng.uniform(logM[0], logM[1], n)
    zz = rng.uniform(z[0], z[1], n)
    return ra, dec, M, zz


def _r200c_com(M, z, cosmo):
    a = 1.0 / (1.0 + z)
    return MassDef(200, "critical").get_radius(Background(cosmo), M, a) / a


def grids(nz=10, nM=30, nr=100, z_min=0.01, z_max=1.0):
    return np.geomspace(z_min, z_max, nz), np.geomspace(1e12, 1e16, nM), np.geomspace(1e-3, 1e2, nr)


def pressure_table(nz=10, nM=30, nr=100, cosmo=COSMO, bad_block=False, grid=None):
    """GNFW-like projected pressure T2D = A (M/1e14)^{5/3} (1+z)^{8/3} [1 + (r/r_c)^2]^{-1.5} * a, r_c = 0.2 R200c.
    Returns (ln(1+z), ln M, ln r, T2D).  grid = (z, M, r) arrays: on that grid instead of grids(nz, nM, nr)."""
    z, M, r = grids(nz, nM, nr) if grid is None else grid
    Z, MM, RR = np.meshgrid(z, M, r, indexing="ij")
    rc = 0.2 * _r200c_com(MM, Z, cosmo)
    T = 1e-6 * (MM / 1e14) ** (5.0 / 3.0) * (1 + Z) ** (8.0 / 3.0) * (1 + (RR / rc) ** 2) ** -1.5 / (1 + Z)
    if bad_block:
        T = T.copy()
        T[1:3, 2:5, 5:20] = 0.0
        T[3, 10, 30:40] = -1.0
        T[4, 12, 50] = np.nan
    return np.log(1 + z), np.log(M), np.log(r), T


def displacement_table(nz=10, nM=30, nr=100, cosmo=COSMO, rdelta=False, grid=None):
    """d = 0.1 Mpc (M/1e14)^{1/3} x (1 - x/4) e^{-x},  x = r / R200c,com  (signed, -> 0 at large r).
    Returns (ln(1+z), ln M, ln r  [or ln r/R_delta], d).  grid = (z, M, r) arrays: on that grid."""
    z, M, r = grids(nz, nM, nr) if grid is None else grid
    if rdelta:
        r = np.geomspace(1e-3, 10.0, nr)
    Z, MM, RR = np.meshgrid(z, M, r, indexing="ij")
    x = RR if rdelta else RR / _r200c_com(MM, Z, cosmo)
    d = 0.1 * (MM / 1e14) ** (1.0 / 3.0) * x * (1 - x / 4) * np.exp(-x)
    return np.log(1 + z), np.log(M), np.log(r), d


def mass_map(nside, seed=7):
    """baryonify input: default_rng(7).uniform(0, 10, Npix) (tests/test_healpix.py:49)"""
    return np.random.default_rng(seed).uniform(0, 10, 12 * nside * nside)
