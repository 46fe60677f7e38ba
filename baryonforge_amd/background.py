"""
Flat wCDM background + spherical-overdensity mass definitions: the three pyccl
calls that sit on the per-halo path of the reference, evaluated host-side once
per process() call (vectorised) or on the GPU per halo (bfg_massdef in the C-ABI).

  ccl.Cosmology(Omega_c, Omega_b, h, sigma8, n_s, w0, 'linear')   Runners/HealpixRunner.py:280-285, :416-421
  ccl.angular_diameter_distance(cosmo, a)                          :299, :431
  mass_def.get_radius(cosmo, M, a)                                 :320, :454 ; Profiles/BaryonCorrection.py:399

pyccl is not a dependency of this package.  The defaults below are CCL's
(flat, T_CMB = 2.7255 K, N_eff = 3.044 massless neutrinos, T_ncdm = 0.71611);
`cosmo.compute_sigma()` (HealpixRunner.py:285) is dead work for tabulated
models and is not reproduced.  A pyccl Cosmology object is also accepted
wherever a cosmology is expected (its parameters are read, not its splines).
"""
import numpy as np

CLIGHT = 299792458.0            # m/s
GNEWT = 6.67430e-11             # m^3/kg/s^2
MPC_TO_METER = 3.085677581491367e22
GM_SUN = 1.3271244e20           # m^3/s^2
STBOLTZ = 5.670374419e-8        # W/m^2/K^4
T_CMB_DEFAULT = 2.7255
N_EFF_DEFAULT = 3.044
T_NCDM_DEFAULT = 0.71611
#: 3 (100 km/s/Mpc)^2 / (8 pi G) in Msun / Mpc^3; G M_sun enters as one well-measured product
RHO_CRITICAL = 3.0e10 * MPC_TO_METER / (8.0 * np.pi * GM_SUN)

REQUIRED_KEYS = ("Omega_m", "sigma8", "h", "Omega_b", "n_s", "w0")


def check_cosmology_dict(cosmo):
    """utils/io.py:79-85, :357-363"""
    keys = cosmo.keys()
    if not all(k in keys for k in REQUIRED_KEYS):
        raise ValueError("Not all cosmology parameters provided. I need Omega_m, sigma8, h, sigma8, Omega_b, n_s, w0")


def as_cosmo_dict(cosmo):
    """dict with at least Omega_m, h, w0 from a dict, a Background, or a pyccl-like object."""
    if isinstance(cosmo, Background):
        return cosmo.params
    if isinstance(cosmo, dict):
        return cosmo
    # pyccl.Cosmology duck type: cosmo['Omega_m'] style item access
    out = {}
    for k in ("Omega_m", "Omega_b", "h", "sigma8", "n_s", "w0"):
        try:
            out[k] = float(cosmo[k])
        except Exception:
            pass
    if "Omega_m" not in out:
        try:
            out["Omega_m"] = float(cosmo["Omega_c"]) + float(cosmo["Omega_b"])
        except Exception as exc:
            raise TypeError(f"cannot read cosmological parameters from {type(cosmo)}") from exc
    out.setdefault("w0", -1.0)
    return out


def lcdm(cosmo):
    """The cosmology the reference's grid and snapshot runners build: the same dict WITHOUT w0 (Map2DRunner.py:462-465,
    :705-708, :851-854; SnapshotRunner.py:197-200 pass no w0 to ccl.Cosmology), i.e. w0 = -1 whatever the catalog says."""
    out = dict(as_cosmo_dict(cosmo))
    out["w0"] = -1.0
    return out


class Background(object):
    """E(a), chi(a), D_A(a), rho_x(a) of a flat wCDM cosmology with radiation."""

    def __init__(self, cosmo, T_CMB=T_CMB_DEFAULT, N_eff=N_EFF_DEFAULT, T_ncdm=T_NCDM_DEFAULT, nu_rel="T_ncdm"):
        """nu_rel: the density of the massless neutrinos, Omega_nu,rel = N_eff 7/8 x^4 Omega_gamma with
        "T_ncdm"   x = T_ncdm = 0.71611 (the default: what pyccl >= 2.1 does for ccl.Cosmology's defaults, as recalled from its
                   source -- T_nu = T_CMB T_ncdm; pyccl is not installed here and the reference's tests hold no distances);
        "4/11"     x = (4/11)^(1/3) = 0.71377, the instantaneous-decoupling textbook value.
        The two differ by 1.3 % in Omega_nu,rel, i.e. 2.5e-7 relative in D_A at z = 0.5 and 1.2e-6 at z = 3 (tests/test_oracle_healpix.py):
        40x inside the 1e-5 map tolerance, but it is the one constant of this module a run against live pyccl
        (tests/golden/make_golden.py --real-deps) has to decide."""
        if nu_rel not in ("T_ncdm", "4/11"):
            raise ValueError("nu_rel must be 'T_ncdm' or '4/11'")
        if nu_rel == "4/11":
            T_ncdm = (4.0 / 11.0) ** (1.0 / 3.0)
        self.nu_rel = nu_rel
        p = as_cosmo_dict(cosmo)
        self.params = dict(p)
        self.h = float(p["h"])
        self.Omega_m = float(p["Omega_m"])
        self.w0 = float(p.get("w0", -1.0))
        rho_crit_si = 3.0 * (100.0e3 / MPC_TO_METER) ** 2 / (8.0 * np.pi * GNEWT) * self.h ** 2
        Omega_g = 4.0 * STBOLTZ / CLIGHT ** 3 * T_CMB ** 4 / rho_crit_si
        Omega_nu = N_eff * 7.0 / 8.0 * T_ncdm ** 4 * Omega_g
        self.Omega_r = Omega_g + Omega_nu
        self.Omega_l = 1.0 - self.Omega_m - self.Omega_r

    def E2(self, a):
        a = np.asarray(a, dtype=np.float64)
        return self.Omega_m / a ** 3 + self.Omega_l * a ** (-3.0 * (1.0 + self.w0)) + self.Omega_r / a ** 4

    _X, _W = np.polynomial.legendre.leggauss(128)

    def comoving_radial_distance(self, a):
        """chi(a) = (c/H0) int_a^1 da' / (a'^2 E(a'))  [Mpc], Gauss-Legendre (rel. error < 1e-12 for z <= 30)"""
        a = np.asarray(a, dtype=np.float64)
        af = np.atleast_1d(a).ravel()
        # substitute a' = exp(u): int du / (a' E)  -- smooth over many decades of a
        u0 = np.log(af)[:, None]
        u = 0.5 * u0 * (1.0 - self._X[None, :])          # from u0 (X=-1) to 0 (X=+1)
        ap = np.exp(u)
        integrand = 1.0 / (ap * np.sqrt(self.E2(ap)))
        chi = np.sum(integrand * self._W[None, :], axis=1) * (-0.5 * u0[:, 0])
        return (CLIGHT / 1.0e5 / self.h * chi).reshape(a.shape)

    def angular_diameter_distance(self, a):
        """physical Mpc (flat: D_A = a chi)"""
        a = np.asarray(a, dtype=np.float64)
        return a * self.comoving_radial_distance(a)

    def rho_x(self, a, rho_type="critical"):
        """physical density, Msun / Mpc^3"""
        a = np.asarray(a, dtype=np.float64)
        if rho_type == "critical":
            return RHO_CRITICAL * self.h ** 2 * self.E2(a)
        if rho_type == "matter":
            return RHO_CRITICAL * self.h ** 2 * self.Omega_m / a ** 3
        raise ValueError(f"rho_type must be 'critical' or 'matter', not {rho_type!r}")


class MassDef(object):
    """Stand-in for ccl.halos.massdef.MassDef(Delta, rho_type) (numeric Delta only)."""

    def __init__(self, Delta=200, rho_type="critical"):
        if isinstance(Delta, str):
            raise NotImplementedError("only numeric overdensities are supported (no 'vir'/'fof')")
        if rho_type not in ("critical", "matter"):
            raise ValueError("rho_type must be 'critical' or 'matter'")
        self.Delta = float(Delta)
        self.rho_type = rho_type

    def get_radius(self, cosmo, M, a):
        """(M / (4.18879020479 Delta rho_x(a)))^(1/3), physical Mpc"""
        bg = cosmo if isinstance(cosmo, Background) else Background(cosmo)
        return (np.asarray(M, dtype=np.float64) / (4.18879020479 * self.Delta * bg.rho_x(a, self.rho_type))) ** (1.0 / 3.0)

    def __repr__(self):
        return f"MassDef(Delta={self.Delta:g}, rho_type={self.rho_type!r})"


def massdef_params(mass_def):
    """(Delta, rho_type) of one of ours or of a pyccl MassDef object."""
    if mass_def is None:
        return 200.0, "critical"
    Delta = getattr(mass_def, "Delta")
    rho_type = getattr(mass_def, "rho_type", "critical")
    if isinstance(Delta, str):
        raise NotImplementedError(f"mass definition Delta={Delta!r} is not supported on the MI355X path")
    return float(Delta), str(rho_type)
