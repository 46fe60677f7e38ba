"""
Displacement-function models, mirroring BaryonForge/Profiles/BaryonCorrection.py:
`BaryonificationClass` (:15-460), `Baryonification2D` (:581-695) and
`Baryonification3D` (:464-578).

* `displacement(r, M, a, **kw)` reads the (z, M, r[, params]) displacement table
  out on the GPU (bfg_table_eval), then applies the model's epsilon_max cut and
  emits the reference's out-of-table UserWarnings (:331-419).  The shell runner
  does not call this method per halo: it hands the table to the fused kernel.
* `setup_interpolator(...)` is the table builder of :142-328 -- enclosed-mass
  profiles of the DMO and DMB models, monotonic masking, two PCHIP inversions,
  d(r) = M_DMB^-1(M_DMO(r)) - r.  It is set-up time work (once per model), kept
  on the host with the same scipy primitives the reference uses.  The DMO/DMB
  profile models themselves (pyccl profile zoo) are out of scope: any object
  with `projected(cosmo, r, M, a)` / `real(...)` works, and `from_arrays`
  accepts a finished table (e.g. `raw_input_d` of a reference-built model).
"""
import warnings
from itertools import product

import numpy as np
from scipy import integrate, interpolate

from ..background import Background, MassDef
from ..utils.Tabulate import _set_parameter

__all__ = ["BaryonificationClass", "Baryonification2D", "Baryonification3D"]


def _monotone_mask(ln_DMB, ln_DMO):
    """Iteratively drop points until ln M_DMB(r) rises by > 1e-5 between kept neighbours
    (BaryonCorrection.py:243-274).  Returns (mask, message-or-None)."""
    keep = np.ones_like(ln_DMB).astype(bool)
    smallest_step, rounds, message = -np.inf, 0, None
    while (smallest_step < 1e-5) & (keep.sum() > 5):
        ok = ((np.diff(ln_DMB[keep], prepend=0) > 1e-5) &
              ((np.abs(ln_DMB - ln_DMO)[keep] > 1e-6) | np.isnan(ln_DMO)[keep]) &
              np.isfinite(ln_DMB)[keep])
        keep[keep] = ok
        keep[0] = True
        rounds += 1
        if rounds > 30:
            keep = np.zeros_like(keep).astype(bool)
            message = "is nearly constant over radius. Suggests density is negative or zero for most of the range."
            break
        if keep.sum() < 5:
            message = "is nearly constant over radius. Or it is broken. Less than 5 datapoints are usable."
            break
        smallest_step = np.min(np.diff(ln_DMB[keep], prepend=0)[1:])
    return keep, message


def _displacement_from_masses(r, M_DMO_i, M_DMB_i, log10M):
    """d(r) = exp(lnr_DMB(lnM_DMO(ln r))) - r for one halo mass (BaryonCorrection.py:237-291, :296-304)."""
    with np.errstate(all="ignore"):
        ln_DMB, ln_DMO = np.log(M_DMB_i), np.log(M_DMO_i)
        lnr = np.log(r)
        keep, message = _monotone_mask(ln_DMB, ln_DMO)
        if message is not None:
            warnings.warn(f"Mass profile of log10(M) = {log10M} {message}", UserWarning)
        if keep.sum() > 5:
            dmo_ok = ((np.diff(ln_DMO, prepend=0) > 1e-5) &
                      ((np.abs(ln_DMB - ln_DMO) > 1e-6) | np.isnan(ln_DMB)) & np.isfinite(ln_DMO))
            r_of_lnM_DMB = interpolate.PchipInterpolator(ln_DMB[keep], lnr[keep], extrapolate=False)
            lnM_DMO_of_r = interpolate.PchipInterpolator(lnr[dmo_ok], ln_DMO[dmo_ok], extrapolate=False)
            offset = np.exp(r_of_lnM_DMB(lnM_DMO_of_r(lnr))) - r
            return np.where(np.isfinite(offset), offset, 0)
    warnings.warn(f"Displacement function for halo with log10(M) = {log10M} failed to compute. Defaulting to d = 0.",
                  UserWarning)
    return np.zeros_like(r)


class BaryonificationClass(object):
    """
    Base displacement-function model (BaryonCorrection.py:15-460).

    Parameters
    ----------
    DMO, DMB : profile models (dark-matter-only, dark-matter+baryon)
    cosmo : cosmology (dict with Omega_m, h, w0, ... or a pyccl Cosmology)
    epsilon_max : float, displacements are zero beyond epsilon_max * R_delta (default 20)
    mass_def : MassDef (default 200 critical)
    r_min_int, r_max_int, N_int : enclosed-mass integration grid
    """

    def __init__(self, DMO, DMB, cosmo, epsilon_max=20, mass_def=None, r_min_int=1e-6, r_max_int=1000, N_int=500):
        self.DMO = DMO
        self.DMB = DMB
        for m in (self.DMO, self.DMB):                                   # :99-100, real-space cutoff at 1 Gpc
            if m is not None and hasattr(m, "set_parameter"):
                m.set_parameter("cutoff", 1000)
        self.cosmo = cosmo
        self.epsilon_max = epsilon_max
        self.mass_def = MassDef(200, "critical") if mass_def is None else mass_def
        self.r_min_int = r_min_int
        self.r_max_int = r_max_int
        self.N_int = N_int

    # ---- construction from a finished table -------------------------------------------
    @classmethod
    def from_arrays(cls, ln1pz, lnM, lnr, raw_input_d, cosmo, epsilon_max=20, mass_def=None,
                    Rdelta_sampling=False, other_params={}):
        """axes as the reference stores them (:316-320): ln(1+z), ln M, ln r (or ln r/R_delta), then params."""
        self = cls(None, None, cosmo, epsilon_max=epsilon_max, mass_def=mass_def)
        self._set_table(ln1pz, lnM, lnr, raw_input_d, Rdelta_sampling, other_params)
        return self

    def _set_table(self, ln1pz, lnM, lnr, d, Rdelta_sampling, other_params):
        self.p_keys = list(other_params.keys())
        self.raw_input_z_range = np.ascontiguousarray(ln1pz, dtype=np.float64)
        self.raw_input_M_range = np.ascontiguousarray(lnM, dtype=np.float64)
        self.raw_input_r_range = np.ascontiguousarray(lnr, dtype=np.float64)
        for k in self.p_keys:
            setattr(self, "raw_input_%s_range" % k, np.ascontiguousarray(other_params[k], dtype=np.float64))
        self.raw_input_d = np.ascontiguousarray(d, dtype=np.float64)
        want = tuple(a.size for a in self._axes())
        if self.raw_input_d.shape != want:
            raise ValueError(f"table shape {self.raw_input_d.shape} does not match its axes {want}")
        self.Rdelta_sampling = bool(Rdelta_sampling)

    def _axes(self):
        return [self.raw_input_z_range, self.raw_input_M_range, self.raw_input_r_range] + \
               [getattr(self, "raw_input_%s_range" % k) for k in self.p_keys]

    def device_table(self, ctx):
        """bfg_table holding the LINEAR displacement table (interp_d of :322)"""
        return ctx.table(self._axes(), self.raw_input_d, log_values=False, cache_key=(self, "d", self.raw_input_d))

    def get_masses(self, model, r, M, a):
        raise NotImplementedError("Implement a get_masses() method first")

    # ---- table builder -----------------------------------------------------------------
    def setup_interpolator(self, z_min=1e-2, z_max=5, N_samples_z=30, z_linear_sampling=False,
                           M_min=1e12, M_max=1e16, N_samples_Mass=30,
                           R_min=1e-3, R_max=1e2, N_samples_R=100,
                           Rdelta_min=1e-3, Rdelta_max=10, Rdelta_sampling=False,
                           other_params={}, verbose=True, device=False):
        """BaryonCorrection.py:142-328.  device=True evaluates the DMO / DMB densities on the integration grid for
        every (z, params) slice on the host (the profile models are the caller's) and runs everything downstream of them
        -- Simpson, the PCHIPs, the monotonic mask, the inversion -- for all rows in one GPU launch
        (bfg_build_displacement_table)."""
        if z_min <= 0:
            assert z_linear_sampling, (f"Geometric series not possible for {z_min} < z < {z_max}. "
                                       "Set z_linear_sampling = True, or z_min > 0")
        M_range = np.geomspace(M_min, M_max, N_samples_Mass)
        r = np.geomspace(R_min, R_max, N_samples_R)
        z_range = np.linspace(z_min, z_max, N_samples_z) if z_linear_sampling else np.geomspace(z_min, z_max, N_samples_z)
        a_range = 1 / (1 + z_range)
        p_keys = list(other_params.keys())
        d_interp = np.zeros([z_range.size, M_range.size, r.size] + [np.size(other_params[k]) for k in p_keys])
        rdelta_range = np.geomspace(Rdelta_min, Rdelta_max, N_samples_R) if Rdelta_sampling else None
        combos = [p for p in product(*[np.arange(np.size(other_params[k])) for k in p_keys])]
        if device:
            d_interp = self._build_on_device(r, M_range, a_range, p_keys, other_params, combos, rdelta_range, d_interp)
        for j in range(z_range.size if not device else 0):
            for c in combos:
                for k_i, key in enumerate(p_keys):
                    _set_parameter(self.DMO, key, other_params[key][c[k_i]])
                    _set_parameter(self.DMB, key, other_params[key][c[k_i]])
                M_DMO = self.get_masses(self.DMO, r, M_range, a_range[j])
                M_DMB = self.get_masses(self.DMB, r, M_range, a_range[j])
                for i in range(M_range.size):
                    offset = _displacement_from_masses(r, M_DMO[i], M_DMB[i], np.log10(M_range[i]))
                    if Rdelta_sampling:                                  # :293-295
                        Rdelta = self.mass_def.get_radius(self.cosmo, M_range[i], a_range[j]) / a_range[j]
                        offset = np.interp(rdelta_range, r / Rdelta, offset)
                    d_interp[tuple([j, i, slice(None)] + list(c))] = offset
        input_rad = np.log(r) if not Rdelta_sampling else np.log(rdelta_range)
        self._set_table(np.log(1 + z_range), np.log(M_range), input_rad, d_interp, Rdelta_sampling,
                        {k: np.asarray(other_params[k], dtype=np.float64) for k in p_keys})

    def _build_on_device(self, r, M_range, a_range, p_keys, other_params, combos, rdelta_range, d_interp):
        from ..engine import get_context
        r_int, _ = _integration_grid(r, self.r_min_int, self.r_max_int, self.N_int)
        dens = {"DMO": [], "DMB": []}
        rdelta, where = [], []
        for j in range(a_range.size):
            for c in combos:
                for k_i, key in enumerate(p_keys):
                    _set_parameter(self.DMO, key, other_params[key][c[k_i]])
                    _set_parameter(self.DMB, key, other_params[key][c[k_i]])
                for name, model in (("DMO", self.DMO), ("DMB", self.DMB)):
                    dd = np.asarray(self._density(model, r_int, M_range, a_range[j]), dtype=np.float64)
                    dens[name].append(np.broadcast_to(dd, (M_range.size, r_int.size)))
                for i in range(M_range.size):
                    where.append(tuple([j, i, slice(None)] + list(c)))
                    if rdelta_range is not None:
                        rdelta.append(self.mass_def.get_radius(self.cosmo, M_range[i], a_range[j]) / a_range[j])
        d, status = get_context().build_displacement_table(
            self._geometry, r_int, np.concatenate(dens["DMO"], axis=0), np.concatenate(dens["DMB"], axis=0), r,
            np.asarray(rdelta) if rdelta_range is not None else None, rdelta_range)
        for row, index in enumerate(where):
            log10M = np.log10(M_range[index[1]])
            if status[row] & 8:
                raise ValueError(f"PchipInterpolator: unusable points in the mass profile of log10(M) = {log10M}")
            if status[row] & 1:
                warnings.warn(f"Mass profile of log10(M) = {log10M} is nearly constant over radius. Suggests density is "
                              "negative or zero for most of the range.", UserWarning)
            if status[row] & 2:
                warnings.warn(f"Mass profile of log10(M) = {log10M} is nearly constant over radius. Or it is broken. "
                              "Less than 5 datapoints are usable.", UserWarning)
            if status[row] & 4:
                warnings.warn(f"Displacement function for halo with log10(M) = {log10M} failed to compute. "
                              "Defaulting to d = 0.", UserWarning)
            d_interp[index] = d[row]
        return d_interp

    # ---- read-out ------------------------------------------------------------------------
    def _model_radius_com(self, M, a):
        """R_delta in comoving Mpc on the MODEL's cosmology and mass definition (:399)"""
        return self.mass_def.get_radius(self.cosmo, M, a) / a

    def _readout(self, r, M, a, **kwargs):
        """BaryonCorrection.py:331-419, the table lookup itself on the GPU"""
        from .. import _lib
        n_axes = 3 + len(self.p_keys)
        if n_axes > _lib.BFG_MAX_DIM:
            # more p_keys axes than the device read-out takes: scipy's N-linear interpolation on the host, literally the reference's
            # (:312-328); such a model runs through the runners' callable-model path, per halo
            from scipy import interpolate
            hit = getattr(self, "_host_interp", None)
            if hit is None or hit[0] is not self.raw_input_d:
                axes = [self.raw_input_z_range, self.raw_input_M_range, self.raw_input_r_range] + \
                       [getattr(self, "raw_input_%s_range" % k) for k in self.p_keys]
                hit = self._host_interp = (self.raw_input_d, interpolate.RegularGridInterpolator(
                    tuple(axes), self.raw_input_d, bounds_error=False, fill_value=np.nan))

            class _HostTable(object):
                eval = staticmethod(hit[1])
            table = _HostTable()
        else:
            from ..engine import get_context
            ctx = get_context()
            table = self.device_table(ctx)
        r_use, M_use = np.atleast_1d(r).astype(np.float64), np.atleast_1d(M).astype(np.float64)
        a_use = np.atleast_1d(a)
        z_use = 1 / a_use - 1
        displ = np.zeros([M_use.size, r_use.size])
        z_tab = np.exp(self.raw_input_z_range) - 1
        M_tab = np.exp(self.raw_input_M_range)
        r_tab = np.exp(self.raw_input_r_range)
        if (np.min(z_use) < np.min(z_tab)) | (np.max(z_use) > np.max(z_tab)):
            warnings.warn(f"Requested redshift range [{np.min(z_use)}, {np.max(z_use)}] outside table's range "
                          f"[{np.min(z_tab)}, {np.max(z_tab)}]", UserWarning)
        if (np.min(M_use) < np.min(M_tab)) | (np.max(M_use) > np.max(M_tab)):
            warnings.warn(f"Requested log_Mass range [{np.log10(np.min(M_use))}, {np.log10(np.max(M_use))}] outside "
                          f"table's range [{np.log10(np.min(M_tab))}, {np.log10(np.max(M_tab))}]", UserWarning)
        if not self.Rdelta_sampling:
            if (np.min(r_use) < np.min(r_tab)) | (np.max(r_use) > np.max(r_tab)):
                warnings.warn(f"Requested Radius range [{np.min(r_use)}, {np.max(r_use)}] outside table's range "
                              f"[{np.min(r_tab)}, {np.max(r_tab)}]", UserWarning)
        with np.errstate(all="ignore"):
            ones = np.ones_like(r_use)
            z_in = np.log(1 / a) * ones
            r_in = np.log(r_use)
            k_in = [kwargs[k] * ones for k in self.p_keys]
            for i in range(M_use.size):
                M_in = np.log(M_use[i]) * ones
                R = self._model_radius_com(M_use[i], a)
                rad = r_in if not self.Rdelta_sampling else r_in - np.log(R)
                vals = table.eval(np.stack([z_in, M_in, rad] + k_in, axis=1))
                displ[i] = np.where(r_use < self.epsilon_max * R, vals, 0)  # zero large-scale displacements
        if np.ndim(r) == 0:
            displ = np.squeeze(displ, axis=-1)
        if np.ndim(M) == 0:
            displ = np.squeeze(displ, axis=0)
        return displ

    def displacement(self, r, M, a, **kwargs):
        """Displacement d(r) in comoving Mpc for comoving radii r (:422-460)."""
        if not hasattr(self, "raw_input_d"):
            raise NameError("No Table created. Run setup_interpolator() method first")
        for k in self.p_keys:
            assert k in kwargs.keys(), "Need to provide %s as input into `displacement'. Table was built with this." % k
        return self._readout(r, M, a, **kwargs)


def _integration_grid(r, r_min_int, r_max_int, N_int):
    """the integration radii of get_masses (:669-675): the table range widened to [r_min_int, r_max_int] and by 20 %"""
    r_min = np.min([np.min(r), r_min_int])
    r_max = np.max([np.max(r), r_max_int])
    r_int = np.geomspace(r_min / 1.2, r_max * 1.2, N_int)
    return r_int, np.log(r_int[1] / r_int[0])


def _enclosed_mass(r, integrand_of_rint, density, r_min_int, r_max_int, N_int):
    """cumulative Simpson in ln r of a non-negative density, then log-log PCHIP onto r
    (shared by the 2D / 3D get_masses, BaryonCorrection.py:669-691 / :552-575)."""
    r_int, dlnr = _integration_grid(r, r_min_int, r_max_int, N_int)
    dens = density(r_int)
    dens = np.where(dens < 0, 0, dens)
    scalar = dens.ndim == 1
    if scalar:
        dens = dens[None, :]
    intgd = integrand_of_rint(r_int) * dens * dlnr
    M_enc = integrate.cumulative_simpson(intgd, axis=-1, initial=0) + intgd[:, [0]]
    lnr = np.log(r)
    M_f = np.zeros([M_enc.shape[0], r.size])
    with np.errstate(all="ignore"):
        for i in range(M_enc.shape[0]):
            good = (dens[i] > 0) & (np.isfinite(M_enc[i]))
            M_f[i] = np.exp(interpolate.PchipInterpolator(np.log(r_int)[good], np.log(M_enc[i])[good],
                                                          extrapolate=False)(lnr))
    return M_f[0] if scalar else M_f


class Baryonification2D(BaryonificationClass):
    """Projected (2D) displacement model: enclosed mass from Sigma(r) (BaryonCorrection.py:581-695)."""

    _geometry = 2

    def _density(self, model, r_int, M, a):
        # Sigma * a: ccl projects in comoving, not physical, coordinates (:676)
        return np.asarray(model.projected(self.cosmo, r_int, M, a)) * a

    def get_masses(self, model, r, M, a):
        return _enclosed_mass(r, lambda x: 2 * np.pi * x ** 2, lambda x: self._density(model, x, M, a),
                              self.r_min_int, self.r_max_int, self.N_int)


class Baryonification3D(BaryonificationClass):
    """3D displacement model: enclosed mass from rho(r) (BaryonCorrection.py:464-578)."""

    _geometry = 3

    def _density(self, model, r_int, M, a):
        return np.asarray(model.real(self.cosmo, r_int, M, a))

    def get_masses(self, model, r, M, a):
        return _enclosed_mass(r, lambda x: 4 * np.pi * x ** 3, lambda x: self._density(model, x, M, a),
                              self.r_min_int, self.r_max_int, self.N_int)
