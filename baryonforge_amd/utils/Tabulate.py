"""
Tabulated profile models, mirroring BaryonForge/utils/Tabulate.py:
`TabulatedProfile` (:99-391) and `ParamTabulatedProfile` (:395-730).

Same public surface (constructor, setup_interpolator kwargs, projected/real,
raw_input_* attributes, p_keys, NameError before the table exists), but the
table lives in HBM and is read out by the HIP kernels: by the shell runners
directly (the hot path), or through `projected()` / `real()` here, which run
the stand-alone batched read-out kernel (bfg_table_eval) with scipy
RegularGridInterpolator(method='linear', bounds_error=False) semantics.

The profile zoo that FILLS the tables (pyccl HaloProfiles, FFTLog) is out of
scope: `setup_interpolator` accepts any object with `real(cosmo, r, M, a)` and
`projected(cosmo, r, M, a)` -- e.g. a real BaryonForge/pyccl profile if the user
has those installed -- and `from_arrays` accepts precomputed tables (e.g. the
raw_input_* arrays of a table built with the reference).
"""
from itertools import product

import numpy as np

__all__ = ["_set_parameter", "_get_parameter", "TabulatedProfile", "ParamTabulatedProfile", "TabulatedCorrelation3D"]


def _is_profile_like(obj):
    return hasattr(obj, "projected") or hasattr(obj, "_projected") or hasattr(obj, "_real")


def _set_parameter(obj, key, value):
    """Tabulate.py:11-60: set `key` on obj and, recursively, on every profile-like attribute of obj."""
    for k in dir(obj):
        if k.startswith("__"):
            continue
        if k == key:
            setattr(obj, key, value)
        else:
            try:
                sub = getattr(obj, k)
            except Exception:
                continue
            if _is_profile_like(sub) and not callable(sub) and sub is not obj:
                _set_parameter(sub, key, value)


def _get_parameter(obj, key):
    """Tabulate.py:63-96: first attribute called `key` found on obj or its profile-like attributes."""
    for k in dir(obj):
        if k.startswith("__"):
            continue
        if k == key:
            return getattr(obj, key)
        try:
            sub = getattr(obj, k)
        except Exception:
            continue
        if _is_profile_like(sub) and not callable(sub) and sub is not obj:
            return _get_parameter(sub, key)


def _grids(z_min, z_max, N_samples_z, z_linear_sampling, M_min, M_max, N_samples_Mass, R_min, R_max, N_samples_R):
    M_range = np.geomspace(M_min, M_max, N_samples_Mass)
    r = np.geomspace(R_min, R_max, N_samples_R)
    z_range = np.linspace(z_min, z_max, N_samples_z) if z_linear_sampling else np.geomspace(z_min, z_max, N_samples_z)
    return z_range, M_range, r


class _TabulatedBase(object):
    """shared read-out plumbing"""

    p_keys = []

    def _axes(self):
        return [self.raw_input_z_range, self.raw_input_M_range, self.raw_input_r_range] + \
               [getattr(self, "raw_input_%s_range" % k) for k in self.p_keys]

    def _has_table(self):
        return hasattr(self, "raw_input_2D") and hasattr(self, "raw_input_3D")

    def device_table(self, ctx, which="2D"):
        """bfg_table holding ln(raw_input_<which>) -- what the reference's interp2D/interp3D hold (Tabulate.py:270-271)"""
        raw = self.raw_input_2D if which == "2D" else self.raw_input_3D
        key = (self, which, raw)
        def log_table():
            with np.errstate(all="ignore"):
                return np.log(raw)
        return ctx.table(self._axes(), log_table, log_values=True, cache_key=key)

    def _host_interpolator(self, which):
        """scipy RegularGridInterpolator over ln(raw_input_<which>) -- literally what the reference builds (Tabulate.py:270-271,
        :585-590).  Only for tables with more axes than the device read-out takes (BFG_MAX_DIM = 6, i.e. more than three p_keys):
        such a model runs through the runners' callable-model path, evaluated per halo on the host like any other callable."""
        from scipy import interpolate
        raw = self.raw_input_2D if which == "2D" else self.raw_input_3D
        hit = getattr(self, "_host_interp_cache", {}).get(which)
        if hit is None or hit[0] is not raw:
            with np.errstate(all="ignore"):
                f = interpolate.RegularGridInterpolator(tuple(self._axes()), np.log(raw), bounds_error=False, fill_value=np.nan)
            self._host_interp_cache = dict(getattr(self, "_host_interp_cache", {}), **{which: (raw, f)})
            hit = (raw, f)
        return hit[1]

    def _readout(self, r, M, a, which, **kwargs):
        """Tabulate.py:279-327 / :598-650 on the GPU."""
        from .. import _lib
        r_use, M_use = np.atleast_1d(r).astype(np.float64), np.atleast_1d(M).astype(np.float64)
        if len(self._axes()) > _lib.BFG_MAX_DIM:
            f = self._host_interpolator(which)
            prof = np.zeros([M_use.size, r_use.size])
            with np.errstate(all="ignore"):
                cols = [np.log(1 / a) * np.ones_like(r_use), None, np.log(r_use)] + [kwargs[k] * np.ones_like(r_use) for k in self.p_keys]
                for i in range(M_use.size):
                    cols[1] = np.log(M_use[i]) * np.ones_like(r_use)
                    prof[i] = np.exp(f(np.stack(cols, axis=1)))
            if np.ndim(r) == 0:
                prof = np.squeeze(prof, axis=-1)
            if np.ndim(M) == 0:
                prof = np.squeeze(prof, axis=0)
            return prof
        from ..engine import get_context
        ctx = get_context()
        table = self.device_table(ctx, which)
        prof = np.zeros([M_use.size, r_use.size])
        with np.errstate(all="ignore"):
            z_in = np.log(1 / a) * np.ones_like(r_use)
            r_in = np.log(r_use)
            k_in = [kwargs[k] * np.ones_like(r_use) for k in self.p_keys]
            for i in range(M_use.size):
                M_in = np.log(M_use[i]) * np.ones_like(r_use)
                prof[i] = table.eval(np.stack([z_in, M_in, r_in] + k_in, axis=1))
        if np.ndim(r) == 0:
            prof = np.squeeze(prof, axis=-1)
        if np.ndim(M) == 0:
            prof = np.squeeze(prof, axis=0)
        return prof


class TabulatedProfile(_TabulatedBase):
    """
    Tabulated (z, M, r) halo profile.  Mirrors Tabulate.py:99-391.

    >>> prof = TabulatedProfile(model, cosmo); prof.setup_interpolator(...)
    >>> prof.projected(cosmo, r, M, a)
    """

    def __init__(self, model, cosmo):
        self.model = model
        self.cosmo = cosmo
        self.mass_def = getattr(model, "mass_def", None)

    @classmethod
    def from_arrays(cls, ln1pz, lnM, lnr, raw_input_2D, raw_input_3D=None, cosmo=None, mass_def=None):
        """Build directly from a precomputed table (axes as stored by the reference: Tabulate.py:264-268)."""
        self = cls.__new__(cls)
        self.model, self.cosmo, self.mass_def = None, cosmo, mass_def
        self._set_table(ln1pz, lnM, lnr, raw_input_2D, raw_input_2D if raw_input_3D is None else raw_input_3D)
        return self

    def _set_table(self, ln1pz, lnM, lnr, t2d, t3d):
        self.raw_input_z_range = np.ascontiguousarray(ln1pz, dtype=np.float64)
        self.raw_input_M_range = np.ascontiguousarray(lnM, dtype=np.float64)
        self.raw_input_r_range = np.ascontiguousarray(lnr, dtype=np.float64)
        self.raw_input_2D = np.ascontiguousarray(t2d, dtype=np.float64)
        self.raw_input_3D = np.ascontiguousarray(t3d, dtype=np.float64)
        want = (self.raw_input_z_range.size, self.raw_input_M_range.size, self.raw_input_r_range.size)
        if self.raw_input_2D.shape != want or self.raw_input_3D.shape != want:
            raise ValueError(f"table shape {self.raw_input_2D.shape} does not match its axes {want}")

    def setup_interpolator(self, z_min=1e-2, z_max=5, N_samples_z=30, z_linear_sampling=False,
                           M_min=1e12, M_max=1e16, N_samples_Mass=30,
                           R_min=1e-3, R_max=1e2, N_samples_R=100,
                           other_params={}, verbose=True):
        """Tabulate.py:193-276: fill the (z, M, r) tables from model.real / model.projected."""
        z_range, M_range, r = _grids(z_min, z_max, N_samples_z, z_linear_sampling, M_min, M_max, N_samples_Mass,
                                     R_min, R_max, N_samples_R)
        interp3D = np.zeros([z_range.size, M_range.size, r.size])
        interp2D = np.zeros([z_range.size, M_range.size, r.size])
        for j in range(z_range.size):
            a_j = 1 / (1 + z_range[j])
            # the factor a: ccl projects in comoving, not physical, units (Tabulate.py:258-259)
            interp3D[j, :, :] = self.model.real(self.cosmo, r, M_range, a_j)
            interp2D[j, :, :] = self.model.projected(self.cosmo, r, M_range, a_j) * a_j
        self._set_table(np.log(1 + z_range), np.log(M_range), np.log(r), interp2D, interp3D)

    def real(self, cosmo, r, M, a):
        return self._real(cosmo, r, M, a)

    def projected(self, cosmo, r, M, a):
        return self._projected(cosmo, r, M, a)

    def _real(self, cosmo, r, M, a):
        if not self._has_table():
            raise NameError("No Table created. Run setup_interpolator() method first")
        return self._readout(r, M, a, "3D")

    def _projected(self, cosmo, r, M, a):
        if not self._has_table():
            raise NameError("No Table created. Run setup_interpolator() method first")
        return self._readout(r, M, a, "2D")


class ParamTabulatedProfile(_TabulatedBase):
    """
    Tabulated (z, M, r, *params) halo profile.  Mirrors Tabulate.py:395-730:
    `other_params` = {name: grid} adds table axes; the per-halo values come
    from catalog columns of the same names (HealpixRunner.py:456).
    """

    def __init__(self, model, cosmo):
        self.model = model
        self.cosmo = cosmo
        assert not isinstance(model, TabulatedProfile), "Input model cannot be 'TabulatedProfile' object."

    @classmethod
    def from_arrays(cls, ln1pz, lnM, lnr, raw_input_2D, raw_input_3D=None, other_params={}, cosmo=None):
        self = cls.__new__(cls)
        self.model, self.cosmo = None, cosmo
        self._set_table(ln1pz, lnM, lnr, other_params, raw_input_2D,
                        raw_input_2D if raw_input_3D is None else raw_input_3D)
        return self

    def _set_table(self, ln1pz, lnM, lnr, other_params, t2d, t3d):
        self.p_keys = list(other_params.keys())
        self.raw_input_z_range = np.ascontiguousarray(ln1pz, dtype=np.float64)
        self.raw_input_M_range = np.ascontiguousarray(lnM, dtype=np.float64)
        self.raw_input_r_range = np.ascontiguousarray(lnr, dtype=np.float64)
        for k in self.p_keys:
            setattr(self, "raw_input_%s_range" % k, np.ascontiguousarray(other_params[k], dtype=np.float64))
        self.raw_input_2D = np.ascontiguousarray(t2d, dtype=np.float64)
        self.raw_input_3D = np.ascontiguousarray(t3d, dtype=np.float64)
        want = tuple(a.size for a in self._axes())
        if self.raw_input_2D.shape != want or self.raw_input_3D.shape != want:
            raise ValueError(f"table shape {self.raw_input_2D.shape} does not match its axes {want}")

    def setup_interpolator(self, z_min=1e-2, z_max=5, N_samples_z=30, z_linear_sampling=False,
                           M_min=1e12, M_max=1e16, N_samples_Mass=30,
                           R_min=1e-3, R_max=1e2, N_samples_R=100,
                           other_params={}, verbose=True):
        """Tabulate.py:497-595"""
        z_range, M_range, r = _grids(z_min, z_max, N_samples_z, z_linear_sampling, M_min, M_max, N_samples_Mass,
                                     R_min, R_max, N_samples_R)
        p_keys = list(other_params.keys())
        shape = [z_range.size, M_range.size, r.size] + [np.size(other_params[k]) for k in p_keys]
        interp3D = np.zeros(shape) + np.nan
        interp2D = np.zeros(shape) + np.nan
        iterator = [p for p in product(*[np.arange(np.size(other_params[k])) for k in p_keys])]
        for j in range(z_range.size):
            a_j = 1 / (1 + z_range[j])
            for c in iterator:
                for k_i in range(len(p_keys)):
                    _set_parameter(self.model, p_keys[k_i], other_params[p_keys[k_i]][c[k_i]])
                index = tuple([j, slice(None), slice(None)] + list(c))
                interp3D[index] = self.model.real(self.cosmo, r, M_range, a_j)
                interp2D[index] = self.model.projected(self.cosmo, r, M_range, a_j) * a_j
        self._set_table(np.log(1 + z_range), np.log(M_range), np.log(r),
                        {k: np.asarray(other_params[k], dtype=np.float64) for k in p_keys}, interp2D, interp3D)

    def real(self, cosmo, r, M, a, **kwargs):
        if not self._has_table():
            raise NameError("No Table created. Run setup_interpolator() method first")
        for k in self.p_keys:
            assert k in kwargs.keys(), "Need to provide %s as input into `real'. Table was built with this." % k
        return self._readout(r, M, a, "3D", **kwargs)

    def projected(self, cosmo, r, M, a, **kwargs):
        if not self._has_table():
            raise NameError("No Table created. Run setup_interpolator() method first")
        for k in self.p_keys:
            assert k in kwargs.keys(), "Need to provide %s as input into `projected'. Table was built with this." % k
        return self._readout(r, M, a, "2D", **kwargs)


class TabulatedCorrelation3D(object):
    """
    Tabulated matter correlation function xi(r, a).  Mirrors Tabulate.py:733-784: a (ln(1+z), ln r) table of
    ccl.correlation_3d, read out as exp(bilinear(ln xi)) with NaN outside the table (RegularGridInterpolator,
    bounds_error=False).  The read-out runs on the GPU (bfg_table_eval); the table itself comes from
    `setup_interpolator` -- which needs a correlation function, pyccl's when it is installed or any
    `correlation_3d(cosmo, a, r)` callable -- or from `from_arrays`.
    """

    def __init__(self, cosmo, R_range=[1e-3, 1e3], N_samples=500, correlation_3d=None):
        self.cosmo = cosmo
        self.R_range = R_range
        self.N_samples = N_samples
        self.correlation_3d = correlation_3d

    @classmethod
    def from_arrays(cls, ln1pz, lnr, raw_input_3D, cosmo=None):
        self = cls(cosmo)
        self._set_table(ln1pz, lnr, raw_input_3D)
        return self

    def _set_table(self, ln1pz, lnr, xi):
        self.raw_input_z_range = np.ascontiguousarray(ln1pz, dtype=np.float64)
        self.raw_input_r_range = np.ascontiguousarray(lnr, dtype=np.float64)
        self.raw_input_3D = np.ascontiguousarray(xi, dtype=np.float64)
        if self.raw_input_3D.shape != (self.raw_input_z_range.size, self.raw_input_r_range.size):
            raise ValueError(f"table shape {self.raw_input_3D.shape} does not match its axes")

    def setup_interpolator(self, z_min=0, z_max=5, N_samples_z=10, verbose=False):
        """Tabulate.py:744-768"""
        xi_of = self.correlation_3d
        if xi_of is None:
            try:
                import pyccl as ccl
                xi_of = ccl.correlation_3d
            except ImportError as e:
                raise ImportError("TabulatedCorrelation3D.setup_interpolator needs pyccl.correlation_3d, or pass "
                                  "correlation_3d=callable(cosmo, a, r)") from e
        r = np.geomspace(self.R_range[0], self.R_range[1], self.N_samples)
        z_range = np.linspace(z_min, z_max, N_samples_z)
        interp3D = np.zeros([z_range.size, r.size]) + np.nan
        for j in range(z_range.size):
            interp3D[j, :] = xi_of(self.cosmo, 1 / (1 + z_range[j]), r)
        self._set_table(np.log(1 + z_range), np.log(r), interp3D)

    def device_table(self, ctx):
        """the (z, r) table as a (z, M, r) bfg_table with a two-node dummy mass axis (the read-out sits at its first node)"""
        def log_table():
            with np.errstate(all="ignore"):
                lnxi = np.log(self.raw_input_3D)
            return np.ascontiguousarray(np.repeat(lnxi[:, None, :], 2, axis=1))
        axes = [self.raw_input_z_range, np.array([0.0, 1.0]), self.raw_input_r_range]
        return ctx.table(axes, log_table, log_values=True, cache_key=(self, "xi", self.raw_input_3D))

    def __call__(self, r, a):
        """Tabulate.py:771-784"""
        if not hasattr(self, "raw_input_3D"):
            raise NameError("No Table created. Run setup_interpolator() method first")
        from ..engine import get_context
        r_use = np.atleast_1d(r).astype(np.float64)
        with np.errstate(all="ignore"):
            z_in = np.log(1 / a) * np.ones_like(r_use)               # log(1 + z)
            coords = np.stack([z_in, np.zeros_like(r_use), np.log(r_use)], axis=1)
        return self.device_table(get_context()).eval(coords)
