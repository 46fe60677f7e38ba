"""
Particle-snapshot runner, mirroring BaryonForge/Runners/SnapshotRunner.py: `DefaultRunnerSnapshot` (:11-158) and
`BaryonifySnapshot` (:162-275).  Same constructor, attributes and `process() -> structured particle catalog`.

The reference builds a periodic scipy KDTree over the particles and loops over halos in python (70-190 halos/s);
here one C-ABI call (bfg_baryonify_snapshot) bins the particles into a periodic cell grid on the GPU, walks each
halo's sphere with one wavefront and accumulates the radial displacements with f64 atomics.  No CPU fallback.
"""
import numpy as np

from ..background import Background, MassDef, lcdm
from ..engine import emit_range_warnings, get_context
from ..utils.Tabulate import ParamTabulatedProfile
from ..Profiles.BaryonCorrection import BaryonificationClass
from .HealpixRunner import _is_disp_table, _table_axes

__all__ = ["DefaultRunnerSnapshot", "BaryonifySnapshot"]


class DefaultRunnerSnapshot(object):
    """
    Base class (SnapshotRunner.py:11-158): holds the halo catalog, the particle snapshot, the cut-out size
    `epsilon_max` (in halo radii), the `model` and the mass definition.  Argument order as in the reference (:84-85):
    (..., model, mass_def, verbose, KDTree_kwargs).  `KDTree_kwargs` is accepted for signature parity and ignored: the
    neighbour search is a uniform cell grid on the GPU, built inside `process()`.
    """

    def __init__(self, HaloNDCatalog, ParticleSnapshot, epsilon_max, model, mass_def=None, verbose=True, KDTree_kwargs={}):
        self.HaloNDCatalog = HaloNDCatalog
        self.ParticleSnapshot = ParticleSnapshot
        self.epsilon_max = epsilon_max
        self.cosmo = HaloNDCatalog.cosmology
        self.model = model
        self.mass_def = MassDef(200, "critical") if mass_def is None else mass_def
        self.verbose = verbose
        self.last_stats = None

    def enforce_periodicity(self, dx):
        """SnapshotRunner.py:134-158"""
        L = self.ParticleSnapshot.L
        dx = np.where(dx > L / 2, dx - L, dx)
        dx = np.where(dx < -L / 2, dx + L, dx)
        return dx

    def compute_distance(self, *args):
        """periodic Euclidean distance (SnapshotRunner.py:102-131)"""
        d = 0
        for dx in args:
            d = d + self.enforce_periodicity(dx) ** 2
        return np.sqrt(d)


class BaryonifySnapshot(DefaultRunnerSnapshot):
    """Displace the particles around every halo with a tabulated displacement model (SnapshotRunner.py:162-275)."""

    def process_device(self):
        """returns the displaced, wrapped coordinates as a float64[n_part, ndim] torch tensor on the GPU"""
        keys = vars(self.model).get("p_keys", []) if self.model is not None else []
        if len(keys) > 0:                                                   # :203-209
            txt = (f"You asked to use {keys} properties in Baryonification. You must pass a ParamTabulatedProfile "
                   f"pr BaryonificationClass as the model. You have passed {type(self.model)} instead. "
                   f"If you did pass in a BaryonificationClass make sure you passed in addition params using "
                   f"the other_params option.")
            ok = isinstance(self.model, (ParamTabulatedProfile, BaryonificationClass)) or \
                type(self.model).__name__ in ("ParamTabulatedProfile", "BaryonificationClass", "Baryonification2D",
                                              "Baryonification3D")
            assert ok, txt
        if len(keys) > 3:                 # (before the table is uploaded: the snapshot kernels read (z, M, r) + up to three extra axes)
            raise NotImplementedError(f"BaryonifySnapshot reads tables with up to 3 p_keys axes on the MI355X path; this model "
                                      f"has {len(keys)}: {list(keys)}")
        if not _is_disp_table(self.model):
            if self.model is not None and hasattr(self.model, "displacement"):
                raise NameError("No Table created. Run setup_interpolator() method first")
            raise TypeError(f"BaryonifySnapshot needs a BaryonificationClass model with a displacement table; "
                            f"got {type(self.model)}")
        ctx = get_context()
        snap, hcat = self.ParticleSnapshot, self.HaloNDCatalog.cat
        is2D = snap.is2D
        ndim = 2 if is2D else 3
        recs = snap.records() if hasattr(snap, "records") else None        # the catalogue as float64[n, 4] (M, x, y, z): a view
        part = None if recs is not None else \
            np.stack([np.asarray(snap.cat[c], dtype=np.float64) for c in ("x", "y", "z")[:ndim]], axis=1)
        # halo columns are float32 in the reference (io.py:204): positions and masses widen exactly, but the table
        # coordinate is np.log of the float32 mass, i.e. a float32 logarithm (BaryonCorrection.py:397)
        with np.errstate(all="ignore"):
            lnM = np.log(hcat["M"]).astype(np.float64)
        cols = [hcat["M"].astype(np.float64), lnM, hcat["x"].astype(np.float64), hcat["y"].astype(np.float64),
                hcat["z"].astype(np.float64)] + [np.asarray(hcat[k], dtype=np.float64) for k in keys]
        halos = np.stack(cols, axis=1) if hcat.size else np.zeros((0, 5 + len(keys)))
        a = 1 / (1 + self.HaloNDCatalog.redshift)                           # :219
        bg = Background(lcdm(self.cosmo))        # the reference builds this cosmology without w0 (:197-200): w0 = -1
        model = self.model
        table = ctx.table(_table_axes(model, list(keys)), lambda: np.asarray(model.raw_input_d, dtype=np.float64),
                          log_values=False, cache_key=(model, "d", model.raw_input_d))
        model_bg = Background(model.cosmo) if getattr(model, "cosmo", None) is not None else bg
        d_halo = ctx.to_device(halos)
        if recs is not None:
            # one upload of the records as they lie in memory; the displaced coordinates are written into a device copy of
            # them, which IS the new catalogue (no column gathering on the host, no per-column write-back)
            d_recs = ctx.to_device(recs)
            self._d_new_records = d_recs.clone()
            d_part, d_out = d_recs[:, 1:1 + ndim], self._d_new_records[:, 1:1 + ndim]
        else:
            self._d_new_records = None
            d_part = ctx.to_device(part)
            d_out = ctx.zeros(part.shape[0], ndim)
        ctx.stats_reset()
        ctx.baryonify_snapshot(d_part, d_halo, ndim, snap.L, a, self.epsilon_max, ctx.massdef_struct(bg, self.mass_def),
                               ctx.massdef_struct(model_bg, getattr(model, "mass_def", None)), model.epsilon_max,
                               getattr(model, "Rdelta_sampling", False), len(keys), table, d_out)
        self.last_stats = ctx.stats()
        emit_range_warnings(self.last_stats, "table")                       # BaryonCorrection.py:382-394
        return d_out

    def process(self):
        """returns new_cat : the particle catalog (structured array) with displaced, box-wrapped coordinates"""
        d_new = self.process_device()
        if self._d_new_records is not None:
            recs = get_context().to_host(self._d_new_records)
            self._d_new_records = None
            return recs.view(self.ParticleSnapshot.cat.dtype).reshape(-1)   # the record matrix seen as the structured array
        new = get_context().to_host(d_new)
        new_cat = self.ParticleSnapshot.cat.copy()                          # :260
        new_cat["x"] = new[:, 0]
        new_cat["y"] = new[:, 1]
        if not self.ParticleSnapshot.is2D:
            new_cat["z"] = new[:, 2]
        return new_cat
