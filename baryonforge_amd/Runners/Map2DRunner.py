"""
Periodic Cartesian-grid runners, mirroring BaryonForge/Runners/Map2DRunner.py: `DefaultRunnerGrid` (:170-373),
`BaryonifyGrid` (:376-621), `PaintProfilesGrid` (:624-829) and the overlap regrid `regrid_pixels_2D/_3D` (:14-162).
Same constructors, attributes and `process() -> ndarray` of the map's shape; the per-halo python loops and the numba
regrid are replaced by bfg_paint_grid / bfg_baryonify_grid_offsets / bfg_regrid_grid (csrc/bfg_grid.hpp).  No CPU
fallback.  Ellipticity (use_ellipticity=True, the per-halo rotation matrices of :497-515) and PaintProfilesAnisGrid (:832-1064) are
built too (bfg_grid_args.d_rmat; golden fixture tests/golden/grid.npz).
"""
import numpy as np

from ..background import Background, MassDef, lcdm
from ..engine import emit_range_warnings, get_context
from ..utils.Tabulate import ParamTabulatedProfile, _get_parameter
from .HealpixRunner import _is_disp_table, _is_paint_table, _table_axes

__all__ = ["DefaultRunnerGrid", "BaryonifyGrid", "PaintProfilesGrid", "PaintProfilesAnisGrid", "regrid_pixels_2D",
           "regrid_pixels_3D"]


def _regrid_host(grid, pix_positions, pix_values):
    """host utility behind regrid_pixels_2D/_3D (API parity; the runners regrid on the GPU): the reference's candidate
    cells and periodic overlap rules (Map2DRunner.py:48-82, :122-162), vectorised over pixels"""
    N, nd = grid.shape[0], grid.ndim
    start = np.mod(np.asarray(pix_positions, dtype=np.float64), N)
    end = start + 1
    offs = np.arange(-2, 4)
    base = start.astype(np.int64)
    cells, ov = [], []
    for ax in range(nd):
        c = base[:, ax][:, None] + offs[None, :]
        valid = c < (end[:, ax].astype(np.int64) + 2)[:, None]
        c = np.where(c < 0, c + N, c)
        c = np.where(c + 1 > N, c % N, c)
        s, e = start[:, ax][:, None], end[:, ax][:, None]
        d = np.minimum(c + 1, e) - np.maximum(c, s)
        d = np.where(d < 0, np.minimum(c + 1, e + N) - np.maximum(c, s + N), d)
        d = np.where(d < 0, np.minimum(c + 1, e - N) - np.maximum(c, s - N), d)
        cells.append(c)
        ov.append(np.where(valid, d, -1.0))
    v = np.asarray(pix_values, dtype=np.float64)
    order = [1, 0] + ([2] if nd == 3 else [])                 # grid[i, j(, k)]: i from the y range, j from x, k from z
    shape = (v.size,) + (offs.size,) * nd
    w = v.reshape((-1,) + (1,) * nd) * np.ones(shape)
    ok = np.ones(shape, dtype=bool)
    idx = []
    for pos, ax in enumerate(order):
        sl = [slice(None)] + [None] * nd
        sl[pos + 1] = slice(None)
        w = w * ov[ax][tuple(sl)]
        ok &= (ov[ax][tuple(sl)] > 0)
        idx.append(np.broadcast_to(cells[ax][tuple(sl)], shape))
    np.add.at(grid, tuple(i[ok] for i in idx), w[ok])
    return grid


def regrid_pixels_2D(grid, pix_positions, pix_values):
    """grid[i, j] += overlap * value for every displaced unit pixel (Map2DRunner.py:14-82); modifies grid in place"""
    _regrid_host(grid, pix_positions, pix_values)


def regrid_pixels_3D(grid, pix_positions, pix_values):
    """3D version (Map2DRunner.py:86-162)"""
    _regrid_host(grid, pix_positions, pix_values)


class DefaultRunnerGrid(object):
    """Base class (Map2DRunner.py:170-373): catalog, gridded map, `epsilon_max`, `model`, mass definition."""

    def __init__(self, HaloNDCatalog, GriddedMap, epsilon_max, model, use_ellipticity=False, mass_def=None,
                 include_pixel_size=True, verbose=True):
        self.HaloNDCatalog = HaloNDCatalog
        self.GriddedMap = GriddedMap
        self.cosmo = HaloNDCatalog.cosmology
        self.model = model
        self.epsilon_max = epsilon_max
        self.mass_def = MassDef(200, "critical") if mass_def is None else mass_def
        self.verbose = verbose
        self.use_ellipticity = use_ellipticity
        self.include_pixel_size = include_pixel_size
        self.last_stats = None
        if use_ellipticity:                                                 # :272-278
            names = HaloNDCatalog.cat.dtype.names
            assert "q_ell" in names, "The 'q_ell' column is missing, but you set use_ellipticity = True"
            if not GriddedMap.is2D:
                assert "c_ell" in names, "The 'c_ell' column is missing, but you set use_ellipticity = True"
            assert "A_ell" in names, "The 'A_ell' column is missing, but you set use_ellipticity = True"

    def build_Rmat(self, A, q):
        """2D shear matrix of a halo with major-axis direction A and axis ratio q (Map2DRunner.py:281-350)"""
        A = np.array(A)
        A /= np.linalg.norm(A)
        if len(A) == 1:
            raise ValueError("Can't rotate a 1-dimensional vector")
        elif len(A) == 2:
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
        raise NotImplementedError("This method has not yet been verified. Use 2D ellipticity method instead")

    def _rmat_device(self, ctx):
        """build_Rmat of every halo, vectorised with the reference's dtypes (float32 catalogue columns); None if
        ellipticity is off.  3D maps: the reference raises inside its halo loop (:558 / :797)."""
        if not self.use_ellipticity:
            return None
        if not self.GriddedMap.is2D:
            if isinstance(self, PaintProfilesGrid):
                raise ValueError("use_ellipticity is not implemented for 3D maps")
            raise NotImplementedError("Currently not able to ellipticities with 3D maps.")
        cat = self.HaloNDCatalog.cat
        q = np.asarray(cat["q_ell"])
        assert np.all(q > 0), "The axis ratio in halo %d is not positive" % int(np.argmin(q))
        A = np.asarray(cat["A_ell"])
        A = A / np.sqrt(np.sum(A ** 2, axis=1))[:, None]                    # :478-480
        A = A / np.linalg.norm(A, axis=1)[:, None]                          # build_Rmat: A /= norm(A)
        beta = np.arccos(A[:, 0].astype(np.float64) * 1.0 + A[:, 1].astype(np.float64) * 0.0)
        with np.errstate(all="ignore"):
            eta = -np.log(q)
            etasq = eta * eta
            eta2g = np.where(eta > 1e-4, np.tanh(0.5 * eta) / np.where(eta == 0, 1, eta),
                             0.5 + etasq * ((-1 / 24) + etasq * (1 / 240))).astype(eta.dtype)
        g = (eta2g * eta) * np.exp(2j * beta)
        det = np.sqrt(1 - np.abs(g) ** 2)
        R = np.stack([(1 + g.real) / det, g.imag / det, g.imag / det, (1 - g.real) / det], axis=1)
        return ctx.to_device(R)

    def coord_array(self, *args):
        return np.vstack([a.flatten() for a in args]).T

    def pick_indices(self, center, width, Npix):
        """Map2DRunner.py:400-428"""
        inds = np.arange(center - width, center + width)
        inds = np.where(inds < 0, inds + Npix, inds)
        inds = np.where(inds >= Npix, inds - Npix, inds)
        return inds

    # ---- shared set-up --------------------------------------------------------------------------
    def _keys_checked(self):
        keys = vars(self.model).get("p_keys", []) if self.model is not None else []
        if len(keys) > 0:                                                   # :455-459 / :692-696
            txt = (f"You asked to use {keys} properties in Baryonification. You must pass a ParamTabulatedProfile"
                   f"as the model. You have passed {type(self.model)} instead")
            assert isinstance(self.model, ParamTabulatedProfile) or type(self.model).__name__ == "ParamTabulatedProfile", txt
        if len(keys) > 3:
            # the grid kernels read the table themselves: (z, M, r) + up to three extra axes; wider tables only exist as per-halo rows
            # on the shell path (csrc/bfg_ndtable.hpp).  Said here, before the table is uploaded, not as a status code afterwards.
            raise NotImplementedError(f"the periodic-grid runners read tables with up to 3 p_keys axes on the MI355X path; "
                                      f"this model has {len(keys)}: {list(keys)}")
        return list(keys)

    def _device_inputs(self, ctx, keys):
        hcat = self.HaloNDCatalog.cat
        # float32 catalogue columns (io.py:204): exact widening, but ln M is a float32 logarithm (Tabulate.py:316)
        with np.errstate(all="ignore"):
            lnM = np.log(hcat["M"]).astype(np.float64)
        cols = [hcat["M"].astype(np.float64), lnM, hcat["x"].astype(np.float64), hcat["y"].astype(np.float64),
                hcat["z"].astype(np.float64)] + [np.asarray(hcat[k], dtype=np.float64) for k in keys]
        halos = np.stack(cols, axis=1) if hcat.size else np.zeros((0, 5 + len(keys)))
        bins = np.asarray(self.GriddedMap.bins, dtype=np.float64)
        if self.GriddedMap.is2D and hcat.size:
            # the reference's 2D branches assert dx <= res and dy <= res per halo (Map2DRunner.py:522 / :761 / :936), with
            # dx = bins[argmin |bins - x|] - x: false only left of the grid by more than a pixel, or for a NaN position
            res = self.GriddedMap.res
            with np.errstate(all="ignore"):
                ok = (bins[0] - cols[2] <= res) & (bins[0] - cols[3] <= res)
            if not np.all(ok):
                j = int(np.argmin(ok))
                dx = bins[np.argmin(np.abs(bins - cols[2][j]))] - cols[2][j]
                dy = bins[np.argmin(np.abs(bins - cols[3][j]))] - cols[3][j]
                raise AssertionError("Halo offsets (%0.2f, %0.2f) are larger than res (%0.2f)" % (dx, dy, res))
        return ctx.to_device(halos), ctx.to_device(bins)


class PaintProfilesGrid(DefaultRunnerGrid):
    """Paint a tabulated profile around every halo onto the periodic grid (Map2DRunner.py:624-829): `model.projected`
    (raw_input_2D) on 2D maps, `model.real` (raw_input_3D) on 3D maps."""

    def process(self):
        assert self.model is not None, "You must provide a model"
        keys = self._keys_checked()
        if not _is_paint_table(self.model):
            if hasattr(self.model, "setup_interpolator"):
                raise NameError("No Table created. Run setup_interpolator() method first")
            raise TypeError(f"PaintProfilesGrid on the MI355X path needs a tabulated model; got {type(self.model)}")
        ctx = get_context()
        gm = self.GriddedMap
        ndim = 2 if gm.is2D else 3
        raw = self.model.raw_input_2D if gm.is2D else self.model.raw_input_3D

        def log_table():
            with np.errstate(all="ignore"):
                return np.log(np.asarray(raw, dtype=np.float64))
        table = ctx.table(_table_axes(self.model, keys), log_table, log_values=True,
                          cache_key=(self.model, "grid%dD" % ndim, raw))
        d_map = self._paint_device(ctx, table, keys)
        if self.include_pixel_size:
            d_map *= float(np.power(gm.res, ndim))                         # :826
        return get_context().to_host(d_map).reshape(np.shape(gm.map))

    def _paint_device(self, ctx, table, keys):
        """sum over halos of the table's profile in every cut-out window (:700-823), on the device"""
        gm = self.GriddedMap
        ndim = 2 if gm.is2D else 3
        d_halo, d_bins = self._device_inputs(ctx, keys)
        bg = Background(lcdm(self.cosmo))
        a = 1 / (1 + self.HaloNDCatalog.redshift)                          # :707
        d_rmat = self._rmat_device(ctx)            # keep the tensor alive until the kernels have been enqueued
        args = ctx.grid_args(ndim, gm.Npix, d_bins, d_halo, a, self.epsilon_max, ctx.massdef_struct(bg, self.mass_def),
                             n_extra=len(keys), d_rmat=d_rmat)
        d_map = ctx.zeros(int(np.prod(np.shape(gm.map))))                  # :685
        ctx.stats_reset()
        ctx.paint_grid(args, table, d_map)
        self.last_stats = ctx.stats()
        return d_map


class BaryonifyGrid(DefaultRunnerGrid):
    """Baryonify a MASS grid with a tabulated displacement model (Map2DRunner.py:376-621)."""

    def process(self):
        keys = self._keys_checked()
        if not _is_disp_table(self.model):
            if self.model is not None and hasattr(self.model, "displacement"):
                raise NameError("No Table created. Run setup_interpolator() method first")
            raise TypeError(f"BaryonifyGrid needs a BaryonificationClass model with a displacement table; got {type(self.model)}")
        ctx = get_context()
        gm, model = self.GriddedMap, self.model
        ndim = 2 if gm.is2D else 3
        table = ctx.table(_table_axes(model, keys), lambda: np.asarray(model.raw_input_d, dtype=np.float64),
                          log_values=False, cache_key=(model, "d", model.raw_input_d))
        d_halo, d_bins = self._device_inputs(ctx, keys)
        bg = Background(lcdm(self.cosmo))
        model_bg = Background(model.cosmo) if getattr(model, "cosmo", None) is not None else bg
        a = 1 / (1 + self.HaloNDCatalog.redshift)                          # :470
        d_rmat = self._rmat_device(ctx)            # keep the tensor alive until the kernels have been enqueued
        args = ctx.grid_args(ndim, gm.Npix, d_bins, d_halo, a, self.epsilon_max, ctx.massdef_struct(bg, self.mass_def),
                             model_md=ctx.massdef_struct(model_bg, getattr(model, "mass_def", None)),
                             model_epsilon_max=model.epsilon_max, rdelta_sampling=getattr(model, "Rdelta_sampling", False),
                             n_extra=len(keys), d_rmat=d_rmat)
        orig = np.ascontiguousarray(gm.map, dtype=np.float64)
        npx = orig.size
        d_off = ctx.zeros(npx, ndim)                                       # :450
        ctx.stats_reset()
        ctx.baryonify_grid_offsets(args, table, d_off)                     # :463-584
        self.last_stats = ctx.stats()
        emit_range_warnings(self.last_stats, "table")
        d_in = ctx.to_device(orig.reshape(-1))
        d_out = ctx.zeros(npx)
        ctx.regrid_grid(ndim, gm.Npix, d_off, d_in, d_out)                 # :586-613
        new_map = get_context().to_host(d_out).reshape(orig.shape)
        _, new_sum = ctx.absmax_sum(d_out)                                 # :616-619, summed on the device
        _, old_sum = ctx.absmax_sum(d_in)
        assert np.isclose(new_sum, old_sum), \
            "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]" % (new_sum, old_sum)
        return new_map


class PaintProfilesAnisGrid(PaintProfilesGrid):
    """
    Tracer-weighted painting on a 2D grid (Map2DRunner.py:833-1015): as `PaintProfilesAnisShell`, the per-pixel weights
    factor out of the halo sum -- two launches of the grid paint kernel (`Mtot_model` without pixel size; the ln-space
    product table of `model` x `Tracer_model`) and element-wise work on the GPU.
    """

    def __init__(self, HaloNDCatalog, GriddedMap, epsilon_max, model, Tracer_model, Mtot_model, background_val,
                 global_tracer_fraction, mass_def=None, include_pixel_size=True, use_ellipticity=False, verbose=True):
        self.Tracer_model = Tracer_model
        self.Mtot_model = Mtot_model
        self.background_val = background_val
        self.global_tracer_fraction = global_tracer_fraction
        super().__init__(HaloNDCatalog, GriddedMap, epsilon_max, model, use_ellipticity, mass_def, include_pixel_size, verbose)

    def process(self):
        import torch
        from .HealpixRunner import _ProductTable
        assert self.GriddedMap.is2D == True, "Can only paint tSZ on 2D maps. You have passed a 3D Map"   # noqa: E712 (:849)
        assert self.model is not None, "You must provide a model"
        keys = self._keys_checked()
        for m in (self.model, self.Tracer_model, self.Mtot_model):
            if not _is_paint_table(m):
                if hasattr(m, "setup_interpolator"):
                    raise NameError("No Table created. Run setup_interpolator() method first")
                raise TypeError(f"PaintProfilesAnisGrid on the MI355X path needs tabulated models; got {type(m)}")
        ctx = get_context()
        gm = self.GriddedMap
        res = gm.res
        with np.errstate(all="ignore"):
            mt = ctx.table(_table_axes(self.Mtot_model, keys), np.log(np.asarray(self.Mtot_model.raw_input_2D, dtype=np.float64)),
                           log_values=True)
        d_mtot = self._paint_device(ctx, mt, keys)                          # :866-871 (include_pixel_size = False)
        dL = 2 * _get_parameter(self.Mtot_model, "proj_cutoff")            # :876-878
        dV = np.power(res, 2) * dL
        rho_halos = float(d_mtot.mean().item()) / dL
        rho_m = float(Background(lcdm(self.cosmo)).rho_x(1.0, "matter"))          # comoving matter density (:886)
        drho_m = float(np.clip(rho_m - rho_halos, 0, None))
        d_mtot += dV * drho_m
        if self.verbose:
            print(f"Inputted halos contribute {100*(rho_halos/rho_m):0.2f}% of the total matter density.")
            print("Remaining density is assigned to a uniform background.")
        if rho_halos > rho_m:
            warnings.warn("Inputted halos contribute more mass than is available for this mean matter density."
                          "Your Mtot_model profiles are either too extended or you are using the wrong cosmology.")
        prod = _ProductTable(self.model, self.Tracer_model, keys)
        d_sum = self._paint_device(ctx, ctx.table(_table_axes(prod, keys), prod.ln_product, log_values=True), keys)
        d_orig = ctx.to_device(np.ascontiguousarray(gm.map, dtype=np.float64).reshape(-1))
        pos = d_mtot > 0
        safe = torch.where(pos, d_mtot, torch.ones_like(d_mtot))
        new_map = d_sum * torch.where(pos, d_orig / safe, torch.zeros_like(d_orig))
        new_map += (self.background_val * self.global_tracer_fraction) * \
            torch.where(pos, (dV * drho_m) / safe, torch.zeros_like(safe)) * d_orig
        if self.include_pixel_size:
            new_map *= float(np.power(res, 2))                             # :1009-1010
        return get_context().to_host(new_map).reshape(np.shape(gm.map))

