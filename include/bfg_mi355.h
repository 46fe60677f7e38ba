/*
 * bfg_mi355.h -- C-ABI of libbfg_mi355.so: the MI355X (gfx950) implementation of
 * BaryonForge's per-halo shell paint / baryonify hot path.
 *
 * The reference (DhayaaAnbajagane/BaryonForge) is pure Python and has NO FFI
 * boundary of its own; its boundary is the Python class API
 * (bfg.Runners / bfg.Profiles / bfg.utils).  This header is the boundary the
 * build inserts beneath those classes: every entry point below names the
 * reference code it replaces (paths relative to the reference checkout,
 * BaryonForge/...).  The Python mirror of the reference classes lives in
 * baryonforge_amd/ and binds this library with ctypes.
 *
 * Conventions
 *   - plain C types, caller-owned buffers, no torch / numpy types;
 *   - every function returns BFG_OK (0) or a negative bfg_status;
 *   - pointers named d_* are DEVICE pointers (HBM of the context's GPU),
 *     everything else is host memory that is only read during the call;
 *   - all work is enqueued on the context's stream (the caller's hipStream_t
 *     if one was given to bfg_ctx_create) and is asynchronous unless stated;
 *   - maps are HEALPix RING-ordered float64[12*nside^2] like LightconeShell.map
 *     (utils/io.py:341-353); catalogs are the float64 records of
 *     HaloLightConeCatalog.cat (utils/io.py:56-75): (M, z, ra, dec, extras...).
 */
#ifndef BFG_MI355_H
#define BFG_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFG_ABI_VERSION 6

typedef enum {
    BFG_OK = 0,
    BFG_ERR_INVALID = -1,      /* bad argument (shape, nside, null pointer)            */
    BFG_ERR_HIP = -2,          /* a HIP runtime call failed; see bfg_last_error()      */
    BFG_ERR_NO_DEVICE = -3,    /* no usable gfx950 device                              */
    BFG_ERR_UNSUPPORTED = -4,  /* e.g. more table dimensions than BFG_MAX_DIM          */
    BFG_ERR_NOMEM = -5,
    BFG_ERR_COMM = -6          /* RCCL missing or a collective failed; see bfg_last_error() */
} bfg_status;

#define BFG_MAX_DIM 6          /* (ln(1+z), ln M, ln r) + up to 3 extra p_keys axes: what the kernels read directly; the shell
                                * calls take wider tables too (up to 13 dimensions, see bfg_table_create) */
#define BFG_MAX_EXTRA (BFG_MAX_DIM - 3)

typedef struct bfg_ctx bfg_ctx;        /* one per GPU / stream                         */
typedef struct bfg_table bfg_table;    /* an interpolation table resident in HBM       */
typedef struct bfg_spline bfg_spline;  /* the D_A(z) cubic spline resident in HBM      */

/* ---- library / context ------------------------------------------------------- */
int bfg_abi_version(void);
const char *bfg_status_string(int status);
const char *bfg_last_error(void);                 /* thread-local detail of the last BFG_ERR_HIP / BFG_ERR_COMM */
int bfg_device_count(int *count);

/* stream: the hipStream_t to enqueue on -- e.g. torch's current stream; NULL is the legacy
 * default stream (which IS torch's default stream) -- or BFG_STREAM_OWN to let the context
 * create and own a non-blocking stream of its own. */
#define BFG_STREAM_OWN ((void *)(intptr_t)-1)
int bfg_ctx_create(int device_id, void *stream, bfg_ctx **out);
int bfg_ctx_destroy(bfg_ctx *ctx);
/* Re-bind the context to another stream of its GPU (a caller whose "current stream" changes between calls, e.g. inside
 * `with torch.cuda.stream(s)`, passes the stream it is on before every call).  The new stream is made to wait (an event,
 * no host synchronisation) for what the context enqueued on the old one: the context's workspaces are shared by
 * consecutive calls.  A stream the context created itself (BFG_STREAM_OWN) is drained and destroyed. */
int bfg_ctx_set_stream(bfg_ctx *ctx, void *stream);
int bfg_ctx_synchronize(bfg_ctx *ctx);
/* device name, CU count, LDS per CU etc. (for bench/DESIGN bookkeeping) */
int bfg_ctx_device_info(bfg_ctx *ctx, char *name, int name_len, int *n_cu, int *lds_per_cu_bytes,
                        int64_t *hbm_bytes);

/* device-memory helpers so that a pure-C / pure-ctypes caller needs no other allocator */
int bfg_dev_malloc(bfg_ctx *ctx, size_t bytes, void **d_ptr);
int bfg_dev_free(bfg_ctx *ctx, void *d_ptr);
int bfg_memcpy_h2d(bfg_ctx *ctx, void *d_dst, const void *src, size_t bytes);   /* synchronous */
int bfg_memcpy_d2h(bfg_ctx *ctx, void *dst, const void *d_src, size_t bytes);   /* synchronous */
int bfg_dev_memset_zero(bfg_ctx *ctx, void *d_ptr, size_t bytes);               /* async        */

/* ---- tables -------------------------------------------------------------------
 * Replaces the scipy RegularGridInterpolator objects the reference builds in
 *   utils/Tabulate.py:270-271 (TabulatedProfile.interp2D/interp3D),
 *   utils/Tabulate.py:589-590 (ParamTabulatedProfile),
 *   Profiles/BaryonCorrection.py:322 (BaryonificationClass.interp_d)
 * and evaluates them with the same semantics (method='linear',
 * bounds_error=False, fill_value=nan; cell = largest i with axis[i] <= x,
 * clipped to [0, n-2]).
 * axes[d] has shape[d] strictly increasing entries; axis order is
 * (ln(1+z), ln M, ln r, extras...), values is C-ordered with that shape.
 * BFG_TABLE_LOG_VALUES: values are ln(T) and the read-out is exp(interp)
 *   (Tabulate.py:314-315); otherwise the read-out is the interpolant itself
 *   (BaryonCorrection.py:403-408).
 * ndim <= BFG_MAX_DIM: the shell, grid and snapshot kernels read the table directly.
 * BFG_MAX_DIM < ndim <= 13 (four to ten p_keys axes; the reference's tables are
 *   N-dimensional): accepted by bfg_paint_shell* and bfg_baryonify_offsets* only -- every
 *   halo's radial row is blended first (the non-radial coordinates of a query are the
 *   halo's), then the same shell kernels run on the rows, in batches of halos whose rows
 *   stay under BFG_ND_ROW_BYTES (environment, default 4 GiB); bfg_table_eval and the grid /
 *   snapshot calls return BFG_ERR_UNSUPPORTED for such a table.  (The shell calls take the
 *   same row path for a 6-dimensional displacement table, where it is the faster of the
 *   two; such a table keeps its directly readable form for every other call.)      */
#define BFG_TABLE_LOG_VALUES 1u
int bfg_table_create(bfg_ctx *ctx, int ndim, const int64_t *shape, const double *const *axes,
                     const double *values, uint32_t flags, bfg_table **out);
int bfg_table_destroy(bfg_ctx *ctx, bfg_table *table);

/* Batched stand-alone read-out (the model.projected()/displacement() call of
 * HealpixRunner.py:472 / :345 on its own): coords is host [npts][ndim], out host [npts].
 * Synchronous; meant for tests and for users calling the model directly.          */
int bfg_table_eval(bfg_ctx *ctx, const bfg_table *table, int64_t npts, const double *coords,
                   double *out);

/* ---- D_A(z) spline --------------------------------------------------------------
 * Replaces D_a = CubicSpline(z_t, ccl.angular_diameter_distance(...)) of
 * HealpixRunner.py:297-299 / :429-431: n knots, coef is scipy's PPoly layout
 * c[4][n-1] (highest power first), evaluated per halo on the GPU.                  */
int bfg_spline_create(bfg_ctx *ctx, int n_knots, const double *knots, const double *coef,
                      bfg_spline **out);
int bfg_spline_destroy(bfg_ctx *ctx, bfg_spline *spline);

/* ---- per-halo radius: ccl MassDef.get_radius on a flat wCDM background ----------
 * R = (M / (4.18879020479 * Delta * rho_x(a)))^(1/3)  [physical Mpc]
 * (HealpixRunner.py:320, :454; BaryonCorrection.py:399), with
 * rho_crit(a) = rho_crit0_h2 * h^2 * E^2(a),
 * E^2(a) = Omega_m a^-3 + Omega_l a^(-3(1+w0)) + Omega_r a^-4.                     */
typedef struct {
    double Omega_m, Omega_l, Omega_r, w0, h;
    double rho_crit0_h2;     /* 3 (100 km/s/Mpc)^2 / (8 pi G) in Msun/Mpc^3              */
    double Delta;            /* overdensity                                              */
    int32_t rho_type;        /* 0 = 'critical', 1 = 'matter'                             */
    int32_t reserved;
} bfg_massdef;

/* ---- the hot path ---------------------------------------------------------------- */
typedef struct {
    int64_t nside;
    int64_t n_halo;
    const double *d_catalog;   /* device, [n_halo][cat_stride] float64 records (M, z, ra, dec, extras) */
    int32_t cat_stride;        /* doubles per record = 4 + n_extra                        */
    int32_t n_extra;           /* number of p_keys columns used as extra table coordinates */
    double epsilon_max;        /* runner cut-out radius in units of R (DefaultRunner)     */
    bfg_massdef runner_md;     /* runner.mass_def on the catalog cosmology                */
    bfg_massdef model_md;      /* model.mass_def on the model cosmology (baryonify only)  */
    double model_epsilon_max;  /* BaryonificationClass.epsilon_max (baryonify only)       */
    int32_t rdelta_sampling;   /* BaryonificationClass.Rdelta_sampling                    */
    int32_t include_pixel_size;/* PaintProfilesShell only                                 */
    int32_t variant;           /* BFG_VARIANT_*                                           */
    int32_t flags;             /* BFG_SHELL_* bits                                        */
} bfg_shell_args;

/* The caller vouches that the output (d_map / d_offsets) is all zeros on entry -- the reference's runners always start
 * from np.zeros (Runners/HealpixRunner.py:313, :424).  The tile kernels then store their tiles instead of
 * read-modify-writing them (one global round trip less at the end of every tile).  Without the flag the calls
 * accumulate INTO whatever the buffer holds.                                                                    */
#define BFG_SHELL_OUT_IS_ZERO 1
/* The output buffer is UNINITIALISED memory and the call defines every element of it: pixels no halo touches become 0.
 * On the tile path the kernels write each sky tile exactly once, so no separate clearing pass over the 101 MB map (302 MB
 * of offsets) is needed; on every other path the library clears the buffer itself first.                        */
#define BFG_SHELL_OUT_OVERWRITE 2
/* One plan, K models (ABI 6).  The reference's workflow paints several models over ONE catalog (examples/05_Paint_tSZ_shell.ipynb
 * :303-324, utils/Parallelize.py:92-113: a list of runners that differ in `model` only).  The per-halo records and the halo -> sky
 * tile pair lists a shell call builds depend on the catalog, NSIDE, epsilon, the mass definitions, the D_A spline and the table's
 * AXES -- not on its values.  With this flag the caller vouches that d_catalog holds the same records as in the context's previous
 * shell call; if that call (same kind: paint / offsets) was built from the same everything-else -- the library compares pointers,
 * sizes, scalars and a hash of the table axes -- the call reuses its records and pair lists and runs only the tile kernels (one
 * small launch re-arms their work counters).  Otherwise the flag is ignored and the call does all of its work.  Results are those
 * of a call without the flag (the same kernels on the same records); bfg_plan_reuses() counts the calls that took the short cut. */
#define BFG_SHELL_REUSE_PLAN 4

#define BFG_VARIANT_AUTO 0
#define BFG_VARIANT_SCATTER_WAVE 1     /* one 64-lane wavefront per halo, global f64 atomics   */
#define BFG_VARIANT_SCATTER_QUARTER 2  /* one 16-lane group per halo, global f64 atomics       */
#define BFG_VARIANT_TILE_LDS 3         /* sky-tile privatised accumulation in LDS              */

/* BaryonifySnapshot.process (Runners/SnapshotRunner.py:176-275): for every halo, the particles within
 * min(epsilon_max R / a, L / 2) (periodic box) are displaced radially by model.displacement(d, M, a)
 * (Profiles/BaryonCorrection.py:331-419, linear table as for bfg_baryonify_offsets); positions are shifted
 * and wrapped into [0, L] once at the end.  Replaces the scipy KDTree + python halo loop.
 * d_halo rows: (M, ln M as used for the table -- the reference's float32 catalogue gives a float32
 * logarithm --, x, y, z [0 in 2D], extras...).  bfg_stats.pixel_updates counts (halo, particle) pairs.   */
typedef struct {
    int32_t ndim;              /* 2 or 3                                                          */
    int32_t rdelta_sampling;   /* model.Rdelta_sampling (BaryonCorrection.py:406-408)             */
    int64_t n_part, n_halo;
    double L;                  /* box size, comoving Mpc                                          */
    double a;                  /* scale factor of the snapshot                                    */
    const double *d_part;      /* device, float64[n_part][ndim]                                   */
    const double *d_halo;      /* device, float64[n_halo][halo_stride]                            */
    int32_t halo_stride;       /* >= 5 + n_extra                                                  */
    int32_t n_extra;           /* extra table coordinates per halo (p_keys)                       */
    double epsilon_max;        /* runner cut in halo radii (SnapshotRunner.py:223)                */
    bfg_massdef runner_md;     /* mass definition of the runner (:222)                            */
    bfg_massdef model_md;      /* mass definition of the model (BaryonCorrection.py:399)          */
    double model_epsilon_max;  /* model.epsilon_max (:410)                                        */
} bfg_snapshot_args;
int bfg_baryonify_snapshot(bfg_ctx *ctx, const bfg_snapshot_args *args, const bfg_table *table,
                           double *d_out /* device, float64[n_part][ndim] */);
/* The same on particle RECORDS: coordinate k of particle i is read at d_part[i * part_stride + k] and written at
 * d_out[i * out_stride + k] (strides in doubles, >= ndim).  ParticleSnapshot.cat (utils/io.py:497-560) is a packed
 * float64 record array (M, x, y, z): with d_part = records + 1, part_stride = 4 the catalogue is used as it lies in
 * memory, and a copy of the records with d_out = copy + 1, out_stride = 4 becomes the new catalogue.             */
int bfg_baryonify_snapshot_strided(bfg_ctx *ctx, const bfg_snapshot_args *args, const bfg_table *table,
                                   double *d_out, int64_t part_stride, int64_t out_stride);

/* Periodic Cartesian grid runners (Runners/Map2DRunner.py), no ellipticity.  d_halo rows as for
 * bfg_baryonify_snapshot: (M, ln M as used for the table, x, y, z [0 in 2D], extras...); d_bins = the npix pixel
 * centres of GriddedMap.bins (utils/io.py:450-470).
 * bfg_paint_grid            the halo loop of PaintProfilesGrid.process (:700-823): d_map[npix^ndim] += profile in every
 *                           halo's cut-out window (log table: raw_input_2D for 2D maps, raw_input_3D for 3D maps); the
 *                           pixel-size factor (:826) is left to the caller.
 * bfg_baryonify_grid_offsets the halo loop of BaryonifyGrid.process (:463-584): d_offsets[npix^ndim][ndim] += radial
 *                           displacement / res; non-finite contributions are added as the reference adds them.
 * bfg_regrid_grid           :586-613 + regrid_pixels_2D/_3D (:14-162): non-finite offsets -> 0, pixel positions
 *                           index + offset, overlap-weighted deposit of d_in_map INTO d_out_map.                     */
typedef struct {
    int32_t ndim;              /* 2 or 3                                                          */
    int32_t rdelta_sampling;
    int64_t n_halo;
    int32_t npix;              /* pixels per side                                                 */
    int32_t reserved;
    double a;                  /* scale factor of the map                                         */
    const double *d_bins;      /* device, float64[npix] ascending, uniform                        */
    const double *d_halo;      /* device, float64[n_halo][halo_stride]                            */
    int32_t halo_stride;       /* >= 5 + n_extra                                                  */
    int32_t n_extra;
    double epsilon_max;
    bfg_massdef runner_md;
    bfg_massdef model_md;
    double model_epsilon_max;
    const double *d_rmat;      /* device, float64[n_halo][4] = build_Rmat(A_ell, q_ell) row-major (Map2DRunner.py:281-350):
                                * 2D use_ellipticity; NULL = circular                                               */
} bfg_grid_args;
int bfg_paint_grid(bfg_ctx *ctx, const bfg_grid_args *args, const bfg_table *table, double *d_map);
int bfg_baryonify_grid_offsets(bfg_ctx *ctx, const bfg_grid_args *args, const bfg_table *table, double *d_offsets);
int bfg_regrid_grid(bfg_ctx *ctx, int ndim, int npix, const double *d_offsets, const double *d_in_map,
                    double *d_out_map);

/* Mass map of a particle set on a periodic N^ndim grid, accumulated INTO d_grid (float64[N^ndim], C order):
 * mode BFG_DEPOSIT_NGP = ParticleSnapshot.make_map (utils/io.py:629-677, numpy.histogramdd on
 * linspace(0, L, N + 1)); BFG_DEPOSIT_CIC = cloud-in-cell on cell centres, periodic.
 * d_mass may be NULL (unit masses).                                                          */
#define BFG_DEPOSIT_NGP 0
#define BFG_DEPOSIT_CIC 1
int bfg_deposit_grid(bfg_ctx *ctx, int ndim, int64_t n_part, const double *d_pos, const double *d_mass,
                     double L, int n_grid, int mode, double *d_grid);
/* ... on particle records: position k of particle i at d_pos[i * pos_stride + k], its mass at d_mass[i * mass_stride]. */
int bfg_deposit_grid_strided(bfg_ctx *ctx, int ndim, int64_t n_part, const double *d_pos, int64_t pos_stride,
                             const double *d_mass, int64_t mass_stride, double L, int n_grid, int mode, double *d_grid);

/* Displacement-table builder on the device: the arithmetic of BaryonificationClass.setup_interpolator
 * (Profiles/BaryonCorrection.py:225-304) with get_masses (:669-691 Baryonification2D, :552-575 Baryonification3D)
 * inlined, for n_rows = all (z, params, M) rows of a table in one launch.  The DMO / DMB profile models stay with the
 * caller (they are the reference's pyccl profile zoo): it supplies their densities on the integration grid,
 *   d_dens_dmo / d_dens_dmb   device float64[n_rows][n_int]: Sigma(r_int) * a (geometry 2) or rho(r_int) (geometry 3),
 *   r_int                     host float64[n_int], the geomspace grid of :672 (n_int >= 3),
 *   r                         host float64[nr], the table radii (:209),
 *   rdelta, rdelta_range      host float64[n_rows] R_delta (comoving) per row and float64[nr] r / R_delta axis for
 *                             Rdelta_sampling tables (:293-295), or both NULL,
 * and receives d_out = device float64[n_rows][nr] displacements and status = host int32[n_rows] with BFG_BUILD_* bits
 * (the reference's UserWarnings; BFG_BUILD_ERROR where scipy's PchipInterpolator would raise ValueError).
 * Synchronous.  Steps: clip negative densities, integrand * dlnr, scipy cumulative_simpson + first term, log-log PCHIP
 * onto r (extrapolate = False), iterative monotonic mask, r(ln M_DMB) and ln M_DMO(ln r) PCHIPs, exp(...) - r,
 * non-finite -> 0, optional np.interp onto rdelta_range.                                                          */
#define BFG_BUILD_CONSTANT 1
#define BFG_BUILD_FEW 2
#define BFG_BUILD_ZERO 4
#define BFG_BUILD_ERROR 8
int bfg_build_displacement_table(bfg_ctx *ctx, int geometry, int n_rows, int n_int, const double *r_int,
                                 const double *d_dens_dmo, const double *d_dens_dmb, int nr, const double *r,
                                 const double *rdelta, const double *rdelta_range, double *d_out, int32_t *status);

/* Counters the kernels maintain (device side), fetched with bfg_stats_read. */
typedef struct {
    uint64_t pixel_updates;   /* P_tot = sum_j |disc_j| (incl. the 4-neighbour fallback)  */
    uint64_t halos_out_of_table; /* halos whose (z, M, extras) lie outside the table hull */
    uint64_t pixels_out_of_table;/* pixel queries whose radius lies outside the r axis     */
    uint64_t halos_fallback4; /* baryonify: halos that used the <4-pixel fallback          */
    uint32_t warn_mask;       /* BFG_WARN_* bits (BaryonCorrection.py:382-394)             */
    uint32_t halos_scatter_fallback; /* tile variant: halos WITH work that were handed to the ~12x slower global-atomic
                               * scatter kernel (disc over > 64 sky tiles, ln(pixarea D^2) outside the fast exp range, or
                               * the whole call after a pair-buffer overflow); results are the same, the time is not    */
} bfg_stats;
#define BFG_WARN_Z_RANGE 1u
#define BFG_WARN_M_RANGE 2u
#define BFG_WARN_R_RANGE 4u

/* PaintProfilesShell.process loop (Runners/HealpixRunner.py:449-481):
 * d_map[pix] += profile, for every halo and every pixel of its disc.
 * d_map is float64[12 nside^2] on the device and is accumulated INTO (zero it
 * first for the reference's behaviour, :424).                                        */
int bfg_paint_shell(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_table *table,
                    const bfg_spline *da_spline, double *d_map);

/* BaryonifyShell.process halo loop (Runners/HealpixRunner.py:315-355) with
 * BaryonificationClass._readout (Profiles/BaryonCorrection.py:331-419) inlined:
 * d_offsets[pix][0..2] += displaced unit vector - pixel unit vector.
 * d_offsets is float64[12 nside^2][3], accumulated INTO.                             */
int bfg_baryonify_offsets(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_table *table,
                          const bfg_spline *da_spline, double *d_offsets);

/* The same two loops with the output handed over IN SLICES while the call is still being enqueued -- the building block of
 * the multi-GPU join inside one process() call.  The sky tiles are cut into n_slices runs of whole ring bands (n_slices is
 * clamped to 16 and to the number of bands); the tile kernel is launched once per run and after each launch the library calls
 *     fn(user, slice, n, elem_begin, elem_end)
 * on the calling thread: once the context's stream reaches this point, d_out[elem_begin .. elem_end) (elements = doubles;
 * contiguous RING pixel ranges, x 3 for the offset field) holds its final values, so the callback can start that part of the
 * exchange on another stream -- e.g. bfg_allreduce_f64_begin(ctx, d_out + elem_begin, elem_end - elem_begin, &ticket), which
 * replaces, slice by slice, the parent-side np.sum(outputs, axis=0) of utils/Parallelize.py:318 -- while the next slice is
 * painted.  The slices cover the output exactly once, in ascending order, and their number and ranges depend on (nside,
 * which of the two loops, n_slices) ONLY -- never on the catalog: the ranks of a process group issue one collective per
 * callback, so a rank whose shard is empty, or whose call cannot be cut (the scatter variants), reports the same ranges as
 * its peers (all of them after its last launch).  A non-zero return of fn aborts the call with BFG_ERR_INVALID.  Halos the
 * tile path leaves to the scatter kernel are handled BEFORE the first slice, so every slice is final when it is reported:
 * with BFG_SHELL_OUT_OVERWRITE into an output the library clears first (the tiles are then added to it); without the flag
 * they -- like the tiles -- are added to what the buffer holds (accumulate INTO; BFG_SHELL_OUT_IS_ZERO: to the zeros the
 * caller vouched for).                                                                                                 */
typedef int (*bfg_slice_fn)(void *user, int slice, int n_slices, int64_t elem_begin, int64_t elem_end);
/* The ranges a sliced call will report, without making the call: elem_cuts[0] = 0 < ... < elem_cuts[*n_out] = all elements
 * (elem_cuts holds at least 17 entries; offsets = 0: bfg_paint_shell_sliced, 1: bfg_baryonify_offsets_sliced).  Lets the
 * caller decide BEFORE the call which pixels a rank will own after the slices' reduce-scatters -- and upload only those of the
 * input map (the distributed BaryonifyShell, Runners/HealpixRunner.py:357-370 on a rank's own pixel range).  Needs no GPU. */
int bfg_shell_slice_cuts(int64_t nside, int offsets, int n_slices, int64_t *elem_cuts, int *n_out);
int bfg_paint_shell_sliced(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_table *table,
                           const bfg_spline *da_spline, double *d_map, int n_slices, bfg_slice_fn fn, void *user);
int bfg_baryonify_offsets_sliced(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_table *table,
                                 const bfg_spline *da_spline, double *d_offsets, int n_slices, bfg_slice_fn fn, void *user);

/* Final regrid (Runners/HealpixRunner.py:357-365 + regrid_pixels_hpix :17-71):
 * every pixel with d_in_map != 0 is moved to pix2vec(p) + d_offsets[p] and
 * deposited on its 4 bilinear neighbours into d_out_map (accumulated INTO).
 * d_sums (device, 2 doubles, may be NULL) receives {sum(in_map), sum(deposits)}
 * for the mass-conservation assert of :368-370.                                       */
int bfg_regrid_shell(bfg_ctx *ctx, int64_t nside, const double *d_offsets, const double *d_in_map,
                     double *d_out_map, double *d_sums);

/* The same regrid for the SOURCE pixels of ring bands [band_lo, band_hi) only -- bands of bfg_regrid_band_rings() rings counted
 * from the north pole, i.e. contiguous RING pixel ranges.  Deposits reach at most 3 rings beyond a band's own rings, except
 * for displacements of more than 3 rings, which are added wherever they land and COUNTED: d_sums3 = {sum(in), sum(deposits),
 * number of threads with such a far deposit}, accumulated over the calls (flags bit 0: cleared first).  So the output of band
 * group k is final once groups k - 1, k and k + 1 have been regridded and the far count is zero: a map can be regridded while
 * it is still arriving over PCIe and leave while its other bands are regridded (Runners/HealpixRunner.py:357-365 band by
 * band).  NSIDE >= 8.                                                                                                    */
int bfg_regrid_band_rings(void);
int bfg_regrid_shell_bands(bfg_ctx *ctx, int64_t nside, const double *d_offsets, const double *d_in_map, double *d_out_map,
                           double *d_sums3, int band_lo, int band_hi, uint32_t flags);

/* Device -> page-locked host memory by a copy kernel on `stream` (a hipStream_t; NULL: the context's stream; the context's own
 * stream binding is left alone) (host_dst: hipHostMalloc'd / registered memory, 16-byte aligned; bytes a multiple of 8): it overlaps a host -> device DMA copy running on another stream, which a second DMA
 * copy in the opposite direction does not on every platform (the map of shell k leaves while the map of shell k + 1 arrives).  */
int bfg_copy_to_mapped_host(bfg_ctx *ctx, void *stream, void *host_dst, const void *d_src, size_t bytes);

/* ---- models that are NOT tabulated (any object with .projected / .displacement) ----------------------------------
 * The reference hands the distances of a halo's disc pixels to a Python callable, once per halo
 * (Runners/HealpixRunner.py:472 Baryons.projected(cosmo, r_sep / a_j, M_j, a_j, **o_j); :345 model.displacement(...)).
 * That call stays on the host; these four entry points are everything around it, for a batch of halos:
 *   bfg_disc_enumerate_count  d_counts[j] = pixels of halo j's disc (hp.query_disc, :463 / :330; with fallback4 != 0 a disc
 *                             of fewer than 4 pixels counts as the 4 bilinear neighbours of the centre, :333-334)
 *   bfg_disc_enumerate        with d_base = exclusive prefix sum of the counts: for entry e of halo j (base[j] <= e <
 *                             base[j] + counts[j]) d_pix[e] = RING pixel, d_r_com[e] = r_sep / a_j (:464-469, :472),
 *                             d_halo[e] = j
 *   bfg_map_add_values        d_map[d_pix[e]] += d_val[e]                                             (:481)
 *   bfg_offsets_add_displacements  d_offsets[d_pix[e]][0..2] += displaced unit vector - pixel unit vector for the comoving
 *                             displacement d_disp[e] the host computed for entry e                    (:345-355)
 * Uses of bfg_shell_args: nside, n_halo, d_catalog, cat_stride, epsilon_max, runner_md; the rest is ignored.          */
int bfg_disc_enumerate_count(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_spline *da_spline, int fallback4,
                             int64_t *d_counts);
int bfg_disc_enumerate(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_spline *da_spline, int fallback4,
                       const int64_t *d_base, int64_t *d_pix, double *d_r_com, int32_t *d_halo);
int bfg_map_add_values(bfg_ctx *ctx, double *d_map, const int64_t *d_pix, const double *d_val, int64_t n);
int bfg_offsets_add_displacements(bfg_ctx *ctx, const bfg_shell_args *args, const bfg_spline *da_spline,
                                  const int64_t *d_pix, const int32_t *d_halo, const double *d_disp, int64_t n,
                                  double *d_offsets);

/* ---- multi-GPU: the one exchange step ---------------------------------------------------
 * One process per GPU, one context per process.  Replaces the parent-side join of the reference's joblib wrapper,
 * map_out = np.sum(outputs, axis=0) (utils/Parallelize.py:312-318), by an RCCL collective over xGMI on the context's
 * stream: the per-rank maps of PaintProfilesShell (float64[Npix]) are all-reduced; BaryonifyShell -- which the
 * reference's splitter refuses (Parallelize.py:206-209) -- shards too because the offset field is linear in halos
 * (Runners/HealpixRunner.py:355): reduce-scatter the offsets float64[Npix][3], every rank regrids the pixel range it
 * owns, all-reduce the output map.
 * bfg_comm_unique_id   one rank creates the 128-byte id (an ncclUniqueId) and hands it to the others by whatever
 *                      channel the job has (a file, MPI, torch.distributed's store);
 * bfg_comm_init        collective over all ranks: builds the communicator of this context (rank in [0, world));
 * bfg_allreduce_f64    d_buf[count] <- sum over ranks, in place, asynchronous on the context's stream; a context
 *                      without a communicator is a world of one (no-op);
 * bfg_allreduce_f64_begin / bfg_comm_wait   the same sum on the context's own communication stream, ordered after what
 *                      the context's stream holds so far, so that it overlaps the work enqueued next (the next slice of
 *                      the same map, or the next shell painted into another buffer).  *ticket (may be NULL) receives
 *                      the collective's ticket, a positive number; bfg_comm_wait(ctx, ticket) makes the context's stream
 *                      wait for THAT collective (and, the communication stream being in order, those begun before it)
 *                      but not for later ones; ticket 0 = every collective begun so far.  Call it before the buffer is
 *                      read, zeroed or reused; the buffer must stay allocated until then.  bfg_reduce_scatter_f64_begin:
 *                      the same for the reduce-scatter half;
 * bfg_reduce_scatter_f64 / bfg_allgather_f64   the two halves, in place: rank r owns elements
 *                      [r count / world, (r + 1) count / world); count must be a multiple of world.
 * RCCL is loaded at the first of these calls (dlopen "librccl.so.1"; override with BFG_RCCL_SO): the library has no
 * link-time dependency on it.                                                                                   */
#define BFG_COMM_ID_BYTES 128
int bfg_comm_unique_id(char *id_out, size_t id_bytes);
int bfg_comm_init(bfg_ctx *ctx, const char *id, size_t id_bytes, int rank, int world);
int bfg_comm_destroy(bfg_ctx *ctx);
int bfg_comm_info(bfg_ctx *ctx, int *rank, int *world);
int bfg_allreduce_f64(bfg_ctx *ctx, double *d_buf, int64_t count);
int bfg_allreduce_f64_begin(bfg_ctx *ctx, double *d_buf, int64_t count, int64_t *ticket);
int bfg_reduce_scatter_f64_begin(bfg_ctx *ctx, double *d_buf, int64_t count, int64_t *ticket);
int bfg_comm_wait(bfg_ctx *ctx, int64_t ticket);
int bfg_reduce_scatter_f64(bfg_ctx *ctx, double *d_buf, int64_t count);
int bfg_allgather_f64(bfg_ctx *ctx, double *d_buf, int64_t count);

/* ---- tables with MORE p_keys axes than the shell kernels read (BFG_MAX_EXTRA) ---------------------------------------
 * ParamTabulatedProfile / BaryonificationClass tables are N-dimensional (utils/Tabulate.py:497-650,
 * Profiles/BaryonCorrection.py:211-227, :404-408).  All non-radial coordinates of a (halo, pixel) query are the halo's, so the
 * multilinear read-out factors: bfg_ndtable_rows blends the 2^n_outer corners ONCE per halo into the halo's radial row, and
 * bfg_ndtable_read interpolates that row per (halo, pixel) entry of a disc enumeration (bfg_disc_enumerate); the values go
 * back through bfg_map_add_values / bfg_offsets_add_displacements.  Nothing is evaluated on the host.
 *   bfg_ndtable_create   n_outer = 2 + number of p_keys axes (<= 12): axes (ln(1+z), ln M, p_1 ...), strictly ascending; the
 *                        radial axis ln r [or ln r/R_delta]; values[z][M][p_1]...[p_n][r] (radial index fastest): ln T for
 *                        paint, d for displacement tables;
 *   bfg_ndtable_rows     d_rows[j][0 .. nr) for the n_halo records (M, z, ra, dec, p_1 ...) of d_catalog; NaN rows for halos
 *                        outside the hull of any axis (RegularGridInterpolator fill_value = nan);
 *   bfg_ndtable_read     d_out[e] for entries (d_halo[e], d_r_com[e]): the row at ln(r_com) - d_shift[halo] (d_shift NULL: 0;
 *                        Rdelta_sampling: ln R_delta,com), NaN outside the radial axis; exp_values != 0 (paint,
 *                        HealpixRunner.py:472-478): exp of it, non-finite -> 0, times d_scale[halo] (NULL: 1; pixarea D^2);
 *                        exp_values == 0 (displacement, BaryonCorrection.py:410-411): 0 where r_com >= d_rcut[halo] (NULL: no
 *                        cut).  d_r_oob (device, may be NULL): incremented by the entries outside the radial axis.          */
typedef struct bfg_ndtable bfg_ndtable;
int bfg_ndtable_create(bfg_ctx *ctx, int n_outer, const int64_t *outer_shape, const double *const *outer_axes, int64_t nr,
                       const double *raxis, const double *values, bfg_ndtable **out);
int bfg_ndtable_destroy(bfg_ctx *ctx, bfg_ndtable *table);
int bfg_ndtable_rows(bfg_ctx *ctx, const bfg_ndtable *table, const double *d_catalog, int64_t n_halo, int cat_stride,
                     double *d_rows);
int bfg_ndtable_read(bfg_ctx *ctx, const bfg_ndtable *table, const double *d_rows, int64_t n, const int32_t *d_halo,
                     const double *d_r_com, const double *d_shift, const double *d_rcut, const double *d_scale, int exp_values,
                     double *d_out, unsigned int *d_r_oob);

/* max |x| over a device array (np.allclose(orig_map, 0) early return, :293-294) and sum. */
int bfg_reduce_absmax_sum(bfg_ctx *ctx, int64_t n, const double *d_x, double *absmax, double *sum);

int bfg_plan_reuses(bfg_ctx *ctx, int64_t *count);   /* shell calls of this context that ran on a reused plan (BFG_SHELL_REUSE_PLAN) */
int bfg_stats_reset(bfg_ctx *ctx);
int bfg_stats_read(bfg_ctx *ctx, bfg_stats *out);   /* synchronises the stream */

/* Optional per-kernel timing with hipEvents on the context's stream (bench.py's
 * roofline leg).  which: 0 = halo preparation kernel, 1 = the dominant shell kernel
 * (tile kernel, or the scatter kernel of the scatter variants), 2 = regrid kernel,
 * 3 = tile binning (work list + overflow fill + row windows), 4 = left-over scatter kernel of the tile
 * variant, 5 = the follow-up kernel that adds the tile kernel's deferred pixels (paint; launched only with BFG_FINAL_DRAIN=kernel,
 * by default the tile workgroups add them themselves, inside class 1), 6 = snap_particle_kernel of bfg_baryonify_snapshot*
 * (the per-particle displacement pass), 7 = the three kernels of the tile-privatised deposit of bfg_deposit_grid*.
 * Returns the accumulated milliseconds and launch count since the last enable.
 * bfg_timing_select restricts the events to the classes in `which_mask` (bit k = class k; the default, and what
 * bfg_timing_enable restores, is all of them): every event pair costs a few microseconds of stream time, which a step of
 * seven small launches feels (0.04 ms per shell call with all classes timed). */
int bfg_timing_enable(bfg_ctx *ctx, int enable);
int bfg_timing_select(bfg_ctx *ctx, unsigned which_mask);
int bfg_timing_read(bfg_ctx *ctx, int which, double *ms_total, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* BFG_MI355_H */
