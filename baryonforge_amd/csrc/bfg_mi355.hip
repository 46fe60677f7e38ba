// bfg_mi355.hip -- libbfg_mi355.so: hand-written gfx950 (CDNA4, wave64) kernels
// and the C-ABI of include/bfg_mi355.h for BaryonForge's shell paint / baryonify
// hot path (Runners/HealpixRunner.py:252-483 of the reference).
//
// Kernels (all float64, HBM/atomic/ALU bound -- there is no dense contraction
// on this path, so no MFMA):
//   halo_prep_kernel      catalog record -> per-halo scalars (a, R_delta, D_A,
//                         unit vector, disc ring range, (z, M, extras) table cell)
//   shell_scatter_kernel  one G-lane group per halo: ring windows of the disc
//                         (query_disc arithmetic) -> prefix scan over rings ->
//                         flattened pixel loop -> LDS-staged blended profile row
//                         -> global f64 atomic scatter-add (paint: 1, baryonify: 3)
//   regrid_kernel         one thread per pixel: displaced direction -> 4 bilinear
//                         neighbours -> 4 f64 atomics
//   reduce kernels        sum / absmax
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bfg_mi355.h"
#include "bfg_device.hpp"

using namespace bfg;

// ------------------------------------------------------------------------------------
// host-side bookkeeping
// ------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return BFG_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

enum { F_X0 = 0, F_Y0, F_Z0, F_PTHETA, F_PPHI, F_D, F_A, F_RM, F_RADIUS, F_NF };   // only what a later kernel reads
enum { I_RFIRST = 0, I_RLAST, I_IRMIN, I_IRMAX, I_FLAGS, I_NI };
#define HF_OOB 1        // (z, M, extras) outside the table hull, or NaN
#define HF_SKIP 2       // nothing to do for this halo (NaN radius etc.)
#define HF_SCATTER 4    // tile variant: this halo is left to the global-atomic scatter kernel
#define HF_SLOW 8       // ... although it has work to do (too many tiles, exp() range, pair buffer full): ~12x slower, counted

struct DevTable {
    int ndim;                       // 3 + n_extra
    int nouter;                     // ndim - 1 : (z, M, extras...)
    int nr;                         // r-axis length
    int oshape[BFG_MAX_DIM];        // outer axis lengths
    int64_t ostride[BFG_MAX_DIM];   // outer strides in doubles (r is fastest)
    const double *oaxis[BFG_MAX_DIM];
    const double *raxis;
    const double *values;           // permuted to [z][M][extras...][r]
    int log_values;
    int r_uniform;                  // ln r axis is a linspace to rounding
    int hot;                        // a log table with finite |ln T| > 650: exp() range handling stays with the scatter kernels
    int hstride;                    // 0: one table for every halo.  > 0: `values` holds ONE radial row PER HALO, hstride doubles apart
                                    // (nouter = 0: the corner blend over a table with more p_keys axes than these kernels read was done
                                    // once per halo by nd_rows_kernel, bfg_ndtable.hpp); every corner offset starts at j * hstride.
                                    // (Sits in what was padding: the kernels' argument layout is the round-4 one.)
    double r0, inv_dr;
};
static_assert(sizeof(DevTable) == 184, "DevTable grew: the tile kernels' argument layout (and with it their register allocation) changes");

struct bfg_ndtable;
struct bfg_table {
    DevTable dev;
    double *d_blob;                 // one allocation: axes + values
    std::vector<int64_t> shape;
    bfg_ndtable *nd = nullptr;      // more than BFG_MAX_DIM dimensions: the table itself (dev then describes the halos' rows: see hstride)
    double ax_inv_h[2] = {0.0, 0.0};   // inverse mean spacing of the z and M axes (find_interval_hint; host side only: DevTable stays 184 B)
    uint64_t axes_hash = 0;            // FNV-1a over the shape and every axis value: two tables on the same grid share it (plan reuse)
};

// (bfg_ndtable is defined with its C-ABI functions at the end of this file)
static const double *ndtable_raxis(const bfg_ndtable *t);
static bool nd_rows_pay(const bfg_table *t, const bfg_shell_args *a);
static int run_shell_nd(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s, double *d_out, int mode,
                        int n_slices, bfg_slice_fn slice_fn, void *slice_user);

struct bfg_spline {
    int n;
    double inv_h;     // (n - 1) / (knots[n-1] - knots[0]): find_interval_hint
    double *d_knots;  // [n]
    double *d_coef;   // [4][n-1]
};

namespace bfg { struct HaloTile; struct HaloDisp; struct DeferredOut; }
constexpr int kTimingSlots = 8;     // bfg_timing_read: prep, dominant shell kernel, regrid, binning, left-overs, deferred pixels,
                                    // snapshot particle kernel, tiled deposit kernel
struct bfg_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    int n_cu;
    int lds_per_cu;
    size_t max_dyn_lds;
    // per-halo workspace
    int64_t cap_halo;
    double *d_rec;      // [F_NF][cap]
    int32_t *d_irec;    // [I_NI][cap]
    int32_t *d_cidx;    // [BFG_MAX_DIM-1][cap]
    double *d_cw;       // [BFG_MAX_DIM-1][cap]
    bfg_stats *d_stats;
    double *d_red;      // scratch for reductions [4]
    void *d_rg_slow;    // regrid: pixels the tile kernel's differential path leaves to regrid_list_kernel (grow-only, one entry per
                        // pixel: int32 while the map has fewer than 2^31 pixels, int64 beyond)
    int64_t rg_slow_cap;
    unsigned long long *d_rg_slow_n;
    // tile variant: geometry of the current nside, binning buffers, ln / exp tables
    struct TileSet {                // tile geometry + binning buffers of one (nside, rings-per-tile)
        int64_t nside;
        TileGeom geo;
        int32_t *d_geo;             // band_ns | band_tile0 | band_nrmin | tile_band
        int32_t *d_tile_count, *d_tile_start;   // d_tile_count: two sets of ntiles + kTileTail counters; a call uses set `flip` and its scan
                                                // kernel clears the other one for the next call (no memset launch per call)
        int flip;
        bool counting;                          // a call counted into set `flip` and its scan kernel was never launched (an error in between)
        int4 *d_work;               // [2 ntiles + kWorkExtra] work items of the tile kernel
        bfg::DeferredOut *d_defer;  // [2 ntiles + kWorkExtra][kDeferCap] pixels left to tile_deferred_kernel (paint)
        int32_t *d_defer_count;     // [2 ntiles + kWorkExtra]
        int cap_direct;             // fixed pair slots per tile of the current call
        int32_t *d_nwork;
        int32_t *d_counters;        // [kCounterInts] item counters of the persistent tile kernel (bfg_tile.hpp)
        int32_t *d_shared;          // [ntiles] 1: the tile's pair list was cut into several work items (atomics on the map)
        int32_t *d_slices;          // [2 kMaxSlices] sliced calls: item range per slice
    } tiles[3];                     // [MODE_PAINT], [MODE_BARYONIFY], [2] = the regrid kernel's tiles
    int32_t *d_pairs;              // [ntiles * cap_direct] slots | [pair_cap] overflow lists
    unsigned long long *d_ovf_mask; // [cap_halo]
    int64_t pairs_alloc;           // entries allocated in d_pairs
    bfg::HaloDisp *d_hd;            // [cap_halo] baryonify tile path
    int32_t *d_left;                // [cap_halo + 1] tile variant: [0] = count, then the halos left to the scatter kernel
    // bfg_baryonify_snapshot workspace (grow-only)
    void *snap_buf[8];
    size_t snap_cap[8];
    void *grid_buf[6];              // grid runners: per-halo records, blended rows, tile counts / starts / pairs / scan scratch
    size_t grid_cap[6];
    bool grid_attr_set;             // MaxDynamicSharedMemorySize raised for the grid tile kernels
    void *dep_buf[5];               // tiled deposit: keys, permutation, tile counts, tile starts, scan scratch (grow-only)
    size_t dep_cap[5];
    int64_t pair_cap;
    unsigned long long *d_pair_total;
    double *d_mathtab;              // logtab (256 doubles) | exptab (64 doubles) | atantab (72 doubles)
    bfg::HaloTile *d_ht;            // [cap_halo]
    double *d_hwin;                 // [hwin_cap] pre-blended row windows
    int64_t hwin_cap;
    double *d_ndrows = nullptr;     // [ndrows_cap] the halos' radial rows of an N-dimensional table (run_shell_nd)
    int64_t ndrows_cap = 0;
    void *d_ndsort = nullptr;       // run_shell_nd, halos grouped by table cell: cell ids, sorted indices, weights, counters (bytes)
    size_t ndsort_cap = 0;
    bool tile_attr_set;             // MaxDynamicSharedMemorySize raised for the tile kernels on this device
    // timing: a growing pool of event pairs per kernel class, resolved lazily in bfg_timing_read
    bool timing;
    unsigned timing_mask;           // classes that get events (bfg_timing_select)
    std::vector<hipEvent_t> *ev_a[kTimingSlots], *ev_b[kTimingSlots];
    size_t ev_used[kTimingSlots];
    double t_ms[kTimingSlots];
    int64_t t_n[kTimingSlots];
    // The records and pair lists the last tile-path shell call left behind, and what they were built from (BFG_SHELL_REUSE_PLAN)
    struct ShellPlanKey {
        const double *cat; const void *spline; int64_t n_halo, nside; uint64_t axes_hash;
        double eps, eps_model, pixfac_area; bfg_massdef md_run, md_model;
        int cat_stride, n_extra, rdelta, mode, win_nodes, win_table, blend, blend_rows, overwrite, out_zero, slice_K, hstride, n_knots;
        int cap_direct, direct_limit; long long pair_cap;      // (test hooks BFG_TILE_CAP / BFG_PAIR_CAP / BFG_DIRECT_LIMIT change them)
    };
    struct ShellPlan {
        bool valid;
        ShellPlanKey key;
        int32_t *tail;              // the planning call's counter set behind its tile counts (left_n, needs_scan, plan statistics)
    } plan;
    unsigned long long plan_reuses; // calls that ran on a reused plan (bfg_plan_reuses: tests, bench)
    hipEvent_t ev_switch;           // orders the context's work across a change of stream (bfg_ctx_set_stream)
    struct bfg_comm_state *comm;    // RCCL communicator of bfg_comm_init (multi-GPU), or null
};
static void bfg_comm_release(bfg_ctx *c);

struct ShellParams {
    Hpx hpx;
    int64_t n_halo;
    int64_t cap;             // stride of the SoA workspace
    const double *rec;
    const int32_t *irec;
    const int32_t *cidx;
    const double *cw;
    const double *cat;       // catalog (for ra/dec of the fallback)
    int cat_stride;
    DevTable tab;
    int win_nodes;           // LDS window length in r nodes
    double eps_model;
    int rdelta;
    double pixfac_area;      // pixarea if include_pixel_size else 0
    double *out;             // map [npix] or offsets [npix][3]
    bfg_stats *stats;
    int only_flagged;        // process only halos the tile binning flagged HF_SCATTER
    const int32_t *left;     // only_flagged: left[1..] = the flagged halos (else nullptr)
    const int32_t *left_n;   // ... and their number
    const int32_t *pair_total_ptr;   // tile_start[ntiles] and the pair buffer capacity (overflow -> every halo is flagged)
    long long pair_cap;
};

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
struct PrepParams {
    Hpx hpx;
    int64_t n_halo, cap;
    const double *cat;
    int cat_stride, n_extra;
    double eps_run;
    bfg_massdef md_run, md_model;
    int spl_n;
    const double *spl_knots, *spl_coef;
    DevTable tab;
    double *rec;
    int32_t *irec;
    int32_t *cidx;
    double *cw;
    bfg_stats *stats;
    int want_model_radius;
    bfg::HaloTile *ht;       // tile variant: per-halo 128-byte records (may be null)
    int win_nodes;
    double pixfac_area;
    bfg::BinCtx bin;         // tile variant: count pass of the halo -> tile binning
    bfg::HaloDisp *hd;       // baryonify tile path
    int32_t *left;           // tile variant: left[1..] = halos flagged for the scatter kernel
    int32_t *left_n;         // ... their number: the int32 after the tile counters, cleared by the same memset
    double eps_model;
    int rdelta;
    double *hwin;            // tile variant, row windows of 4 k nodes: the kernel also builds the halos' blended rows (null: a
                             // separate halo_row*_kernel does)
    int lazy_soa;            // tile path, 3-D table: the SoA workspace rows only of halos the scatter kernel / the fill pass will read
    // inverse mean spacing of the D_A knots and of the table's z and M axes (find_interval_hint: the cell from a guess + the same
    // comparisons instead of a bisection; 0 = unknown, any value gives the same cell)
    double spl_inv_h, ax_inv_h[2];
};

#define MODE_PAINT 0
#define MODE_BARYONIFY 1
constexpr int kMaxCorner = 1 << (BFG_MAX_DIM - 1);
// Everything halo_prep_kernel derives from one catalog record.  A pure function of the record (no atomics, no stores), so
// that tile_fill_kernel can redo it for the rare call that degrades to the scatter kernel after the fact (see `lazy_soa`).
struct HaloCalc {
    double x0, y0, z0v, ptheta, pphi, D, a, Rm, radius;     // what the scatter kernel reads (F_* columns of the SoA workspace)
    double st, sp, cp, lnpf, pixfac, Rm_com;
    int32_t flags, rfirst, rlast, irmin, irmax;
    uint32_t warn;
    bool oob;
};

// axis(k): the k-th non-radial table axis (LDS or global); cell(k, i, y): receives the halo's cell index and weight on it
// (functors, not arrays: private arrays indexed by k end up in scratch memory)
template <class AxisFn, class CellFn>
__device__ inline void halo_calc(const PrepParams &P, int64_t j, const double *__restrict__ knots, AxisFn axis, CellFn cell,
                                 HaloCalc &o);
__device__ inline void halo_write_soa(const PrepParams &P, int64_t j, const HaloCalc &o, int32_t flags);
#include "bfg_tile.hpp"

__device__ inline double massdef_radius(const bfg_massdef &md, double M, double a)
{
    double rho;
    if (md.rho_type == 0) {
        const double de = (md.w0 == -1.0) ? 1.0 : pow(a, -3.0 * (1.0 + md.w0));    // pow(a, -0.0) == 1: skip it for Lambda
        double E2 = md.Omega_m / (a * a * a) + md.Omega_l * de +
                    md.Omega_r / (a * a * a * a);
        rho = md.rho_crit0_h2 * md.h * md.h * E2;
    } else {
        rho = md.rho_crit0_h2 * md.h * md.h * md.Omega_m / (a * a * a);
    }
    return cbrt(M / (4.18879020479 * md.Delta * rho));
}

#include "bfg_snapshot.hpp"
#include "bfg_grid.hpp"
#include "bfg_tablebuild.hpp"

// scipy PPoly evaluation of the not-a-knot CubicSpline of HealpixRunner.py:299 (extrapolates)
__device__ inline double spline_eval(int n, const double *__restrict__ x, const double *__restrict__ c, double v, double inv_h = 0.0)
{
    int i = find_interval_hint(x, n, v, inv_h);
    double s = v - x[i];
    int m = n - 1;
    return ((c[i] * s + c[m + i]) * s + c[2 * m + i]) * s + c[3 * m + i];
}

constexpr int kPrepKnots = 1024;     // D_A spline knots staged in LDS (the reference uses 1000, HealpixRunner.py:297)
constexpr int kPrepAxis = 64;        // nodes of a non-radial table axis staged in LDS
constexpr int kPrepRadial = 512;     // nodes of the radial axis staged in LDS (the window search is one more bisection)
// dynamic LDS of halo_prep_kernel ahead of the row phase's corner arrays: cell weights [5][256] f64, ln(pixfac) [256] f64, cell
// indices [5][256] i32, window starts and flags [2][256] i32 -- present only when the kernel builds row windows or the table has
// extra axes (see the kernel)
constexpr size_t kPrepRowLds = (size_t)(BFG_MAX_DIM - 1) * 256 * 8 + 256 * 8 + (size_t)(BFG_MAX_DIM - 1) * 256 * 4 + 2 * 256 * 4;

// knots / axes: the D_A spline knots and the non-radial table axes, in LDS (the prep kernel stages them) or in global memory
template <class AxisFn, class CellFn>
__device__ inline void halo_calc(const PrepParams &P, int64_t j, const double *__restrict__ knots, AxisFn axis, CellFn cell,
                                 HaloCalc &o)
{
    const double *c = P.cat + j * (int64_t)P.cat_stride;
    const double M = c[0], zred = c[1], ra = c[2], dec = c[3];
    const double a = 1.0 / (1.0 + zred);                                   // HealpixRunner.py:319/:453
    const double R = massdef_radius(P.md_run, M, a);                        // :320/:454
    const double D = spline_eval(P.spl_n, knots, P.spl_coef, zred, P.spl_inv_h);   // :321/:455
    // hp.ang2vec(ra, dec, lonlat=True)                                      // :327/:460
    const double theta = kHalfPi - dec * kDeg2Rad, phi = ra * kDeg2Rad;
    // (trigonometry without libm where the argument is in range -- sincos_range / atan2_upper: <= 2 ulp, a third of ocml's
    // instructions; this kernel runs at ~45 % VALU utilisation on chains of dependent f64 operations, so instructions are time:
    // profiles/r04_prep_ab.txt.  Out-of-range arguments -- |dec| > 90, ra outside [0, 360] -- take libm as before.)
    double st, ct, sp, cp;
    sincos_range(theta, st, ct);
    sincos_range(phi, sp, cp);
    const double x0 = st * cp, y0 = st * sp, z0v = ct;
    // pointing(vec3) inside query_disc
    const double rxy = sqrt(x0 * x0 + y0 * y0);
    double ptheta = (rxy == rxy && z0v == z0v) ? atan2_upper(rxy, z0v) : atan2(rxy, z0v);
    double pphi = 0.0;
    if (!(x0 == 0.0 && y0 == 0.0)) {
        if (x0 == x0 && y0 == y0) { const double ap = atan2_upper(fabs(y0), x0); pphi = (y0 < 0.0) ? -ap : ap; }
        else pphi = atan2(y0, x0);
    }
    if (pphi < 0.0) pphi += kTwoPi;
    const double radius = R * P.eps_run / D;                                 // :329/:462
    o.x0 = x0; o.y0 = y0; o.z0v = z0v; o.ptheta = ptheta; o.pphi = pphi; o.D = D; o.a = a; o.radius = radius;
    o.st = st; o.sp = sp; o.cp = cp;
    o.Rm = P.want_model_radius ? massdef_radius(P.md_model, M, a) / a : 0.0;   // BaryonCorrection.py:399
    o.Rm_com = P.want_model_radius ? o.Rm : 1.0;
    const double lnM = log(M), lnz = log(1.0 / a);                           // Tabulate.py:308,312

    // disc ring range (query_disc_internal, fact = 0)
    int32_t flags = 0, rfirst = 1, rlast = 0, irmin = 1, irmax = 0;
    const int64_t nl4 = 4 * P.hpx.nside;
    if (!(radius >= 0.0) || !isfinite(ptheta)) {
        flags |= HF_SKIP;   // NaN radius: every healpy comparison is false -> empty disc
    } else if (radius >= kPi) {
        rfirst = 1; rlast = (int32_t)(nl4 - 1); irmin = (int32_t)nl4; irmax = (int32_t)nl4;  // every ring complete
    } else {
        const double rlat1 = ptheta - radius;
        const double zmax = cos_range(rlat1);
        int64_t imin = ring_above(P.hpx, zmax) + 1;
        const double rlat2 = ptheta + radius;
        const double zmin = cos_range(rlat2);
        int64_t imax = ring_above(P.hpx, zmin);
        bool north = (rlat1 <= 0) && (imin > 1);
        bool south = (rlat2 >= kPi) && (imax + 1 < nl4);
        irmin = (int32_t)imin; irmax = (int32_t)imax;
        rfirst = north ? 1 : irmin;
        rlast = south ? (int32_t)(nl4 - 1) : irmax;
    }
    // table cell of the halo in the outer (non-radial) dimensions
    bool oob = false;
    uint32_t warn = 0;
#pragma unroll
    for (int k = 0; k < BFG_MAX_DIM - 1; ++k) {          // (unrolled: k is a constant in every copy, so the callers' cell functors can
        if (k >= P.tab.nouter) break;                    // keep the first two cells in registers)
        double x = (k == 0) ? lnz : (k == 1) ? lnM : c[4 + (k - 2)];
        int n = P.tab.oshape[k];
        const double *g = axis(k);
        if (!(x >= g[0]) || !(x <= g[n - 1])) {
            oob = true;
            if (k == 0) warn |= BFG_WARN_Z_RANGE;
            if (k == 1) warn |= BFG_WARN_M_RANGE;
        }
        int i = (k < 2) ? find_interval_hint(g, n, x, P.ax_inv_h[k]) : find_interval(g, n, x);
        cell(k, i, (x - g[i]) / (g[i + 1] - g[i]));
    }
    if (oob) flags |= HF_OOB;
    // paint: the tile path folds ln(pixarea D^2) into the halo's row window; keep exp() range handling exact
    o.pixfac = (P.pixfac_area != 0.0) ? P.pixfac_area * (D * D) : 1.0;
    o.lnpf = (P.pixfac_area != 0.0) ? log(o.pixfac) : 0.0;
    if (P.ht && P.bin.mode == MODE_PAINT && !(fabs(o.lnpf) < 50.0)) flags |= HF_SCATTER | HF_SLOW;
    o.flags = flags; o.rfirst = rfirst; o.rlast = rlast; o.irmin = irmin; o.irmax = irmax;
    o.warn = warn; o.oob = oob;
}

// the SoA workspace row of one halo (what shell_scatter_kernel and the fill pass of the binning read)
__device__ inline void halo_write_soa(const PrepParams &P, int64_t j, const HaloCalc &o, int32_t flags)
{
    const int64_t cap = P.cap;
    double *rec = P.rec + j;
    rec[F_X0 * cap] = o.x0; rec[F_Y0 * cap] = o.y0; rec[F_Z0 * cap] = o.z0v;
    rec[F_PTHETA * cap] = o.ptheta; rec[F_PPHI * cap] = o.pphi;
    rec[F_D * cap] = o.D; rec[F_A * cap] = o.a;
    rec[F_RM * cap] = o.Rm;
    rec[F_RADIUS * cap] = o.radius;
    int32_t *irec = P.irec + j;
    irec[I_RFIRST * cap] = o.rfirst; irec[I_RLAST * cap] = o.rlast;
    irec[I_IRMIN * cap] = o.irmin; irec[I_IRMAX * cap] = o.irmax;
    irec[I_FLAGS * cap] = flags;
}

#include "bfg_enum.hpp"
#include "bfg_ndtable.hpp"

#ifndef BFG_PREP_WAVES
#define BFG_PREP_WAVES 1
#endif
__global__ __launch_bounds__(256, BFG_PREP_WAVES) void halo_prep_kernel(const PrepParams P)
{
    // the bisections below (spline knots: 10 steps; table axes: 4-5 steps each) are chains of dependent loads: from L2 they
    // cost ~10 us per wavefront, from LDS well under 1 us.  Same comparisons on the same values, so the same cells.
    __shared__ double s_knots[kPrepKnots];
    __shared__ double s_axis[BFG_MAX_DIM - 1][kPrepAxis];
    __shared__ double s_raxis[kPrepRadial];
    // Dynamic LDS, present only when the kernel builds row windows (P.hwin) or the table has extra axes (kPrepRowLds bytes + the
    // row phase's corner arrays): per halo of the block the outer cell (index, weight per axis), window start, flags, ln(pixfac).
    // The usual call -- 3-D table, windows blended in the tile kernel -- keeps the two cells in registers and the block at 15 KB of
    // LDS instead of 34 KB.
    extern __shared__ double smem_prep[];
    const bool cells_lds = P.hwin != nullptr || P.tab.nouter > 2;
    double (*s_cy)[256] = reinterpret_cast<double (*)[256]>(smem_prep);                                   // [5][256]
    double *s_ln = smem_prep + (BFG_MAX_DIM - 1) * 256;                                                   // [256]
    int32_t (*s_ci)[256] = reinterpret_cast<int32_t (*)[256]>(smem_prep + (BFG_MAX_DIM) * 256);          // [5][256]
    int32_t *s_wl = reinterpret_cast<int32_t *>(smem_prep + (BFG_MAX_DIM) * 256) + (BFG_MAX_DIM - 1) * 256, *s_fl = s_wl + 256;
    double *smem_rows = smem_prep + kPrepRowLds / 8;     // row phase: corner weights / row offsets of the halos of one pass
    int32_t ci0 = 0, ci1 = 0;                            // the halo's cell on the z and M axes when it stays in registers
    double cy0 = 0.0, cy1 = 0.0;
    const bool knots_lds = P.spl_n <= kPrepKnots;
    const bool raxis_lds = P.ht && P.tab.nr <= kPrepRadial;
    if (raxis_lds) for (int i = threadIdx.x; i < P.tab.nr; i += blockDim.x) s_raxis[i] = P.tab.raxis[i];
    if (knots_lds) for (int i = threadIdx.x; i < P.spl_n; i += blockDim.x) s_knots[i] = P.spl_knots[i];
    for (int k = 0; k < P.tab.nouter; ++k)
        if (P.tab.oshape[k] <= kPrepAxis)
            for (int i = threadIdx.x; i < P.tab.oshape[k]; i += blockDim.x) s_axis[k][i] = P.tab.oaxis[k][i];
    __syncthreads();
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cells_lds) { s_fl[threadIdx.x] = HF_SKIP; s_wl[threadIdx.x] = 0; s_ln[threadIdx.x] = 0.0; }
    if (j < P.n_halo) {
    HaloCalc hc;
    halo_calc(P, j, knots_lds ? s_knots : P.spl_knots,
              [&](int k) -> const double * { return (P.tab.oshape[k] <= kPrepAxis) ? s_axis[k] : P.tab.oaxis[k]; },
              [&](int k, int i, double y) {
                  if (k == 0) { ci0 = i; cy0 = y; }
                  if (k == 1) { ci1 = i; cy1 = y; }
                  if (cells_lds) { s_ci[k][threadIdx.x] = i; s_cy[k][threadIdx.x] = y; }
              }, hc);
    if (hc.oob) {
        atomicAdd((unsigned long long *)&P.stats->halos_out_of_table, 1ull);
        atomicOr(&P.stats->warn_mask, hc.warn);
        if (P.left_n) { atomicAdd(&P.left_n[kPlanOob], 1); atomicOr(&P.left_n[kPlanWarn], (int32_t)hc.warn); }   // (for a call that reuses this plan)
    }
    int32_t flags = hc.flags;
    const int32_t rfirst = hc.rfirst, rlast = hc.rlast, irmin = hc.irmin, irmax = hc.irmax;
    const double radius = hc.radius, ptheta = hc.ptheta, pphi = hc.pphi;
    // Everything that is stored per halo goes out BEFORE the binning (with the flags as they stand; the binning may add
    // HF_SCATTER / HF_SLOW, patched in afterwards with one 4-byte store): what stays live across the binning's atomics is the disc's
    // ring range and pointing, not the twenty doubles of the record.
    if (P.ht) {
        const double D = hc.D, a = hc.a;
        HaloTile h;
        h.st = hc.st; h.ct = hc.z0v; h.pphi = pphi;
        h.S = (D / a) * (D / a);
        h.cosr = cos_range(radius);
        h.z0 = cos_range(ptheta);
        h.xa = 1.0 / sqrt((1.0 - h.z0) * (1.0 + h.z0));
        h.pixfac = hc.pixfac;
        h.rfirst = rfirst; h.rlast = rlast; h.irmin = irmin; h.irmax = irmax;
        // staged row window: ends at the node above the largest radius of the disc (on the table's radial axis)
        const double sr = sin_range(0.5 * fmin(radius, kPi));
        const double axis_shift = (P.hd && P.rdelta) ? log(hc.Rm_com) : 0.0;
        const double rho_max = 0.5 * log(4.0 * h.S * sr * sr) - axis_shift;
        int win_lo = find_interval_hint(raxis_lds ? s_raxis : P.tab.raxis, P.tab.nr, rho_max, P.spl_inv_h != 0.0 ? P.tab.inv_dr : 0.0) + 1 - (P.win_nodes - 1);
        if (win_lo > P.tab.nr - P.win_nodes) win_lo = P.tab.nr - P.win_nodes;
        if (win_lo < 0) win_lo = 0;
        if (P.hd) {
            HaloDisp hd;
            hd.cp0 = hc.cp; hd.sp0 = hc.sp; hd.a = a; hd.D = D;
            hd.xcut = (P.eps_model * hc.Rm_com) * (P.eps_model * hc.Rm_com);
            hd.tshift = -axis_shift * P.tab.inv_dr;
            hd.pad[0] = hd.pad[1] = 0.0;
            P.hd[j] = hd;
        }
        h.win_lo = win_lo; h.flags = flags;
        h.ci0 = ci0; h.ci1 = (P.tab.nouter > 1) ? ci1 : 0;
        h.spare[0] = hc.lnpf; h.spare[1] = cy0; h.spare[2] = (P.tab.nouter > 1) ? cy1 : 0.0;
        h.spare[3] = 0.0;
        P.ht[j] = h;
        if (cells_lds) { s_wl[threadIdx.x] = win_lo; s_fl[threadIdx.x] = flags; s_ln[threadIdx.x] = hc.lnpf; }
    }
    // The SoA workspace is what the scatter kernel and the fill pass of the binning read.  On the tile path with a 3-D table
    // (lazy_soa) only the halos they will touch get it -- the ones left to the scatter kernel and the ones with pairs in overflow
    // lists: none at all for a catalog spread over the sky, 124 B per halo less to write --, and those recompute their record after
    // the binning (halo_calc is pure: the rare halo pays twice, no halo keeps its record in registers across the atomics).
    // (Should the call degrade to the scatter kernel after the fact -- pair buffer exhausted -- tile_fill_kernel recomputes the rows.)
    auto write_soa = [&](const HaloCalc &o, int32_t fl) {
        halo_write_soa(P, j, o, fl);
        if (cells_lds) for (int k = 0; k < P.tab.nouter; ++k) { P.cidx[k * P.cap + j] = s_ci[k][threadIdx.x]; P.cw[k * P.cap + j] = s_cy[k][threadIdx.x]; }
        else {
            if (P.tab.nouter > 0) { P.cidx[j] = ci0; P.cw[j] = cy0; }
            if (P.tab.nouter > 1) { P.cidx[P.cap + j] = ci1; P.cw[P.cap + j] = cy1; }
        }
    };
    if (!P.lazy_soa) write_soa(hc, flags);
    unsigned long long ovf = 0ull;
    const int32_t flags0 = flags;
    if (P.ht) {
        if (P.bin.mode == MODE_BARYONIFY && !(flags & HF_SKIP) && rlast >= rfirst && rlast - rfirst < 8) {
            // small disc: count its pixels exactly; fewer than 4 -> 4-neighbour fallback (HealpixRunner.py:333-334),
            // which only the scatter kernel implements
            const double cosr = cos(radius), zc = cos(ptheta);
            const double xa_ = 1.0 / sqrt((1.0 - zc) * (1.0 + zc));
            int total = 0;
            for (int ring = rfirst; ring <= rlast; ++ring) {
                int64_t sp_, nr_; bool sh_;
                ring_info_small(P.hpx, ring, sp_, nr_, sh_);
                if (ring < irmin || ring > irmax) { total += (int)nr_; continue; }
                const double zr = ring2z(P.hpx, ring);
                const double xx = (cosr - zr * zc) * xa_;
                const double ysq = 1.0 - zr * zr - xx * xx;
                const double dphi = (ysq <= 0.0) ? 0.0 : atan2(sqrt(ysq), xx);
                if (dphi > 0.0) {
                    const double shift = sh_ ? 0.5 : 0.0;
                    const int64_t l64 = (int64_t)floor((double)nr_ * kInvTwoPi * (pphi - dphi) - shift) + 1;
                    const int64_t h64 = (int64_t)floor((double)nr_ * kInvTwoPi * (pphi + dphi) - shift);
                    int64_t cc = h64 - l64 + 1;
                    if (cc > nr_) cc = nr_;
                    if (cc > 0) total += (int)cc;
                }
            }
            if (total < 4) flags |= HF_SCATTER;
        }
        if (!(flags & HF_SCATTER))
            flags = tile_bin_halo(P.bin, false, j, flags, rfirst, rlast, irmin, irmax, ptheta, pphi, radius, ovf);
        P.bin.ovf_mask[j] = ovf;
        if (flags != flags0) {
            P.ht[j].flags = flags;
            if (cells_lds) s_fl[threadIdx.x] = flags;
            if (!P.lazy_soa) P.irec[I_FLAGS * P.cap + j] = flags;
        }
    }
    if (P.lazy_soa && ((flags & HF_SCATTER) || ovf != 0ull)) {
        HaloCalc again;
        halo_calc(P, j, knots_lds ? s_knots : P.spl_knots,
                  [&](int k) -> const double * { return (P.tab.oshape[k] <= kPrepAxis) ? s_axis[k] : P.tab.oaxis[k]; },
                  [&](int, int, double) {}, again);
        write_soa(again, flags);
    }
    if (P.left && (flags & HF_SCATTER) && !(flags & HF_SKIP)) P.left[1 + atomicAdd(P.left_n, 1)] = (int32_t)j;
    if ((flags & HF_SLOW) && !(flags & (HF_SKIP | HF_OOB))) {
        atomicAdd(&P.stats->halos_scatter_fallback, 1u);
        if (P.left_n) atomicAdd(&P.left_n[kPlanFallback], 1);
    }
    }   // j < n_halo
    // ---- row phase (what halo_row4_kernel does, same arithmetic): hwin[j][e] = sum over the corners of the halo's outer cell of
    // w_c T[c][win_lo + e] (+ ln(pixfac) for ln tables), four nodes per thread.  Done here the rows cost neither a second
    // read of the 128-byte halo records and the cell arrays nor a launch, and their 256 B per halo of stores overlap the
    // atomics of the binning in other workgroups (separate: 0.168 + 0.125 ms at 1e6 halos).
    if (!P.hwin) return;
    __syncthreads();
    {
        const DevTable &T = P.tab;
        const int W = P.win_nodes, tph = W >> 2, hpb = min(256 / tph, 64);
        const int hl = threadIdx.x / tph, q = threadIdx.x - hl * tph;
        const int ncorner = 1 << T.nouter;
        double *s_w = smem_rows;                                               // [64][ncorner]
        int64_t *s_off = reinterpret_cast<int64_t *>(smem_rows + 64 * ncorner);
        const int64_t j0 = (int64_t)blockIdx.x * blockDim.x;
        for (int h0 = 0; h0 < 256 && j0 + h0 < P.n_halo; h0 += hpb) {
            const int hh = h0 + hl;
            const bool in = hl < hpb && hh < 256 && j0 + hh < P.n_halo;
            if (in) {
                for (int cc = q; cc < ncorner; cc += tph) {                    // corner order and products of halo_row_kernel
                    double w = 1.0;
                    int64_t off = (j0 + hh) * (int64_t)T.hstride;
                    for (int k = 0; k < T.nouter; ++k) {
                        const int bit = (cc >> (T.nouter - 1 - k)) & 1;
                        const double y = s_cy[k][hh];
                        w = w * (bit ? y : 1.0 - y);
                        off += (int64_t)(s_ci[k][hh] + bit) * T.ostride[k];
                    }
                    s_w[hl * ncorner + cc] = w; s_off[hl * ncorner + cc] = off;
                }
            }
            __syncthreads();
            if (in && !(s_fl[hh] & (HF_SKIP | HF_OOB))) {
                const int e4 = q << 2;
                double b0, b1, b2, b3;
                blend_row4(T, s_w + hl * ncorner, s_off + hl * ncorner, ncorner, s_wl[hh] + e4, b0, b1, b2, b3);
                const double add = T.log_values ? s_ln[hh] : 0.0;
                double4 out;
                out.x = b0 + add; out.y = b1 + add; out.z = b2 + add; out.w = b3 + add;
                *reinterpret_cast<double4 *>(P.hwin + (j0 + hh) * W + e4) = out;
            }
            __syncthreads();
        }
    }
}

// per-group LDS layout (G entries each), followed by the (axis, value) window
template <int G>
struct RingLds {
    int32_t cum[G];
    int32_t nr[G];
    int32_t iplo[G];
    int32_t pad[G];
    int64_t start[G];
    double z[G], sth[G], phistep[G], phioff[G];
    double cwgt[kMaxCorner];      // weights / row offsets of the halo's outer-cell corners
    int64_t coff[kMaxCorner];
};

template <int G, int MODE>
__device__ __forceinline__ void scatter_halo(const ShellParams &P, const int64_t j, unsigned char *smem_raw)
{
    constexpr int GPB = 256 / G;
    const int lane = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    RingLds<G> *rl = reinterpret_cast<RingLds<G> *>(smem_raw) + grp;
    double2 *win = reinterpret_cast<double2 *>(smem_raw + sizeof(RingLds<G>) * GPB) + (size_t)grp * P.win_nodes;

    const int64_t cap = P.cap;
    const int32_t flags = P.irec[I_FLAGS * cap + j];
    if (flags & HF_SKIP) return;
    if (P.only_flagged && !(flags & HF_SCATTER)) return;
    // paint: a halo outside the table hull paints NaN -> 0 everywhere (HealpixRunner.py:473);
    // baryonify: offset NaN -> 0 (:347) -- but its pixels still count toward P_tot.
    const bool halo_oob = (flags & HF_OOB) != 0;

    const double *rec = P.rec + j;
    const double x0 = rec[F_X0 * cap], y0 = rec[F_Y0 * cap], z0v = rec[F_Z0 * cap];
    const double ptheta = rec[F_PTHETA * cap], pphi = rec[F_PPHI * cap];
    const double D = rec[F_D * cap], a = rec[F_A * cap];
    const double radius = rec[F_RADIUS * cap];
    const double Rm = rec[F_RM * cap];
    const int32_t rfirst = P.irec[I_RFIRST * cap + j], rlast = P.irec[I_RLAST * cap + j];
    const int32_t irmin = P.irec[I_IRMIN * cap + j], irmax = P.irec[I_IRMAX * cap + j];
    const Hpx &hp = P.hpx;
    const DevTable &T = P.tab;

    // ---- outer-cell corners ------------------------------------------------------
    const int ncorner = 1 << T.nouter;
    for (int c = lane; c < ncorner; c += G) {
        double w = 1.0;
        int64_t off = j * (int64_t)T.hstride;
        for (int k = 0; k < T.nouter; ++k) {
            int bit = (c >> (T.nouter - 1 - k)) & 1;
            double y = P.cw[k * cap + j];
            int i = P.cidx[k * cap + j];
            w = w * (bit ? y : 1.0 - y);
            off += (int64_t)(i + bit) * T.ostride[k];
        }
        rl->cwgt[c] = w; rl->coff[c] = off;
    }
    __builtin_amdgcn_wave_barrier();
    const double *cwgt = rl->cwgt;
    const int64_t *coff = rl->coff;
    const double r_lo = T.raxis[0], r_hi = T.raxis[T.nr - 1];

    // ---- stage the blended radial row window in LDS --------------------------------
    // largest radius any disc pixel can have -> top node of the window
    const double q_rdelta = (MODE == MODE_BARYONIFY && P.rdelta) ? log(Rm) : 0.0;
    int win_lo = 0;
    if (!halo_oob) {
        double rmax_com = 2.0 * D * sin(0.5 * fmin(radius, kPi)) / a;
        double rho_max = log(rmax_com) - q_rdelta;
        int itop = find_interval(T.raxis, T.nr, rho_max) + 1;          // top node that can be touched
        win_lo = itop - (P.win_nodes - 1);
        if (win_lo > T.nr - P.win_nodes) win_lo = T.nr - P.win_nodes;
        if (win_lo < 0) win_lo = 0;
        for (int e = lane; e < P.win_nodes; e += G) {
            int ir = win_lo + e;
            double acc = 0.0;
            for (int c = 0; c < ncorner; ++c) acc = acc + T.values[coff[c] + ir] * cwgt[c];
            win[e] = make_double2(T.raxis[ir], acc);
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- per-pixel work --------------------------------------------------------------
    const double cosrbig = cos(radius);
    const double z0 = cos(ptheta);
    const double xa = 1.0 / sqrt((1.0 - z0) * (1.0 + z0));
    const double pixfac = (MODE == MODE_PAINT && P.pixfac_area != 0.0) ? P.pixfac_area * (D * D) : 1.0;
    const double posj0 = x0 * D, posj1 = y0 * D, posj2 = z0v * D;
    unsigned long long n_r_oob = 0;

    auto process_pixel = [&](int64_t pix, double z, double sth, double phi) {
        double sphi, cphi;
        sincos(phi, &sphi, &cphi);
        const double vx = sth * cphi, vy = sth * sphi, vz = z;
        const double p0 = vx * D, p1 = vy * D, p2 = vz * D;            // pos = vec * D_j
        const double d0 = p0 - posj0, d1 = p1 - posj1, d2 = p2 - posj2;
        const double r_sep = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        const double r_com = r_sep / a;
        double val = NAN;   // interpolant (ln T for paint, d for baryonify); NaN = outside hull
        if (!halo_oob) {
            const double rho = log(r_com) - q_rdelta;
            if (rho >= r_lo && rho <= r_hi) {
                bool done = false;
                if (T.r_uniform) {
                    int i = (int)((rho - T.r0) * T.inv_dr);
                    i = i < 0 ? 0 : (i > T.nr - 2 ? T.nr - 2 : i);
                    int e = i - win_lo;
                    if (e >= 0 && e + 1 < P.win_nodes) {
                        double2 a0 = win[e], a1 = win[e + 1];
                        if (rho >= a0.x && (rho < a1.x || i == T.nr - 2)) {
                            double w = (rho - a0.x) / (a1.x - a0.x);
                            val = a0.y * (1.0 - w) + a1.y * w;
                            done = true;
                        }
                    }
                }
                if (!done) {   // generic path: exact cell by bisection, corners from L2
                    int i = find_interval(T.raxis, T.nr, rho);
                    double g0 = T.raxis[i], g1 = T.raxis[i + 1];
                    double v0 = 0.0, v1 = 0.0;
                    for (int c = 0; c < ncorner; ++c) {
                        v0 = v0 + T.values[coff[c] + i] * cwgt[c];
                        v1 = v1 + T.values[coff[c] + i + 1] * cwgt[c];
                    }
                    double w = (rho - g0) / (g1 - g0);
                    val = v0 * (1.0 - w) + v1 * w;
                }
            } else {
                ++n_r_oob;
            }
        }
        if (MODE == MODE_PAINT) {
            double v = exp(val);                                       // Tabulate.py:315
            if (!isfinite(v)) v = 0.0;                                 // HealpixRunner.py:473
            v = v * pixfac;                                            // :478
            if (v != 0.0) unsafeAtomicAdd(P.out + pix, v);             // :481
        } else {
            double d = val;
            if (!(r_com < P.eps_model * Rm)) d = 0.0;                  // BaryonCorrection.py:410-411
            d = d * a;                                                 // HealpixRunner.py:345
            double o0 = d * (d0 / r_sep), o1 = d * (d1 / r_sep), o2 = d * (d2 / r_sep);   // :346
            if (!isfinite(o0)) o0 = 0.0;                               // :347
            if (!isfinite(o1)) o1 = 0.0;
            if (!isfinite(o2)) o2 = 0.0;
            const double n0 = p0 + o0, n1 = p1 + o1, n2 = p2 + o2;     // :350
            const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);       // :351
            const double f0 = n0 / nn - vx, f1 = n1 / nn - vy, f2 = n2 / nn - vz;   // :352
            if (o0 != 0.0 || o1 != 0.0 || o2 != 0.0) {                 // zero offsets add exactly 0
                double *o = P.out + 3 * pix;
                unsafeAtomicAdd(o + 0, f0);                            // :355
                unsafeAtomicAdd(o + 1, f1);
                unsafeAtomicAdd(o + 2, f2);
            }
        }
    };

    unsigned long long my_pixels = 0;
    const int nrings = rlast - rfirst + 1;
    for (int ring0 = rfirst; ring0 <= rlast || ring0 == rfirst; ring0 += G) {
        // phase A: one lane per ring -> phi window of the disc on that ring
        const int ring = ring0 + lane;
        int cnt = 0, iplo = 0;
        RingGeom g;
        g.start = 0; g.nr = 1; g.z = 0; g.sth = 0; g.phistep = 0; g.phioff = 0;
        if (ring <= rlast) {
            g = ring_geom(hp, ring);
            if (ring < irmin || ring > irmax) {
                cnt = g.nr;              // ring completely inside the disc (pole in disc)
                iplo = 0;
            } else {
                const double z = ring2z(hp, ring);
                const double x = (cosrbig - z * z0) * xa;
                const double ysq = 1.0 - z * z - x * x;
                const double dphi = (ysq <= 0.0) ? 0.0 : atan2(sqrt(ysq), x);
                if (dphi > 0.0) {
                    const double shift = (g.phioff != 0.0) ? 0.5 : 0.0;
                    int64_t lo = (int64_t)floor((double)g.nr * kInvTwoPi * (pphi - dphi) - shift) + 1;
                    int64_t hi = (int64_t)floor((double)g.nr * kInvTwoPi * (pphi + dphi) - shift);
                    int64_t c = hi - lo + 1;
                    if (c > g.nr) c = g.nr;   // the two healpy ranges overlap -> whole ring once
                    if (c > 0) { cnt = (int)c; iplo = (int)lo; }
                }
            }
        }
        const int cum = group_inclusive_scan<G>(cnt, lane);
        const int total = __shfl(cum, G - 1, G);
        rl->cum[lane] = cum; rl->nr[lane] = g.nr; rl->iplo[lane] = iplo; rl->start[lane] = g.start;
        rl->z[lane] = g.z; rl->sth[lane] = g.sth; rl->phistep[lane] = g.phistep; rl->phioff[lane] = g.phioff;
        __builtin_amdgcn_wave_barrier();

        if (MODE == MODE_BARYONIFY && nrings <= G && total < 4) {
            // fewer than 4 pixels: use the 4 bilinear neighbours of the halo centre  (:333-334)
            if (lane < 4) {
                const double *c = P.cat + j * (int64_t)P.cat_stride;
                int64_t fp[4]; double fw[4];
                get_interpol(hp, kHalfPi - c[3] * kDeg2Rad, c[2] * kDeg2Rad, fp, fw);
                int64_t pix = fp[lane], fring, fip;
                pix2ring(hp, pix, fring, fip);
                RingGeom fg = ring_geom(hp, fring);
                process_pixel(pix, fg.z, fg.sth, ((double)fip + fg.phioff) * fg.phistep);
            }
            if (lane == 0) { my_pixels += 4; atomicAdd((unsigned long long *)&P.stats->halos_fallback4, 1ull); }
            break;
        }

        // phase B: flattened loop over the pixels of these rings
        for (int t = lane; t < total; t += G) {
            int lo = 0, hi = G - 1;        // smallest jr with cum[jr] > t
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (rl->cum[mid] > t) hi = mid; else lo = mid + 1;
            }
            const int jr = lo;
            const int before = (jr == 0) ? 0 : rl->cum[jr - 1];
            const int nr = rl->nr[jr];
            int ip = rl->iplo[jr] + (t - before);
            if (ip < 0) ip += nr;
            if (ip >= nr) ip -= nr;
            if (ip >= nr) ip -= nr;
            const int64_t pix = rl->start[jr] + ip;
            process_pixel(pix, rl->z[jr], rl->sth[jr], ((double)ip + rl->phioff[jr]) * rl->phistep[jr]);
        }
        if (lane == 0) my_pixels += (unsigned long long)total;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && my_pixels) atomicAdd((unsigned long long *)&P.stats->pixel_updates, my_pixels);
    if (n_r_oob) {
        atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_r_oob);
        atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);
    }
}

// One G-lane group per halo.  only_flagged (the tile variant's left-overs): a fixed grid strides over the compact
// list of flagged halos the prep kernel built -- or over every halo if the pair buffer overflowed and the fill kernel
// flagged them all.
template <int G, int MODE>
__global__ __launch_bounds__(256) void shell_scatter_kernel(const ShellParams P)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int GPB = 256 / G;
    const int grp = threadIdx.x / G;
    const int64_t first = (int64_t)blockIdx.x * GPB + grp, stride = (int64_t)gridDim.x * GPB;
    if (!P.left) {
        if (first < P.n_halo) scatter_halo<G, MODE>(P, first, smem_raw);
        return;
    }
    const bool all = (long long)P.pair_total_ptr[0] > P.pair_cap;
    const int64_t n = all ? P.n_halo : (int64_t)*P.left_n;
    for (int64_t it = first; it < n; it += stride) {
        scatter_halo<G, MODE>(P, all ? it : (int64_t)P.left[1 + it], smem_raw);
        __builtin_amdgcn_wave_barrier();
    }
}

// Final regrid, HealpixRunner.py:357-365: one thread per pixel.
__global__ __launch_bounds__(256) void regrid_kernel(Hpx hp, const double *__restrict__ off,
                                                     const double *__restrict__ in_map,
                                                     double *__restrict__ out_map, double *sums)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double v_in = 0.0, v_dep = 0.0;
    if (p < hp.npix) {
        const double val = in_map[p];
        if (val != 0.0) {                                              // :359
            int64_t ring, ip;
            pix2ring(hp, p, ring, ip);
            RingGeom g = ring_geom(hp, ring);
            double sphi, cphi;
            sincos(((double)ip + g.phioff) * g.phistep, &sphi, &cphi);
            const double vx = g.sth * cphi + off[3 * p + 0];           // :357
            const double vy = g.sth * sphi + off[3 * p + 1];
            const double vz = g.z + off[3 * p + 2];
            const double dnorm = sqrt(vx * vx + vy * vy + vz * vz);    // hp.vec2ang :358
            const double theta = acos(vz / dnorm);
            double phi = atan2(vy, vx);
            if (phi < 0) phi += kTwoPi;
            int64_t cp[4]; double cw[4];
            get_interpol(hp, theta, phi, cp, cw);                      // :361
            for (int k = 0; k < 4; ++k) {
                const double d = cw[k] * val;                          // :64-68
                if (d != 0.0) unsafeAtomicAdd(out_map + cp[k], d);
                v_dep += d;
            }
        }
        v_in = val;
    }
    if (sums) {
        for (int o = 32; o > 0; o >>= 1) { v_in += __shfl_down(v_in, o, 64); v_dep += __shfl_down(v_dep, o, 64); }
        __shared__ double s_in[4], s_dep[4];
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { s_in[w] = v_in; s_dep[w] = v_dep; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsafeAtomicAdd(sums + 0, s_in[0] + s_in[1] + s_in[2] + s_in[3]);
            unsafeAtomicAdd(sums + 1, s_dep[0] + s_dep[1] + s_dep[2] + s_dep[3]);
        }
    }
}

// Tile-privatised regrid: one workgroup per sky tile of SOURCE pixels (the 64-ring paint tiles).  Displacements
// are a few pixels at most, so almost every deposit lands inside the tile grown by a halo of kRgHalo rings /
// pixels: those are accumulated in LDS (ds_add_f64) and flushed once with row-contiguous global atomics
// (~1.4 per pixel instead of 4 scattered ones); the rare far deposit goes straight to a global atomic.
#ifndef BFG_RG_HALO
#define BFG_RG_HALO 3
#endif
constexpr int kRgHalo = BFG_RG_HALO;
constexpr int kRgRows = 64 + 2 * kRgHalo;
constexpr int kRgWidth = kTileWidth + 2 * kRgHalo;
struct __align__(16) RgRow {
    int64_t start;                   // first pixel of the ring
    int32_t nr, istart, w, shifted;  // ring length; first ring index of the LDS row (may be < 0: modulo nr); row width
    double theta;                    // ring colatitude as get_ring_info2 gives it (get_interpol's theta1 / theta2)
    double z, sth, phistep;          // pixel-centre geometry of the ring (ring_geom); phi offset = 0.5 where `shifted`, else 0
    double inv_dth;                  // 1 / (colatitude of the next ring - this ring's); 0 for the last ring
};
static_assert(sizeof(RgRow) == 64, "RgRow is four 16-byte pieces");

// Waves per SIMD: the long dependent chain per pixel (sincos -> sqrt -> atan2 x 2 -> divisions) needs wavefronts to overlap, spills
// cost more.  Round 1's body: 184 VGPRs spill-free (2 waves) 0.43 ms, 128 VGPRs + 172 B of scratch (4 waves) 0.33 ms.  With the
// table-driven atan2, the fast sqrt and multiplications by nr / 2 pi in place of four divisions the body fits 168 VGPRs without
// scratch: 3 waves 0.29 ms, 4 waves (148 B of scratch) 0.30 ms (profiles/r02_regrid_ab.txt).
#ifndef BFG_RG_WAVES
#define BFG_RG_WAVES 3
#endif
// IdxT: the type of the slow list's entries -- int32_t while 12 NSIDE^2 < 2^31 (NSIDE <= 8192: 4 B per pixel of list instead of 8)
template <typename IdxT>
__global__ __launch_bounds__(256, BFG_RG_WAVES) void regrid_tile_kernel(Hpx hp, TileGeom geo, const double *__restrict__ off,
                                                          const double *__restrict__ in_map,
                                                          double *__restrict__ out_map, double *sums,
                                                          int no_shortcut, int tile0, IdxT *__restrict__ slow_list,
                                                          unsigned long long *slow_n)
{
    __shared__ double acc[kRgRows * kRgWidth];
    __shared__ RgRow rows[kRgRows];
    __shared__ double s_in[4], s_dep[4];
    const int tile = tile0 + (int)blockIdx.x;      // (a launch over the tiles of some bands only: bfg_regrid_shell_bands)
    const int band = geo.tile_band[tile];
    const int sector = tile - geo.band_tile0[band];
    const int NS = geo.band_ns[band];
    const int nl4 = (int)(4 * hp.nside);
    const int ring_lo = 1 + band * geo.tr;
    const int ring_hi = min(nl4 - 1, ring_lo + geo.tr - 1);
    const int row_ring0 = ring_lo - kRgHalo;                       // ring of LDS row 0 (may be < 1)
    const int tid = threadIdx.x;
    for (int i = tid; i < kRgRows * kRgWidth; i += 256) acc[i] = 0.0;
    if (tid < kRgRows) {
        const int ring = row_ring0 + tid;
        RgRow r;
        r.start = 0; r.nr = 0; r.istart = 0; r.w = 0; r.shifted = 0; r.theta = 0; r.z = 0; r.sth = 0; r.phistep = 0;
        r.inv_dth = 0;
        if (ring >= 1 && ring <= nl4 - 1 && ring <= ring_hi + kRgHalo) {
            int64_t sp, nr; bool sh; double th;
            ring_info2(hp, ring, sp, nr, th, sh);
            if (ring + 1 <= nl4 - 1) {
                int64_t sp2_, nr2_; bool sh2_; double th2_;
                ring_info2(hp, ring + 1, sp2_, nr2_, th2_, sh2_);
                r.inv_dth = 1.0 / (th2_ - th);
            }
            const int k0 = (int)(((int64_t)sector * nr) / NS), k1 = (int)(((int64_t)(sector + 1) * nr) / NS);
            r.start = sp; r.nr = (int)nr; r.shifted = sh ? 1 : 0; r.theta = th;
            r.w = min(k1 - k0 + 2 * kRgHalo, (int)nr);
            r.w = min(r.w, kRgWidth);
            r.istart = k0 - kRgHalo;                                // may be negative: taken modulo nr
            const RingGeom g = ring_geom(hp, ring);
            r.z = g.z; r.sth = g.sth; r.phistep = g.phistep;         // (g.phioff == 0.5 exactly where sh: same parity rule)
        }
        rows[tid] = r;
    }
    __syncthreads();

    // deposit d on ring index i of ring row lr / pixel sp + i
    auto deposit = [&](int lr, int64_t sp, int i, double d) {
        if (d == 0.0) return;
        bool local = lr >= 0;
        int rel = 0;
        if (local) {
            const RgRow &t = rows[lr];
            rel = i - t.istart;
            if (rel >= t.nr) rel -= t.nr;
            if (rel < 0) rel += t.nr;
            local = (rel >= 0) && (rel < t.w);
        }
        if (local) unsafeAtomicAdd(&acc[lr * kRgWidth + rel], d);
        else unsafeAtomicAdd(out_map + sp + i, d);                   // (same ring rows, beyond the window's width)
    };

    double v_in = 0.0, v_dep = 0.0;
    for (int e = tid; e < geo.tr * kTileWidth; e += 256) {
        const int row = e / kTileWidth, col = e % kTileWidth;
        const int ring = ring_lo + row;
        if (ring > ring_hi) break;
        const RgRow &sr = rows[row + kRgHalo];
        const int k0 = sr.istart + kRgHalo;
        const int k1 = (int)(((int64_t)(sector + 1) * sr.nr) / NS);
        const int ip = k0 + col;
        if (ip >= k1) continue;
        const int64_t p = sr.start + ip;
        const double val = in_map[p];
        v_in += val;
        if (val == 0.0) continue;                                      // :359
        const double ox = off[3 * p + 0], oy = off[3 * p + 1], oz = off[3 * p + 2];
        // A pixel no halo moved stays at its own centre, where the bilinear weights are (1, 0, 0, 0): its mass goes back to itself.
        // (The general code below gives the same up to the rounding of acos / atan2 -- weights like 1 - 1e-13 and 1e-13, noise the
        // reference carries as well --, at ~400 instructions per pixel; on a sparse shell most pixels take this exit: regrid at
        // NSIDE 1024 behind 1e4 halos 0.298 -> 0.253 ms.  No difference from 1e5 halos up -- and what is left there is not this
        // arithmetic but the loads (32 B per pixel) and the ~1.3 flush atomics per pixel: profiles/r03_regrid_ab.txt.)
        if (ox == 0.0 && oy == 0.0 && oz == 0.0 && !(no_shortcut & 1)) {
            deposit(row + kRgHalo, sr.start, ip, val);
            v_dep += val;
            continue;
        }
        double sphi, cphi;
        const double phi_p = ((double)ip + (sr.shifted ? 0.5 : 0.0)) * sr.phistep;
        sincos_2pi(phi_p, sphi, cphi);
        // ---- the usual case: a displacement of a fraction of a pixel, away from the poles.  The displaced direction is the pixel's
        // own centre (theta_p, phi_p) plus small angles, so hp.vec2ang (:358) is taken DIFFERENTIALLY: with the offset rotated into
        // the pixel's meridian plane, tan(dphi) = cross / dot and tan(dtheta) = (rho z_p - v_z sin theta_p) / (rho sin theta_p +
        // v_z z_p), rho = dot sqrt(1 + tan^2 dphi) -- two reciprocals and three short series (|tan| <= 2^-7: truncation < 1e-16)
        // instead of two square roots, a division and two full-range atan2; and healpix_cxx's get_interpol (:361) needs no
        // ring_above(z): the ring pair is found by comparing dtheta with the colatitudes of the neighbouring rings in the LDS row
        // table, 1 / (theta_2 - theta_1) comes from there too.  Same weights as the general code (regrid_list_kernel) to ~1e-13 (both
        // are continuous in the angles; the reference's own acos / atan2 carry rounding of that order): ~280 instead of ~500
        // instructions per pixel, and this kernel is VALU-issue bound (profiles/r04_sq_counters_bary1e5_*.txt).
        if (!(no_shortcut & 2)) {
            const double dotp = sr.sth + (ox * cphi + oy * sphi);      // rho cos(dphi)
            const double crs = oy * cphi - ox * sphi;                  // rho sin(dphi)
            const double vzz = sr.z + oz;
            if (dotp > 0.0 && fabs(crs) <= 0.0078125 * dotp) {
                const double u = crs * rcp_newton(dotp), u2 = u * u;
                double pa = fma(u2, -1.0 / 7.0, 0.2);
                pa = fma(u2, pa, -1.0 / 3.0);
                const double dph = fma(u * u2, pa, u);                 // atan(u)
                double pr = fma(u2, 0.0625, -0.125);
                pr = fma(u2, pr, 0.5);
                const double rho = fma(dotp * u2, pr, dotp);           // dot sqrt(1 + u^2)
                const double aa = rho * sr.z - vzz * sr.sth, bb = rho * sr.sth + vzz * sr.z;
                if (bb > 0.25 && fabs(aa) <= 0.0078125 * bb) {
                    const double tq = aa * rcp_newton(bb), tq2 = tq * tq;
                    double pt = fma(tq2, -1.0 / 7.0, 0.2);
                    pt = fma(tq2, pt, -1.0 / 3.0);
                    const double dth = fma(tq * tq2, pt, tq);          // theta - theta_p
                    int lrA = row + kRgHalo + ((dth >= 0.0) ? 0 : -1);
                    double phi = phi_p + dph;
                    if (phi < 0.0) phi += kTwoPi;
                    // (phi == 2 pi after rounding is healpix_cxx's unwrapped-index case: left to the general code, which keeps it)
                    bool found = false;
                    double dA = 0.0;
#pragma unroll
                    for (int it = 0; it < 5 && !found && phi < kTwoPi; ++it) {     // (the row table reaches kRgHalo rings either way)
                        if (lrA < 0 || lrA + 1 >= kRgRows) break;
                        const RgRow &ta = rows[lrA], &tb = rows[lrA + 1];
                        if (ta.nr <= 0 || tb.nr <= 0) break;
                        dA = ta.theta - sr.theta;
                        const double dB = tb.theta - sr.theta;
                        if (dth < dA) lrA -= 1;
                        else if (dth >= dB) lrA += 1;
                        else found = true;
                    }
                    if (found) {
                        const RgRow &ta = rows[lrA], &tb = rows[lrA + 1];
                        const double wtheta = (dth - dA) * ta.inv_dth;
                        auto ring_w = [&](const RgRow &t, int &ia, int &ib, double &ww) {
                            const double tmp = fma(phi, (double)t.nr * kInvTwoPi, t.shifted ? -0.5 : 0.0);
                            const double fl = floor(tmp);
                            ia = (int)fl;
                            ww = tmp - fl;
                            ib = ia + 1;
                            if (ia < 0) ia += t.nr;
                            if (ia >= t.nr) ia -= t.nr;
                            if (ib >= t.nr) ib -= t.nr;
                        };
                        int a0, a1, b0, b1;
                        double wa, wb;
                        ring_w(ta, a0, a1, wa);
                        ring_w(tb, b0, b1, wb);
                        const double d0 = ((1 - wa) * (1 - wtheta)) * val, d1 = (wa * (1 - wtheta)) * val;
                        const double d2 = ((1 - wb) * wtheta) * val, d3 = (wb * wtheta) * val;   // :64-68
                        v_dep += d0; v_dep += d1; v_dep += d2; v_dep += d3;
                        // the two deposits of a ring share its row record (already in registers) and the index arithmetic
                        auto deposit2 = [&](int lr, const RgRow &t, int ia, int ib, double da, double db) {
                            int ra = ia - t.istart;
                            ra += (ra < 0) ? t.nr : 0;
                            ra -= (ra >= t.nr) ? t.nr : 0;
                            int rb = ib - t.istart;
                            rb += (rb < 0) ? t.nr : 0;
                            rb -= (rb >= t.nr) ? t.nr : 0;
                            double *base = &acc[lr * kRgWidth];
                            if (da != 0.0) { if (ra < t.w) unsafeAtomicAdd(base + ra, da); else unsafeAtomicAdd(out_map + t.start + ia, da); }
                            if (db != 0.0) { if (rb < t.w) unsafeAtomicAdd(base + rb, db); else unsafeAtomicAdd(out_map + t.start + ib, db); }
                        };
                        deposit2(lrA, ta, a0, a1, d0, d1);
                        deposit2(lrA + 1, tb, b0, b1, d2, d3);
                        continue;
                    }
                }
            }
        }
        // Everything else -- next to the poles, displaced by more than a few pixels, phi == 2 pi -- is left to regrid_list_kernel
        // (healpix_cxx's get_interpol in full, one thread per pixel, global atomics): rare, and keeping that code out of THIS kernel is
        // what lets it run at 68 instead of 168 VGPRs -- five workgroups per CU (LDS-limited) instead of three wavefronts per SIMD:
        // 0.265 -> 0.183 ms at NSIDE 1024 (profiles/r04_regrid_split_ab.txt).
        // (the wavefront's lanes that get here reserve their slots together: one atomic per wavefront, not per pixel -- with
        // BFG_REGRID=general, or a shell displaced by several pixels everywhere, every pixel comes this way)
        {
            const unsigned long long m = __ballot(1);
            const int lane_ = tid & 63, leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0ull;
            if (lane_ == leader) base = atomicAdd(slow_n, (unsigned long long)__popcll(m));
            base = __shfl(base, leader, 64);
            slow_list[base + (unsigned long long)__popcll(m & ((1ull << lane_) - 1ull))] = (IdxT)p;
        }
    }
    __syncthreads();
    for (int i = tid; i < kRgRows * kRgWidth; i += 256) {
        const double v = acc[i];
        if (v == 0.0) continue;
        const RgRow &t = rows[i / kRgWidth];
        int ii = t.istart + (i % kRgWidth);
        if (ii < 0) ii += t.nr;
        if (ii >= t.nr) ii -= t.nr;
        unsafeAtomicAdd(out_map + t.start + ii, v);
    }
    if (sums) {
        for (int o = 32; o > 0; o >>= 1) { v_in += __shfl_down(v_in, o, 64); v_dep += __shfl_down(v_dep, o, 64); }
        const int w = tid >> 6;
        if ((tid & 63) == 0) { s_in[w] = v_in; s_dep[w] = v_dep; }
        __syncthreads();
        if (tid == 0) {
            unsafeAtomicAdd(sums + 0, s_in[0] + s_in[1] + s_in[2] + s_in[3]);
            unsafeAtomicAdd(sums + 1, s_dep[0] + s_dep[1] + s_dep[2] + s_dep[3]);
        }
    }
}

// The pixels regrid_tile_kernel leaves aside (next to the poles, displaced by more than a few pixels, phi == 2 pi): healpix_cxx's
// get_interpol in full, one thread per pixel of the list, global atomics -- the arithmetic of regrid_kernel.  band_rings > 0
// (bfg_regrid_shell_bands): a deposit more than kRgHalo rings outside the source pixel's band is counted in *far.
template <typename IdxT>
__global__ __launch_bounds__(256) void regrid_list_kernel(Hpx hp, const IdxT *__restrict__ list, const unsigned long long *n_ptr,
                                                          const double *__restrict__ off, const double *__restrict__ in_map,
                                                          double *__restrict__ out_map, double *sums, int band_rings, double *far)
{
    const int64_t n = (int64_t)*n_ptr;
    double v_dep = 0.0;
    bool far_hit = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = (int64_t)list[e];
        const double val = in_map[p];
        int64_t ring, ip;
        pix2ring(hp, p, ring, ip);
        RingGeom g = ring_geom(hp, ring);
        double sphi, cphi;
        sincos(((double)ip + g.phioff) * g.phistep, &sphi, &cphi);
        const double vx = g.sth * cphi + off[3 * p + 0];               // :357
        const double vy = g.sth * sphi + off[3 * p + 1];
        const double vz = g.z + off[3 * p + 2];
        const double dnorm = sqrt(vx * vx + vy * vy + vz * vz);        // hp.vec2ang :358
        const double theta = acos(vz / dnorm);
        double phi = atan2(vy, vx);
        if (phi < 0) phi += kTwoPi;
        int64_t cp[4]; double cw[4];
        get_interpol(hp, theta, phi, cp, cw);                          // :361
        int64_t lo_pix = 0, hi_pix = hp.npix;
        if (band_rings > 0) {                                          // the pixel range within kRgHalo rings of the source's band
            const int64_t band = (ring - 1) / band_rings;
            const int64_t r0 = 1 + band * band_rings - kRgHalo, r1 = band * band_rings + band_rings + kRgHalo + 1;
            int64_t nr_; bool sh_;
            if (r0 >= 1) ring_info_small(hp, r0, lo_pix, nr_, sh_);
            if (r1 <= 4 * hp.nside - 1) ring_info_small(hp, r1, hi_pix, nr_, sh_);
        }
        for (int k = 0; k < 4; ++k) {
            const double d = cw[k] * val;                              // :64-68
            if (d != 0.0) {
                unsafeAtomicAdd(out_map + cp[k], d);
                far_hit = far_hit || cp[k] < lo_pix || cp[k] >= hi_pix;
            }
            v_dep += d;
        }
    }
    if (far && far_hit) unsafeAtomicAdd(far, 1.0);                       // (rare: a displacement of more than kRgHalo rings)
    if (sums && n > 0) {
        for (int o = 32; o > 0; o >>= 1) v_dep += __shfl_down(v_dep, o, 64);
        if ((threadIdx.x & 63) == 0 && v_dep != 0.0) unsafeAtomicAdd(sums + 1, v_dep);
    }
}

__device__ inline double nanmax(double a, double b)
{
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}

__global__ __launch_bounds__(256) void reduce_absmax_sum_kernel(int64_t n, const double *__restrict__ x, double *red)
{
    double s = 0.0, m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = x[i];
        s += v;
        m = nanmax(m, fabs(v));     // NaN is sticky, like np.allclose failing on it
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        m = nanmax(m, __shfl_down(m, o, 64));
    }
    __shared__ double ss[4], sm[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { ss[w] = s; sm[w] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = ss[0] + ss[1] + ss[2] + ss[3];
        double Mx = sm[0];
        for (int k = 1; k < 4; ++k) Mx = nanmax(Mx, sm[k]);
        unsafeAtomicAdd(red + 0, S);
        // absmax: values are >= 0 or NaN -> integer max on the bit pattern is order preserving
        atomicMax((unsigned long long *)(red + 1), (unsigned long long)__double_as_longlong(Mx));
    }
}

__global__ void table_eval_kernel(DevTable T, int64_t npts, const double *__restrict__ coords, double *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    const double *x = coords + p * (int64_t)T.ndim;   // (z, M, r, extras...)
    bool oob = false;
    int idx[BFG_MAX_DIM]; double y[BFG_MAX_DIM];
    for (int k = 0; k < T.nouter; ++k) {
        double xv = (k < 2) ? x[k] : x[k + 1];
        const double *g = T.oaxis[k]; int n = T.oshape[k];
        if (!(xv >= g[0]) || !(xv <= g[n - 1])) oob = true;
        idx[k] = find_interval(g, n, xv);
        y[k] = (xv - g[idx[k]]) / (g[idx[k] + 1] - g[idx[k]]);
    }
    const double rho = x[2];
    if (!(rho >= T.raxis[0]) || !(rho <= T.raxis[T.nr - 1])) oob = true;
    const int ir = find_interval(T.raxis, T.nr, rho);
    const double wr = (rho - T.raxis[ir]) / (T.raxis[ir + 1] - T.raxis[ir]);
    double v0 = 0.0, v1 = 0.0;
    const int ncorner = 1 << T.nouter;
    for (int c = 0; c < ncorner; ++c) {
        double w = 1.0; int64_t off = 0;
        for (int k = 0; k < T.nouter; ++k) {
            int bit = (c >> (T.nouter - 1 - k)) & 1;
            w = w * (bit ? y[k] : 1.0 - y[k]);
            off += (int64_t)(idx[k] + bit) * T.ostride[k];
        }
        v0 = v0 + T.values[off + ir] * w;
        v1 = v1 + T.values[off + ir + 1] * w;
    }
    double v = v0 * (1.0 - wr) + v1 * wr;
    if (oob) v = NAN;
    out[p] = T.log_values ? exp(v) : v;
}

// ------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------
extern "C" {

int bfg_abi_version(void) { return BFG_ABI_VERSION; }

const char *bfg_status_string(int s)
{
    switch (s) {
    case BFG_OK: return "ok";
    case BFG_ERR_INVALID: return "invalid argument";
    case BFG_ERR_HIP: return "HIP runtime error";
    case BFG_ERR_NO_DEVICE: return "no usable gfx950 device";
    case BFG_ERR_UNSUPPORTED: return "unsupported configuration";
    case BFG_ERR_NOMEM: return "out of memory";
    case BFG_ERR_COMM: return "RCCL error";
    default: return "unknown status";
    }
}

const char *bfg_last_error(void) { return g_last_error.c_str(); }

int bfg_device_count(int *count)
{
    if (!count) return BFG_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return BFG_OK;
}

// Every entry point makes the context's GPU current for the duration of the call and puts the caller's device back on
// return: a process that drives several GPUs through several contexts keeps its own notion of the current device.
struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static int ctx_enter(bfg_ctx *c, DeviceGuard &g)
{
    if (!c) return BFG_ERR_INVALID;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    if (cur != c->device) {
        HIP_TRY(hipSetDevice(c->device));
        g.prev = cur;
    }
    return BFG_OK;
}

static int ctx_create_body(bfg_ctx *c, void *stream)
{
    if (stream == BFG_STREAM_OWN) { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    else { c->stream = (hipStream_t)stream; c->own_stream = false; }   // NULL = the legacy default stream
    HIP_TRY(hipMalloc((void **)&c->d_stats, sizeof(bfg_stats)));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(bfg_stats), c->stream));
    HIP_TRY(hipMalloc((void **)&c->d_red, 4 * sizeof(double)));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming));
    {   // ln / exp / atan tables of the tile kernels: {1/c, ln c} with c = 1 + (i + 0.5)/128, 2^(j/64), atan(k/64)
        std::vector<double> mt(2 * kLogTab + kExpTab + kAtanTab, 0.0);
        for (int i = 0; i < kLogTab; ++i) {
            double cc = 1.0 + ((double)i + 0.5) / (double)kLogTab;
            mt[2 * i] = 1.0 / cc; mt[2 * i + 1] = std::log(cc);
        }
        for (int j = 0; j < kExpTab; ++j) mt[2 * kLogTab + j] = std::exp2((double)j / (double)kExpTab);
        for (int k = 0; k <= 64; ++k) mt[2 * kLogTab + kExpTab + k] = std::atan((double)k / 64.0);
        HIP_TRY(hipMalloc((void **)&c->d_mathtab, mt.size() * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->d_mathtab, mt.data(), mt.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMalloc((void **)&c->d_pair_total, sizeof(unsigned long long)));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return BFG_OK;
}

static void ctx_free_all(bfg_ctx *c);

int bfg_ctx_create(int device_id, void *stream, bfg_ctx **out)
{
    if (!out) return BFG_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        g_last_error = "hipGetDeviceCount found no device";
        return BFG_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) return BFG_ERR_INVALID;
    DeviceGuard dg_;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
    HIP_TRY(hipSetDevice(device_id));
    if (cur != device_id) dg_.prev = cur;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    bfg_ctx *c = new bfg_ctx();
    std::memset((void *)c, 0, sizeof(*c));
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    c->lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    c->max_dyn_lds = prop.sharedMemPerBlock;
    for (int k = 0; k < kTimingSlots; ++k) { c->ev_a[k] = new std::vector<hipEvent_t>(); c->ev_b[k] = new std::vector<hipEvent_t>(); }
    const int rc = ctx_create_body(c, stream);
    if (rc != BFG_OK) { ctx_free_all(c); return rc; }          // nothing of a half-built context is left behind
    *out = c;
    return BFG_OK;
}

static void ctx_free_all(bfg_ctx *c)
{
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_irec) (void)hipFree(c->d_irec);
    if (c->d_cidx) (void)hipFree(c->d_cidx);
    if (c->d_cw) (void)hipFree(c->d_cw);
    if (c->d_ht) (void)hipFree(c->d_ht);
    if (c->d_hwin) (void)hipFree(c->d_hwin);
    if (c->d_ndrows) (void)hipFree(c->d_ndrows);
    if (c->d_ndsort) (void)hipFree(c->d_ndsort);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_red) (void)hipFree(c->d_red);
    for (int m = 0; m < 3; ++m) {
        if (c->tiles[m].d_geo) (void)hipFree(c->tiles[m].d_geo);
        if (c->tiles[m].d_tile_count) (void)hipFree(c->tiles[m].d_tile_count);
        if (c->tiles[m].d_tile_start) (void)hipFree(c->tiles[m].d_tile_start);
        if (c->tiles[m].d_work) (void)hipFree(c->tiles[m].d_work);
        if (c->tiles[m].d_nwork) (void)hipFree(c->tiles[m].d_nwork);
        if (c->tiles[m].d_counters) (void)hipFree(c->tiles[m].d_counters);
        if (c->tiles[m].d_defer) (void)hipFree(c->tiles[m].d_defer);
        if (c->tiles[m].d_defer_count) (void)hipFree(c->tiles[m].d_defer_count);
        if (c->tiles[m].d_shared) (void)hipFree(c->tiles[m].d_shared);
        if (c->tiles[m].d_slices) (void)hipFree(c->tiles[m].d_slices);
    }
    if (c->d_hd) (void)hipFree(c->d_hd);
    if (c->d_left) (void)hipFree(c->d_left);
    if (c->d_rg_slow) (void)hipFree(c->d_rg_slow);
    if (c->d_rg_slow_n) (void)hipFree(c->d_rg_slow_n);
    for (int k = 0; k < 8; ++k) if (c->snap_buf[k]) (void)hipFree(c->snap_buf[k]);
    for (int k = 0; k < 6; ++k) if (c->grid_buf[k]) (void)hipFree(c->grid_buf[k]);
    for (int k = 0; k < 5; ++k) if (c->dep_buf[k]) (void)hipFree(c->dep_buf[k]);
    if (c->d_pairs) (void)hipFree(c->d_pairs);
    if (c->d_ovf_mask) (void)hipFree(c->d_ovf_mask);
    if (c->d_mathtab) (void)hipFree(c->d_mathtab);
    if (c->d_pair_total) (void)hipFree(c->d_pair_total);
    for (int k = 0; k < kTimingSlots; ++k) {
        if (!c->ev_a[k]) continue;
        for (hipEvent_t e : *c->ev_a[k]) (void)hipEventDestroy(e);
        for (hipEvent_t e : *c->ev_b[k]) (void)hipEventDestroy(e);
        delete c->ev_a[k]; delete c->ev_b[k];
    }
    if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
    if (c->comm) bfg_comm_release(c);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bfg_ctx_destroy(bfg_ctx *c)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    (void)hipStreamSynchronize(c->stream);
    ctx_free_all(c);
    return BFG_OK;
}

// Re-bind the context to another stream of its GPU (e.g. torch's current stream at call time).  The context's
// workspaces are shared by consecutive calls, so the new stream is made to wait for everything the context has
// enqueued on the old one (an event, no host synchronisation): calls on alternating streams stay ordered.
int bfg_ctx_set_stream(bfg_ctx *c, void *stream)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (stream == BFG_STREAM_OWN) return BFG_ERR_INVALID;
    hipStream_t ns = (hipStream_t)stream;
    if (ns == c->stream) return BFG_OK;
    HIP_TRY(hipEventRecord(c->ev_switch, c->stream));
    HIP_TRY(hipStreamWaitEvent(ns, c->ev_switch, 0));
    if (c->own_stream) { HIP_TRY(hipStreamSynchronize(c->stream)); (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = ns;
    return BFG_OK;
}

int bfg_ctx_synchronize(bfg_ctx *c)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFG_OK;
}

int bfg_ctx_device_info(bfg_ctx *c, char *name, int name_len, int *n_cu, int *lds, int64_t *hbm)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    if (name && name_len > 0) { std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds) *lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    return BFG_OK;
}

int bfg_dev_malloc(bfg_ctx *c, size_t bytes, void **d_ptr)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!d_ptr) return BFG_ERR_INVALID;
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
    return BFG_OK;
}

int bfg_dev_free(bfg_ctx *c, void *d_ptr)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(d_ptr));
    return BFG_OK;
}

int bfg_memcpy_h2d(bfg_ctx *c, void *d_dst, const void *src, size_t bytes)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFG_OK;
}

int bfg_memcpy_d2h(bfg_ctx *c, void *dst, const void *d_src, size_t bytes)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return BFG_OK;
}

int bfg_dev_memset_zero(bfg_ctx *c, void *d_ptr, size_t bytes)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, c->stream));
    return BFG_OK;
}

// ---- tables -------------------------------------------------------------------------
constexpr int kNdMaxOuterHost = 12;              // = bfg::kNdMaxOuter (bfg_ndtable.hpp): z, M and up to 10 p_keys axes
int bfg_table_create(bfg_ctx *c, int ndim, const int64_t *shape, const double *const *axes,
                     const double *values, uint32_t flags, bfg_table **out)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!shape || !axes || !values || !out) return BFG_ERR_INVALID;
    if (ndim < 3) return BFG_ERR_INVALID;
    // More dimensions than the shell kernels read (BFG_MAX_DIM): the table is kept as an N-dimensional one (bfg_ndtable) and the shell
    // calls blend every halo's radial row first (nd_rows_kernel), then run on those rows (DevTable::hstride); see run_shell_nd.
    // (A displacement table with three p_keys axes takes the row path as well: measured at 1e6 halos, NSIDE 1024, 3-node axes --
    // tools/nd_probe.py, profiles/r05_nd_tables.txt -- offsets 8.07 ms with the kernels blending 32 corners per window node against
    // 3.64 ms on rows; paint tables and narrower displacement tables are faster read directly: 2.27 vs 2.86 ms, 2.95 vs 2.95 ms.)
    // With the halos grouped by table cell (nd_rows_blocked_kernel) the rows also beat the kernels' own corner blend for a paint table
    // with three extra axes (2.31 -> 1.78 ms) and for displacement tables with one or two (2.93 -> 2.53 ms) -- where a cell holds
    // several halos.  Such tables keep BOTH forms and run_shell decides per call (nd_rows_pay); a paint table with one or two extra
    // axes stays direct (1.30 vs 1.77 ms on rows).
    bool nd = ndim > BFG_MAX_DIM || ndim == BFG_MAX_DIM || (!(flags & BFG_TABLE_LOG_VALUES) && ndim >= 4);
    if (const char *e = std::getenv("BFG_ND_FROM_DIM")) nd = ndim > BFG_MAX_DIM || ndim >= std::max(4, std::atoi(e));   // A/B switch
    if (nd && ndim - 1 > kNdMaxOuterHost) return BFG_ERR_UNSUPPORTED;
    int64_t total = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape[d] < 2 || shape[d] > (1 << 24)) return BFG_ERR_INVALID;
        for (int64_t i = 1; i < shape[d]; ++i)
            if (!(axes[d][i] > axes[d][i - 1])) return BFG_ERR_INVALID;   // strictly ascending, no NaN
        total *= shape[d];
    }
    bfg_table *t = new bfg_table();
    t->shape.assign(shape, shape + ndim);
    for (int d = 0; d < 2; ++d) t->ax_inv_h[d] = (double)(shape[d] - 1) / (axes[d][shape[d] - 1] - axes[d][0]);
    {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *p_, size_t n_) { const unsigned char *b = (const unsigned char *)p_; for (size_t i = 0; i < n_; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(&ndim, sizeof(ndim)); mix(&flags, sizeof(flags));
        for (int d = 0; d < ndim; ++d) { mix(&shape[d], sizeof(shape[d])); mix(axes[d], (size_t)shape[d] * sizeof(double)); }
        t->axes_hash = h;
    }
    // permute values so that r (dim 2) is the fastest axis: [z][M][extras...][r]
    const int nouter = ndim - 1;
    std::vector<int> odim(nouter);   // outer k -> original dim
    odim[0] = 0; odim[1] = 1;
    for (int k = 2; k < nouter; ++k) odim[k] = k + 1;
    const int64_t NR = shape[2];
    std::vector<int64_t> src_stride(ndim);
    { int64_t s = 1; for (int d = ndim - 1; d >= 0; --d) { src_stride[d] = s; s *= shape[d]; } }
    std::vector<double> perm((size_t)total);
    std::vector<int64_t> ostride(nouter);
    { int64_t s = NR; for (int k = nouter - 1; k >= 0; --k) { ostride[k] = s; s *= shape[odim[k]]; } }
    const int64_t nrows = total / NR;
    for (int64_t row = 0; row < nrows; ++row) {
        int64_t rem = row, src = 0;
        for (int k = nouter - 1; k >= 0; --k) {
            int64_t n = shape[odim[k]];
            int64_t i = rem % n; rem /= n;
            src += i * src_stride[odim[k]];
        }
        for (int64_t ir = 0; ir < NR; ++ir) perm[(size_t)(row * NR + ir)] = values[src + ir * src_stride[2]];
    }
    DevTable &D = t->dev;
    std::memset(&D, 0, sizeof(D));
    D.ndim = ndim; D.nouter = nouter; D.nr = (int)NR;
    D.log_values = (flags & BFG_TABLE_LOG_VALUES) ? 1 : 0;
    t->d_blob = nullptr;
    if (nd) {
        // the N-dimensional table proper (axes and permuted values on the device) ...
        std::vector<int64_t> oshape(nouter);
        std::vector<const double *> oaxes(nouter);
        for (int k = 0; k < nouter; ++k) { oshape[k] = shape[odim[k]]; oaxes[k] = axes[odim[k]]; }
        rc = bfg_ndtable_create(c, nouter, oshape.data(), oaxes.data(), NR, axes[2], perm.data(), &t->nd);
        if (rc) { delete t; return rc; }
    }
    if (ndim > BFG_MAX_DIM) {
        // ... and, where the kernels cannot read the table themselves, a DevTable that only describes per-halo rows: no outer axes,
        // the radial axis, `values` filled in per call (a table of at most BFG_MAX_DIM dimensions that takes the row path in the
        // shell calls keeps its directly readable form too: the grid and snapshot calls and bfg_table_eval use that one)
        D.nouter = 0;
        D.raxis = ndtable_raxis(t->nd);
        const double *r = axes[2];
        double dr = (r[NR - 1] - r[0]) / (double)(NR - 1);
        bool uni = dr > 0;
        for (int64_t i = 0; i < NR && uni; ++i)
            if (std::fabs(r[i] - (r[0] + dr * (double)i)) > 1e-9 * dr) uni = false;
        D.r_uniform = uni ? 1 : 0; D.r0 = r[0]; D.inv_dr = uni ? 1.0 / dr : 0.0;
        if (D.log_values)
            for (double v : perm) if (std::isfinite(v) && std::fabs(v) > 650.0) { D.hot = 1; break; }
        D.hstride = (int)NR;
        *out = t;
        return BFG_OK;
    }
    // one device blob: [outer axes...][r axis][values]
    size_t n_axes = 0;
    for (int d = 0; d < ndim; ++d) n_axes += (size_t)shape[d];
    std::vector<double> blob(n_axes + (size_t)total);
    if (hipMalloc((void **)&t->d_blob, blob.size() * sizeof(double)) != hipSuccess) {
        (void)hipGetLastError();
        if (t->nd) (void)bfg_ndtable_destroy(c, t->nd);
        delete t; return BFG_ERR_NOMEM;
    }
    size_t pos = 0;
    for (int k = 0; k < nouter; ++k) {
        int d = odim[k];
        std::copy(axes[d], axes[d] + shape[d], blob.begin() + pos);
        D.oaxis[k] = t->d_blob + pos; D.oshape[k] = (int)shape[d]; D.ostride[k] = ostride[k];
        pos += (size_t)shape[d];
    }
    std::copy(axes[2], axes[2] + NR, blob.begin() + pos);
    D.raxis = t->d_blob + pos; pos += (size_t)NR;
    std::copy(perm.begin(), perm.end(), blob.begin() + pos);
    D.values = t->d_blob + pos;
    // is the radial axis a linspace (geomspace in r) to rounding?
    const double *r = axes[2];
    double dr = (r[NR - 1] - r[0]) / (double)(NR - 1);
    bool uni = dr > 0;
    for (int64_t i = 0; i < NR && uni; ++i)
        if (std::fabs(r[i] - (r[0] + dr * (double)i)) > 1e-9 * dr) uni = false;
    D.r_uniform = uni ? 1 : 0; D.r0 = r[0]; D.inv_dr = uni ? 1.0 / dr : 0.0;
    if (D.log_values)
        for (double v : perm) if (std::isfinite(v) && std::fabs(v) > 650.0) { D.hot = 1; break; }
    if (hipMemcpyAsync(t->d_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        g_last_error = std::string("bfg_table_create upload: ") + hipGetErrorString(hipGetLastError());
        if (t->nd) (void)bfg_ndtable_destroy(c, t->nd);
        (void)hipFree(t->d_blob); delete t;
        return BFG_ERR_HIP;
    }
    *out = t;
    return BFG_OK;
}

int bfg_table_destroy(bfg_ctx *c, bfg_table *t)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!t) return BFG_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    if (t->nd) (void)bfg_ndtable_destroy(c, t->nd);
    if (t->d_blob) (void)hipFree(t->d_blob);
    delete t;
    return BFG_OK;
}

int bfg_table_eval(bfg_ctx *c, const bfg_table *t, int64_t npts, const double *coords, double *out)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!t || !coords || !out || npts < 0) return BFG_ERR_INVALID;
    if (!t->d_blob) return BFG_ERR_UNSUPPORTED;     // (more than BFG_MAX_DIM dimensions: bfg_ndtable_rows / bfg_ndtable_read)
    if (npts == 0) return BFG_OK;
    double *d_c = nullptr, *d_o = nullptr;
    size_t nb = (size_t)npts * t->dev.ndim * sizeof(double);
    HIP_TRY(hipMalloc((void **)&d_c, nb));
    HIP_TRY(hipMalloc((void **)&d_o, (size_t)npts * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(d_c, coords, nb, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(table_eval_kernel, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, c->stream,
                       t->dev, npts, d_c, d_o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_o, (size_t)npts * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(d_c); (void)hipFree(d_o);
    return BFG_OK;
}

// ---- spline ----------------------------------------------------------------------------
int bfg_spline_create(bfg_ctx *c, int n, const double *knots, const double *coef, bfg_spline **out)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (n < 2 || !knots || !coef || !out) return BFG_ERR_INVALID;
    for (int i = 1; i < n; ++i) if (!(knots[i] > knots[i - 1])) return BFG_ERR_INVALID;
    bfg_spline *s = new bfg_spline();
    s->n = n; s->d_knots = nullptr; s->d_coef = nullptr;
    s->inv_h = (n >= 2 && knots[n - 1] > knots[0]) ? (double)(n - 1) / (knots[n - 1] - knots[0]) : 0.0;
    if (hipMalloc((void **)&s->d_knots, (size_t)n * sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&s->d_coef, (size_t)4 * (n - 1) * sizeof(double)) != hipSuccess ||
        hipMemcpyAsync(s->d_knots, knots, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(s->d_coef, coef, (size_t)4 * (n - 1) * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        g_last_error = std::string("bfg_spline_create: ") + hipGetErrorString(hipGetLastError());
        if (s->d_knots) (void)hipFree(s->d_knots);
        if (s->d_coef) (void)hipFree(s->d_coef);
        delete s;
        return BFG_ERR_HIP;
    }
    *out = s;
    return BFG_OK;
}

int bfg_spline_destroy(bfg_ctx *c, bfg_spline *s)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!s) return BFG_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(s->d_knots); (void)hipFree(s->d_coef);
    delete s;
    return BFG_OK;
}

// ---- hot path ----------------------------------------------------------------------------
static int ensure_workspace(bfg_ctx *c, int64_t n)
{
    if (n <= c->cap_halo) return BFG_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_rec) { (void)hipFree(c->d_rec); (void)hipFree(c->d_irec); (void)hipFree(c->d_cidx); (void)hipFree(c->d_cw); (void)hipFree(c->d_ht); (void)hipFree(c->d_hd); (void)hipFree(c->d_left); (void)hipFree(c->d_ovf_mask); }
    c->d_rec = nullptr; c->d_ht = nullptr; c->d_hd = nullptr; c->d_left = nullptr; c->d_ovf_mask = nullptr; c->cap_halo = 0;
    int64_t cap = (n + 1023) / 1024 * 1024;
    HIP_TRY(hipMalloc((void **)&c->d_rec, (size_t)cap * F_NF * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_irec, (size_t)cap * I_NI * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&c->d_cidx, (size_t)cap * (BFG_MAX_DIM - 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&c->d_cw, (size_t)cap * (BFG_MAX_DIM - 1) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_ht, (size_t)cap * sizeof(HaloTile)));
    HIP_TRY(hipMalloc((void **)&c->d_hd, (size_t)cap * sizeof(HaloDisp)));
    HIP_TRY(hipMalloc((void **)&c->d_left, (size_t)(cap + 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&c->d_ovf_mask, (size_t)cap * sizeof(unsigned long long)));
    c->cap_halo = cap;
    return BFG_OK;
}

static void timing_fold(bfg_ctx *c, int which)
{
    for (size_t i = 0; i < c->ev_used[which]; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize((*c->ev_b[which])[i]) == hipSuccess &&
            hipEventElapsedTime(&ms, (*c->ev_a[which])[i], (*c->ev_b[which])[i]) == hipSuccess) {
            c->t_ms[which] += ms; c->t_n[which] += 1;
        }
    }
    c->ev_used[which] = 0;
}

static void timing_begin(bfg_ctx *c, int which)
{
    if (!c->timing || !((c->timing_mask >> which) & 1u)) return;
    if (c->ev_used[which] >= 4096) timing_fold(c, which);
    if (c->ev_used[which] >= c->ev_a[which]->size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { c->timing = false; return; }
        c->ev_a[which]->push_back(a); c->ev_b[which]->push_back(b);
    }
    (void)hipEventRecord((*c->ev_a[which])[c->ev_used[which]], c->stream);
}

static void timing_end(bfg_ctx *c, int which)
{
    if (!c->timing || !((c->timing_mask >> which) & 1u)) return;
    (void)hipEventRecord((*c->ev_b[which])[c->ev_used[which]], c->stream);
    c->ev_used[which] += 1;
}

// sectors per band: the longest ring of the band cut into pieces of at most tw pixels
static int band_sectors_host(int64_t nside, int tr, int tw, int b)
{
    const int64_t nrings = 4 * nside - 1;
    const int64_t lo = 1 + (int64_t)b * tr, hi = std::min<int64_t>(nrings, lo + tr - 1);
    int64_t mx = 0;
    for (int64_t r = lo; r <= hi; ++r) mx = std::max(mx, (r < nside) ? 4 * r : (r <= 3 * nside ? 4 * nside : 4 * (4 * nside - r)));
    return (int)((mx + tw - 1) / tw);
}

// tile geometry of one (nside, rings-per-tile) (cached per mode) and the binning buffers
constexpr int kRegridSet = 2, kRegridTR = 64;     // regrid_tile_kernel: 64-ring x kTileWidth tiles of source pixels
static int ensure_tiles(bfg_ctx *c, int mode, int tr, int tw, int64_t nside, int64_t n_halo)
{
    bfg_ctx::TileSet &ts = c->tiles[mode];
    if (ts.nside != nside || ts.geo.tr != tr || ts.geo.tw != tw) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (ts.d_geo) { (void)hipFree(ts.d_geo); (void)hipFree(ts.d_tile_count); (void)hipFree(ts.d_tile_start); (void)hipFree(ts.d_work); (void)hipFree(ts.d_nwork); }
        if (ts.d_defer) { (void)hipFree(ts.d_defer); (void)hipFree(ts.d_defer_count); }
        if (ts.d_shared) (void)hipFree(ts.d_shared);
        if (ts.d_slices) (void)hipFree(ts.d_slices);
        if (ts.d_counters) (void)hipFree(ts.d_counters);
        ts.d_shared = nullptr; ts.d_slices = nullptr; ts.d_counters = nullptr;
        ts.d_geo = nullptr; ts.d_tile_count = nullptr; ts.d_tile_start = nullptr; ts.d_work = nullptr; ts.d_nwork = nullptr; ts.nside = 0;
        ts.d_defer = nullptr; ts.d_defer_count = nullptr;
        const int64_t nrings = 4 * nside - 1;
        const int nbands = (int)((nrings + tr - 1) / tr);
        std::vector<int32_t> ns(nbands), t0(nbands + 1), nrmin(nbands);
        int ntiles = 0;
        for (int b = 0; b < nbands; ++b) {
            int64_t lo = 1 + (int64_t)b * tr, hi = std::min<int64_t>(nrings, lo + tr - 1);
            int64_t mx = 0, mn = INT64_MAX;
            for (int64_t r = lo; r <= hi; ++r) {
                int64_t nr = (r < nside) ? 4 * r : (r <= 3 * nside ? 4 * nside : 4 * (4 * nside - r));
                mx = std::max(mx, nr); mn = std::min(mn, nr);
            }
            ns[b] = (int32_t)((mx + tw - 1) / tw);
            nrmin[b] = (int32_t)mn;
            t0[b] = ntiles;
            ntiles += ns[b];
        }
        t0[nbands] = ntiles;
        std::vector<int32_t> blob;
        blob.insert(blob.end(), ns.begin(), ns.end());
        blob.insert(blob.end(), t0.begin(), t0.end());
        blob.insert(blob.end(), nrmin.begin(), nrmin.end());
        for (int b = 0; b < nbands; ++b) for (int s = 0; s < ns[b]; ++s) blob.push_back(b);
        HIP_TRY(hipMalloc((void **)&ts.d_geo, blob.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(ts.d_geo, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMalloc((void **)&ts.d_tile_count, (size_t)2 * (ntiles + kTileTail) * sizeof(int32_t)));   // per set: + the left-over list's length, needs_scan, plan statistics
        HIP_TRY(hipMemsetAsync(ts.d_tile_count, 0, (size_t)2 * (ntiles + kTileTail) * sizeof(int32_t), c->stream));
        ts.flip = 0; ts.counting = false;
        HIP_TRY(hipMalloc((void **)&ts.d_tile_start, (size_t)(ntiles + 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&ts.d_work, (size_t)(2 * ntiles + kWorkExtra) * 2 * sizeof(int4)));
        HIP_TRY(hipMalloc((void **)&ts.d_nwork, 2 * sizeof(int32_t)));     // [0] items in the work list
        HIP_TRY(hipMalloc((void **)&ts.d_counters, kCounterInts * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&ts.d_shared, (size_t)ntiles * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void **)&ts.d_slices, (size_t)2 * kMaxSlices * sizeof(int32_t)));
        if (mode == MODE_PAINT) {
            // a failed allocation (the list is ~1 KB per work item) only means the tile workgroups drain their own queues
            const size_t items = (size_t)(2 * ntiles + kWorkExtra);
            if (hipMalloc((void **)&ts.d_defer, items * kDeferCap * sizeof(bfg::DeferredOut)) != hipSuccess) { (void)hipGetLastError(); ts.d_defer = nullptr; }
            else if (hipMalloc((void **)&ts.d_defer_count, items * sizeof(int32_t)) != hipSuccess) {
                (void)hipGetLastError(); (void)hipFree(ts.d_defer); ts.d_defer = nullptr; ts.d_defer_count = nullptr;
            }
        }
        ts.geo.tr = tr; ts.geo.tw = tw; ts.geo.nbands = nbands; ts.geo.ntiles = ntiles;
        ts.geo.band_ns = ts.d_geo;
        ts.geo.band_tile0 = ts.d_geo + nbands;
        ts.geo.band_nrmin = ts.d_geo + 2 * nbands + 1;
        ts.geo.tile_band = ts.d_geo + 3 * nbands + 1;
        ts.nside = nside;
    }
    // (the regrid kernel bins nothing: it must not resize -- or re-cap -- the pair buffer a shell call's plan lives in)
    if (mode == kRegridSet) return BFG_OK;
    // pair buffer: cap_direct fixed slots per tile (filled by the count pass) followed by the overflow lists.  Slots for
    // ~8x the mean number of halos per tile hold every pair of a catalog spread over the sky (4-5 pairs per halo).
    int64_t cap_direct = 32;
    while (cap_direct < 8 * n_halo / std::max(ts.geo.ntiles, 1) && cap_direct < 8192) cap_direct *= 2;
    while (cap_direct > 32 && cap_direct * ts.geo.ntiles > ((int64_t)1 << 28)) cap_direct /= 2;
    if (const char *tc = std::getenv("BFG_TILE_CAP")) cap_direct = std::max<int64_t>(1, std::atoll(tc));   // test hook
    ts.cap_direct = (int)cap_direct;
    int64_t want = std::min<int64_t>(8 * n_halo + 65536, ((int64_t)1 << 31) - 1 - cap_direct * ts.geo.ntiles);   // positions are 32-bit
    if (const char *pc = std::getenv("BFG_PAIR_CAP")) want = std::max<int64_t>(1, std::atoll(pc));   // test hook: tiny overflow region
    c->pair_cap = want;
    const int64_t alloc = cap_direct * ts.geo.ntiles + want;
    if (alloc > c->pairs_alloc) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_pairs) (void)hipFree(c->d_pairs);
        c->d_pairs = nullptr; c->pairs_alloc = 0;
        HIP_TRY(hipMalloc((void **)&c->d_pairs, (size_t)alloc * sizeof(int32_t)));
        c->pairs_alloc = alloc;
    }
    return BFG_OK;
}

static int check_args(const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s, const void *d_out)
{
    if (!a || !t || !s || !d_out) return BFG_ERR_INVALID;
    if (a->nside < 1 || a->nside > (1 << 20)) return BFG_ERR_INVALID;
    if (a->n_halo < 0) return BFG_ERR_INVALID;
    if (a->n_halo > 0 && !a->d_catalog) return BFG_ERR_INVALID;
    if (a->n_extra < 0 || a->n_extra > BFG_MAX_EXTRA) return BFG_ERR_UNSUPPORTED;
    if (a->cat_stride < 4) return BFG_ERR_INVALID;
    if (t->dev.hstride == 0 && a->cat_stride < 4 + a->n_extra) return BFG_ERR_INVALID;
    if (t->dev.ndim != 3 + a->n_extra) return BFG_ERR_INVALID;
    if (t->nd && !t->d_blob) return BFG_ERR_INVALID; // (run_shell_nd hands run_shell the per-halo rows, never an N-dimensional table the kernels cannot read)
    if (!(a->epsilon_max >= 0)) return BFG_ERR_INVALID;
    return BFG_OK;
}

// Sliced calls: the left-over scatter kernel runs BEFORE the tile kernels (a slice of the output must be final when its tile
// launch ends).  If there is anything left over (usually not) the output is cleared here, the scatter kernel adds to it and the
// tile kernels add their tiles instead of storing them (TileParams::accum_left); otherwise this kernel returns at once.
__global__ __launch_bounds__(256) void out_clear_if_left_kernel(double *out, int64_t n, const int32_t *left_n, const int32_t *pair_total,
                                                                long long pair_cap)
{
    if (!(*left_n > 0 || (long long)*pair_total > pair_cap)) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0.0;
}

// first RING pixel of ring `ring` (1 .. 4 nside - 1; 4 nside -> npix)
static int64_t ring_first_pixel(int64_t nside, int64_t ring)
{
    const int64_t npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1);
    if (ring >= 4 * nside) return npix;
    if (ring < nside) return 2 * ring * (ring - 1);
    if (ring < 3 * nside) return ncap + (ring - nside) * 4 * nside;
    const int64_t nr = 4 * nside - ring;
    return npix - 2 * nr * (nr + 1);
}

// The element ranges of a sliced call's K slices: cuts[0] = 0 < ... < cuts[K] = all elements; K = min(n_slices, 16, bands of the
// loop's tile geometry), every slice a run of whole bands.  A function of (nside, mode, n_slices) only.
static int shell_slice_cuts(int64_t nside, int mode, int n_slices, int64_t *cuts)
{
    const int tr = (mode == MODE_PAINT) ? TileCfg<MODE_PAINT>::TR : TileCfg<MODE_BARYONIFY>::TR;
    const int nbands = (int)((4 * nside - 1 + tr - 1) / tr);
    const int K = std::max(1, std::min(std::min(n_slices, kMaxSlices), nbands));
    for (int k = 0; k <= K; ++k)
        cuts[k] = (mode == MODE_PAINT ? 1 : 3) * ring_first_pixel(nside, 1 + ((int64_t)nbands * k / K) * tr);
    return K;
}

int bfg_shell_slice_cuts(int64_t nside, int offsets, int n_slices, int64_t *elem_cuts, int *n_out)
{
    if (nside < 1 || nside > (1 << 20) || n_slices < 1 || !elem_cuts || !n_out) return BFG_ERR_INVALID;
    *n_out = shell_slice_cuts(nside, offsets ? MODE_BARYONIFY : MODE_PAINT, n_slices, elem_cuts);
    return BFG_OK;
}

static int run_shell(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s,
                     double *d_out, int mode, int n_slices = 1, bfg_slice_fn slice_fn = nullptr, void *slice_user = nullptr)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (t && t->nd && nd_rows_pay(t, a)) return run_shell_nd(c, a, t, s, d_out, mode, n_slices, slice_fn, slice_user);
    rc = check_args(a, t, s, d_out);
    if (rc) return rc;
    // whatever plan the context holds is void from here on (workspaces may be resized below, a failure may leave them half-built);
    // it is reinstated at the end of a tile-path call that ran to its end.  A call without halos touches nothing and keeps it.
    const bool had_plan = c->plan.valid && a->n_halo > 0;
    if (a->n_halo > 0) c->plan.valid = false;
    if (mode == MODE_PAINT && !t->dev.log_values) return BFG_ERR_INVALID;
    if (mode == MODE_BARYONIFY && t->dev.log_values) return BFG_ERR_INVALID;
    // BFG_SHELL_OUT_OVERWRITE: the output is uninitialised.  The tile kernels write every pixel themselves; every other route
    // (no halos, scatter variants, the wave-chunk kernel) clears it here first.
    const size_t out_bytes = (size_t)12 * a->nside * a->nside * sizeof(double) * (mode == MODE_PAINT ? 1 : 3);
    bool overwrite = (a->flags & BFG_SHELL_OUT_OVERWRITE) != 0;
    bool out_zero = (a->flags & BFG_SHELL_OUT_IS_ZERO) != 0;
    const int64_t out_elems = (int64_t)(out_bytes / sizeof(double));
    // The slices of a sliced call are a function of (nside, mode, n_slices) ONLY -- never of the catalog: the ranks of a process group
    // issue one collective per callback, so a rank with an empty shard (or one whose call takes a scatter variant) must report the
    // same K ranges as its peers.  The cuts are runs of whole bands of the tile geometry (contiguous RING pixel ranges).
    const int slice_tr = (mode == MODE_PAINT) ? TileCfg<MODE_PAINT>::TR : TileCfg<MODE_BARYONIFY>::TR;
    const int slice_nbands = (int)((4 * a->nside - 1 + slice_tr - 1) / slice_tr);
    int64_t slice_elem[kMaxSlices + 1];
    const int slice_K = slice_fn ? shell_slice_cuts(a->nside, mode, n_slices, slice_elem) : 0;
    // a call whose kernels are not launched per slice (no halos, scatter variants, the wave kernel) reports the same K ranges, all
    // of them final once the stream gets there
    auto whole_output = [&]() -> int {
        for (int k = 0; k < slice_K; ++k)
            if (slice_fn(slice_user, k, slice_K, slice_elem[k], slice_elem[k + 1]) != 0) { g_last_error = "the slice callback failed"; return BFG_ERR_INVALID; }
        return BFG_OK;
    };
    if (a->n_halo == 0) {
        if (overwrite) HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
        return whole_output();
    }
    rc = ensure_workspace(c, a->n_halo);
    if (rc) return rc;

    int variant = a->variant;
    // the tile variant needs a uniform radial axis (direct cell computation)
    // pair counts and positions are 32-bit: at most kMaxPairsPerHalo = 64 pairs per halo -> n_halo <= 2^31 / 64 on the tile path
    const bool tile_ok = t->dev.r_uniform && !t->dev.hot && a->nside >= 8 && a->nside <= (1 << 24) &&
                         a->n_halo <= ((1ll << 31) / kMaxPairsPerHalo - 1);
    if (variant == BFG_VARIANT_AUTO) variant = tile_ok ? BFG_VARIANT_TILE_LDS : BFG_VARIANT_SCATTER_QUARTER;
    if (variant == BFG_VARIANT_TILE_LDS && !tile_ok) variant = BFG_VARIANT_SCATTER_QUARTER;
    const bool tile = (variant == BFG_VARIANT_TILE_LDS);
    if (const char *e = std::getenv("BFG_OUT_ZERO")) out_zero = std::atoi(e) != 0;                // A/B switches
    if (const char *e = std::getenv("BFG_OUT_OVERWRITE")) overwrite = overwrite && std::atoi(e) != 0;
    if (overwrite && !tile) {
        HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
        overwrite = false; out_zero = true;
    }
    if ((a->flags & BFG_SHELL_OUT_OVERWRITE) && !overwrite && tile) {       // switched off by the environment
        HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
        out_zero = true;
    }
    // row window of the tile path: ~3.5 e-folds of radius below the disc edge (r_max/33 .. r_max)
    int win_nodes = 0;
    bool blend = false;          // row windows blended per pair inside the tile kernel (set below)
    bool blend_rows = false;     // ... from the halos' own rows (run_shell_nd)
    bool win_table = false;      // finely sampled radial axis: no row windows, the pixel stage reads the table (bfg_tile.hpp)
    if (tile) {
        win_nodes = (int)std::ceil(3.5 * t->dev.inv_dr) + 2;
        win_nodes = std::max(8, std::min(win_nodes, 256));
        if (win_nodes <= kWinLds + kWinLds / 4) win_nodes = std::min(win_nodes, kWinLds);   // fits the LDS staging
        win_nodes = (int)std::min<int64_t>(win_nodes, t->dev.nr);
        win_table = win_nodes > kWinLds && (2 << t->dev.nouter) <= kWinLds;         // weights + offsets fit the pair's LDS slot
        if (const char *e = std::getenv("BFG_WINDOWS")) if (!std::strcmp(e, "hbm")) win_table = false;   // A/B: windows in HBM
        if (win_table) win_nodes = (int)t->dev.nr;                   // the "window" is the whole axis
        rc = ensure_tiles(c, mode, mode == MODE_PAINT ? TileCfg<MODE_PAINT>::TR : TileCfg<MODE_BARYONIFY>::TR,
                          mode == MODE_PAINT ? TileCfg<MODE_PAINT>::TW : TileCfg<MODE_BARYONIFY>::TW, a->nside, a->n_halo);
        if (rc) return rc;
        // Paint, 3-D tables with 32-node windows: no row windows in HBM at all, the tile kernel's stage b blends them per pair from the
        // L2-resident table (the BLEND instantiation of shell_tile_kernel; BFG_BLEND=0: the windows of round 2, built by the prep
        // kernel and fetched by LDS-DMA).  Measured (profiles/r03_blend_ab.txt): 1e6 halos step 1.27 -> 1.14 ms (prep 0.223 ->
        // 0.156, tile kernel 1.01 -> 0.95), 1e5 halos 0.242 -> 0.235, steep mass function 0.533 -> 0.446.
        // (per-halo rows of an N-dimensional table -- no outer axes, hstride > 0 --: the same instantiation's ROWS variant copies the
        // windows straight out of the rows)
        blend_rows = t->dev.nouter == 0 && t->dev.hstride > 0;
        const bool can_blend = mode == MODE_PAINT && !win_table && win_nodes == kWinLds && (t->dev.nouter == 2 || blend_rows);
        blend = can_blend;
        if (const char *e = std::getenv("BFG_BLEND")) blend = can_blend && std::atoi(e) != 0;
        const int64_t want = (win_table || blend) ? 0 : a->n_halo * (int64_t)win_nodes;
        if (want > c->hwin_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_hwin) (void)hipFree(c->d_hwin);
            c->d_hwin = nullptr; c->hwin_cap = 0;
            HIP_TRY(hipMalloc((void **)&c->d_hwin, (size_t)want * sizeof(double)));
            c->hwin_cap = want;
        }
    }
    const double pixfac_area = a->include_pixel_size ? 4.0 * kPi / (double)(12 * a->nside * a->nside) : 0.0;

    // launch geometry of the tile kernel (host arithmetic only; decided here because a reused plan must have been built for the same one)
    bool light = false;
    int items_max = 0, persist = 0, tile_grid = 0, n_counters = 1;
    if (tile) {
        const bfg_ctx::TileSet &ts = c->tiles[mode];
        // grid of the tile kernel: persistent (a few workgroups per CU looping over the work list) unless BFG_TILE_PERSIST=0
        const bool win_lds_path = win_nodes <= kWinLds;
        // Sparse catalogs take the 256-thread instantiation with four (paint) / three (offsets) workgroups per CU (TileCfg<., 1>):
        // below ~6 (paint) / ~10 (offsets) halos per sky tile an item is a chain of latencies, and more items in flight beat larger
        // chunks (profiles/r03_light_ab.txt: paint 1e4 halos at NSIDE 1024 0.072 -> 0.057 ms, offsets 1e4 halos 0.170 -> 0.115,
        // offsets 1e5 halos 0.292 -> 0.283; paint 1e5 halos 0.140 -> 0.160, so not there).  BFG_TILE_LIGHT=0 / 1 forces either.
        const double halos_per_tile = (double)a->n_halo / (double)std::max(1, ts.geo.ntiles);
        light = win_lds_path && halos_per_tile < (mode == MODE_PAINT ? 6.0 : 10.0);
        if (const char *e = std::getenv("BFG_TILE_LIGHT")) light = win_lds_path && std::atoi(e) != 0;
        items_max = 2 * ts.geo.ntiles + kWorkExtra;
        persist = c->n_cu * (light ? (mode == MODE_PAINT ? 4 : 3) : 2);
        if (const char *e = std::getenv("BFG_TILE_PERSIST")) persist = std::atoi(e) > 1 ? std::atoi(e) : (std::atoi(e) ? persist : 0);
        tile_grid = persist > 0 ? std::min(persist, items_max) : items_max;
        // item counters of the persistent grid (bfg_tile.hpp: one address serialises the hand-out); BFG_ITEM_COUNTERS=1: the A/B
        n_counters = 8;
        if (const char *e = std::getenv("BFG_ITEM_COUNTERS")) n_counters = std::atoi(e);
        n_counters = std::max(1, std::min(std::min(n_counters, kMaxCounters), tile_grid));
    }
    // ---- BFG_SHELL_REUSE_PLAN: the per-halo records (HaloTile / HaloDisp / SoA rows), the halo -> tile pair lists and the work list
    // are functions of the catalog, the geometry (nside, epsilon, mass definitions, D_A spline) and the table's AXES -- not of its
    // values.  A call that carries the flag and finds the context's last tile-path call built from exactly these runs neither
    // halo_prep_kernel nor the binning kernels: one small launch (plan_reinit_kernel) re-arms the item counters, and -- where windows
    // are pre-blended (offsets, tables with extra axes) -- halo_row4_kernel blends them from THIS table.  What the library cannot
    // check is that the catalog buffer still holds the same records: that is what the caller vouches for with the flag.
    bfg_ctx::ShellPlanKey key;
    std::memset(&key, 0, sizeof(key));
    key.cat = a->d_catalog; key.spline = s; key.n_knots = s->n; key.n_halo = a->n_halo; key.nside = a->nside; key.axes_hash = t->axes_hash;
    key.eps = a->epsilon_max; key.eps_model = (mode == MODE_BARYONIFY) ? a->model_epsilon_max : 0.0; key.pixfac_area = pixfac_area;
    key.md_run = a->runner_md; key.md_run.reserved = 0;
    if (mode == MODE_BARYONIFY) { key.md_model = a->model_md; key.md_model.reserved = 0; }
    key.cat_stride = a->cat_stride; key.n_extra = a->n_extra; key.rdelta = (mode == MODE_BARYONIFY) ? a->rdelta_sampling : 0; key.mode = mode;
    key.win_nodes = win_nodes; key.win_table = win_table; key.blend = blend; key.blend_rows = blend_rows;
    key.overwrite = overwrite; key.out_zero = out_zero; key.slice_K = slice_fn ? slice_K : 0; key.hstride = t->dev.hstride;
    key.n_knots = s->n + 1000 * (light ? 1 : 0) + 10000 * n_counters;          // (launch geometry folded in: a changed A/B switch misses)
    int direct_limit = 0;
    if (tile) {
        // tiles of up to 512 pairs are one work item each and need no scan (256: the 1e6-halo headline, ~115 pairs per tile with
        // a tail beyond 256, paid the 0.02 ms single-workgroup scan -- 0.043 ms on the offsets tiles -- for items no better balanced)
        direct_limit = std::min(c->tiles[mode].cap_direct, 512);
        if (const char *e = std::getenv("BFG_DIRECT_LIMIT")) direct_limit = std::min(c->tiles[mode].cap_direct, std::atoi(e));
        if (std::getenv("BFG_TILE_SCAN")) direct_limit = 0;                   // A/B: always the scan kernel
        key.cap_direct = c->tiles[mode].cap_direct; key.direct_limit = direct_limit; key.pair_cap = c->pair_cap;
    }
    bool reuse = tile && (a->flags & BFG_SHELL_REUSE_PLAN) && had_plan && t->dev.hstride == 0 &&
                 std::memcmp(&key, &c->plan.key, sizeof(key)) == 0;
    if (const char *e = std::getenv("BFG_PLAN_REUSE")) if (!std::atoi(e)) reuse = false;               // A/B switch
    int32_t *const plan_tail = reuse ? c->plan.tail : nullptr;

    PrepParams pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.hpx = make_hpx(a->nside);
    pp.n_halo = a->n_halo; pp.cap = c->cap_halo;
    pp.cat = a->d_catalog; pp.cat_stride = a->cat_stride; pp.n_extra = a->n_extra;
    pp.eps_run = a->epsilon_max;
    pp.md_run = a->runner_md; pp.md_model = a->model_md;
    pp.spl_n = s->n; pp.spl_knots = s->d_knots; pp.spl_coef = s->d_coef;
    pp.tab = t->dev;
    if (!std::getenv("BFG_NO_HINT")) { pp.spl_inv_h = s->inv_h; pp.ax_inv_h[0] = t->ax_inv_h[0]; pp.ax_inv_h[1] = t->ax_inv_h[1]; }   // A/B switch
    pp.rec = c->d_rec; pp.irec = c->d_irec; pp.cidx = c->d_cidx; pp.cw = c->d_cw;
    pp.stats = c->d_stats;
    pp.want_model_radius = (mode == MODE_BARYONIFY);
    pp.ht = tile ? c->d_ht : nullptr; pp.win_nodes = win_nodes; pp.pixfac_area = pixfac_area;
    if (tile) {
        bfg_ctx::TileSet &tsw = c->tiles[mode];
        if (!reuse && tsw.counting) {           // left dirty by a call that failed half-way: start from clean counters
            HIP_TRY(hipMemsetAsync(tsw.d_tile_count, 0, (size_t)2 * (tsw.geo.ntiles + kTileTail) * sizeof(int32_t), c->stream));
            tsw.flip = 0;
        }
        if (!reuse) tsw.counting = true;
        // this call's counters (all zero) -- or, on a reused plan, the planning call's set: its tile counts are spent, the entries
        // behind them (left-over halos, needs_scan, plan statistics) stay valid until the next planning call's scan clears the set
        int32_t *const tile_count = reuse ? plan_tail - tsw.geo.ntiles
                                          : tsw.d_tile_count + (size_t)tsw.flip * (tsw.geo.ntiles + kTileTail);
        pp.bin.geo = c->tiles[mode].geo; pp.bin.tile_count = tile_count;
        pp.bin.tile_start = c->tiles[mode].d_tile_start;
        pp.bin.pairs = c->d_pairs; pp.bin.pair_total = c->d_pair_total; pp.bin.pair_cap = c->pair_cap;
        pp.bin.cap_direct = c->tiles[mode].cap_direct; pp.bin.ovf_mask = c->d_ovf_mask;
        pp.bin.mode = mode;
        pp.hd = (mode == MODE_BARYONIFY) ? c->d_hd : nullptr;
        pp.eps_model = a->model_epsilon_max; pp.rdelta = a->rdelta_sampling;
        // the tile counters and, right behind them, the length of the left-over list and the needs_scan flag
        pp.bin.needs_scan = tile_count + c->tiles[mode].geo.ntiles + 1;
        pp.bin.direct_limit = direct_limit;
        pp.left = c->d_left; pp.left_n = tile_count + c->tiles[mode].geo.ntiles;
    }
    // row windows of 4 k nodes are built by the prep kernel itself (BFG_ROWS=separate: by halo_row4_kernel, the A/B)
    bool fuse_rows = tile && !win_table && !blend && win_nodes % 4 == 0 && win_nodes >= 8;
    if (const char *e = std::getenv("BFG_ROWS")) if (!std::strcmp(e, "separate")) fuse_rows = false;
    if (reuse) fuse_rows = false;                // no prep kernel: the windows of THIS table come from halo_row4_kernel
    // the row phase needs 64 x 2^nouter x 16 B of dynamic LDS on top of the kernel's 34 KB of static LDS: tables with three extra
    // axes would pass 64 KB -- those take the separate row kernel
    if (fuse_rows && 15360 + kPrepRowLds + (size_t)64 * ((size_t)1 << t->dev.nouter) * 16 > std::min<size_t>(c->max_dyn_lds, 65536)) fuse_rows = false;
    pp.hwin = fuse_rows ? c->d_hwin : nullptr;
    pp.lazy_soa = (tile && t->dev.nouter == 2 && !std::getenv("BFG_EAGER_SOA")) ? 1 : 0;
    const size_t prep_lds = fuse_rows ? kPrepRowLds + (size_t)64 * ((size_t)1 << t->dev.nouter) * 16 : (t->dev.nouter > 2 ? kPrepRowLds : 0);
    if (!reuse) {
        timing_begin(c, 0);
        hipLaunchKernelGGL(halo_prep_kernel, dim3((unsigned)((a->n_halo + 255) / 256)), dim3(256), prep_lds, c->stream, pp);
        HIP_TRY(hipGetLastError());
        timing_end(c, 0);
    }

    ShellParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.hpx = pp.hpx; sp.n_halo = a->n_halo; sp.cap = c->cap_halo;
    sp.rec = c->d_rec; sp.irec = c->d_irec; sp.cidx = c->d_cidx; sp.cw = c->d_cw;
    sp.cat = a->d_catalog; sp.cat_stride = a->cat_stride;
    sp.tab = t->dev;
    sp.eps_model = a->model_epsilon_max; sp.rdelta = a->rdelta_sampling;
    sp.pixfac_area = pixfac_area;
    sp.out = d_out; sp.stats = c->d_stats;

    // the global-atomic kernel: every halo (scatter variants) or the tile path's left-overs (sp.left, set below)
    auto launch_scatter = [&]() -> int {
        const int G = (variant == BFG_VARIANT_SCATTER_WAVE) ? 64 : 16;
        const int tslot = (variant == BFG_VARIANT_TILE_LDS) ? 4 : 1;
        const int gpb = 256 / G;
        // LDS: ring records + per-group (axis, value) window; keep a block at or under 64 KiB
        const size_t ring_bytes = (G == 64 ? sizeof(RingLds<64>) : sizeof(RingLds<16>)) * (size_t)gpb;
        size_t budget = 64 * 1024 - ring_bytes;
        int win = (int)std::min<int64_t>(t->dev.nr, (int64_t)(budget / (sizeof(double2) * (size_t)gpb)));
        if (win < 2) return BFG_ERR_UNSUPPORTED;
        sp.win_nodes = win;
        const size_t lds = ring_bytes + (size_t)win * sizeof(double2) * (size_t)gpb;
        unsigned grid = (unsigned)((a->n_halo + gpb - 1) / gpb);
        if (sp.left) grid = std::min(grid, 512u);       // fixed grid striding over the left-over list (usually empty); two blocks per CU fit
        timing_begin(c, tslot);
        if (mode == MODE_PAINT) {
            if (G == 64) hipLaunchKernelGGL((shell_scatter_kernel<64, MODE_PAINT>), dim3(grid), dim3(256), lds, c->stream, sp);
            else hipLaunchKernelGGL((shell_scatter_kernel<16, MODE_PAINT>), dim3(grid), dim3(256), lds, c->stream, sp);
        } else {
            if (G == 64) hipLaunchKernelGGL((shell_scatter_kernel<64, MODE_BARYONIFY>), dim3(grid), dim3(256), lds, c->stream, sp);
            else hipLaunchKernelGGL((shell_scatter_kernel<16, MODE_BARYONIFY>), dim3(grid), dim3(256), lds, c->stream, sp);
        }
        HIP_TRY(hipGetLastError());
        timing_end(c, tslot);
        return BFG_OK;
    };

    if (tile) {
        timing_begin(c, 3);
        const bfg_ctx::TileSet &ts = c->tiles[mode];
        // sliced call: cut the tiles into n_slices runs of whole bands (contiguous ring ranges = contiguous RING pixel ranges)
        SliceCuts cuts;
        std::memset(&cuts, 0, sizeof(cuts));
        if (slice_fn && slice_K > 1 && persist > 0) {
            const int K = slice_K;
            if (ts.geo.nbands != slice_nbands || ts.geo.tr != slice_tr) { g_last_error = "slice cuts: tile geometry mismatch"; return BFG_ERR_INVALID; }
            int b = 0, tile0 = 0;
            for (int k = 0; k <= K; ++k) {
                const int bk = (int)((int64_t)ts.geo.nbands * k / K);
                for (; b < bk; ++b) tile0 += band_sectors_host(a->nside, ts.geo.tr, ts.geo.tw, b);
                cuts.tile[k] = tile0;
            }
            if (cuts.tile[K] != ts.geo.ntiles) { g_last_error = "slice cuts do not cover the tiles"; return BFG_ERR_INVALID; }
            cuts.n = K; cuts.range = ts.d_slices;
        }
        if (reuse) {
            ReinitParams rq;
            std::memset(&rq, 0, sizeof(rq));
            rq.geo = ts.geo; rq.hpx = pp.hpx; rq.counters = ts.d_counters; rq.n_counters = n_counters; rq.first_dynamic = 3 * tile_grid;
            rq.n_slices = cuts.n; rq.stats = c->d_stats; rq.tail = plan_tail; rq.overwrite = overwrite ? 1 : 0;
            rq.nacc = (mode == MODE_PAINT) ? 1 : 3; rq.shared_flag = ts.d_shared; rq.out = d_out;
            hipLaunchKernelGGL(plan_reinit_kernel, dim3((unsigned)std::min(std::max(ts.geo.ntiles / 8, 8), 1024)), dim3(256), 0, c->stream, rq);
            ++c->plan_reuses;
        } else {
        hipLaunchKernelGGL(tile_scan_kernel, dim3((unsigned)(1 + (ts.geo.ntiles + 1023) / 1024)), dim3(1024), 0, c->stream, ts.geo,
                           ts.cap_direct, pp.bin.tile_count, ts.d_tile_start, ts.d_work, ts.d_nwork, ts.d_counters, n_counters, 3 * tile_grid,
                           overwrite ? 1 : 0, ts.d_shared, pp.bin.tile_count + ts.geo.ntiles + 1,
                           ts.d_tile_count + (size_t)(1 - ts.flip) * (ts.geo.ntiles + kTileTail), cuts);
        c->tiles[mode].flip = 1 - ts.flip;             // the next call counts in the set this scan kernel clears
        c->tiles[mode].counting = false;
        FillParams fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.overwrite = overwrite ? 1 : 0; fp.nacc = (mode == MODE_PAINT) ? 1 : 3; fp.shared_flag = ts.d_shared; fp.out = d_out;
        fp.hpx = pp.hpx;
        fp.stats = c->d_stats; fp.n_halo = a->n_halo; fp.cap = c->cap_halo; fp.rec = c->d_rec; fp.irec = c->d_irec; fp.ht = c->d_ht; fp.bin = pp.bin; fp.prep = pp;
        hipLaunchKernelGGL(tile_fill_kernel, dim3((unsigned)((a->n_halo + 255) / 256)), dim3(256), 0, c->stream, fp);
        }
        RowParams rp;
        std::memset(&rp, 0, sizeof(rp));
        rp.n_halo = a->n_halo; rp.cap = c->cap_halo; rp.ht = c->d_ht; rp.cidx = c->d_cidx; rp.cw = c->d_cw;
        rp.tab = t->dev; rp.win_nodes = win_nodes; rp.hwin = c->d_hwin;
        if (win_table || fuse_rows || blend) {
            // no row windows / built by halo_prep_kernel / blended in the tile kernel
        } else if (win_nodes % 4 == 0 && win_nodes >= 8) {
            const int hpb = std::min(256 / (win_nodes / 4), 64);
            const size_t rlds = (size_t)64 * ((size_t)1 << t->dev.nouter) * 16;           // weights + offsets of 64 halos
            hipLaunchKernelGGL(halo_row4_kernel, dim3((unsigned)((a->n_halo + hpb - 1) / hpb)), dim3(256), rlds, c->stream, rp);
        } else {
            const int hpb = 256 / win_nodes;
            hipLaunchKernelGGL(halo_row_kernel, dim3((unsigned)((a->n_halo + hpb - 1) / hpb)), dim3(256), 0, c->stream, rp);
        }
        HIP_TRY(hipGetLastError());
        timing_end(c, 3);

        TileParams tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.hpx = pp.hpx; tp.n_halo = a->n_halo; tp.cap = c->cap_halo;
        tp.ht = c->d_ht; tp.cidx = c->d_cidx; tp.cw = c->d_cw;
        tp.tab = t->dev; tp.geo = ts.geo; tp.tile_start = ts.d_tile_start; tp.pairs = c->d_pairs;
        tp.work = ts.d_work; tp.n_work = ts.d_nwork;
        tp.hd = c->d_hd;
        tp.hwin = c->d_hwin; tp.win_nodes = win_nodes; tp.win_table = win_table ? 1 : 0; tp.pair_cap = c->pair_cap;
        tp.out = d_out; tp.stats = c->d_stats;
        tp.logtab = reinterpret_cast<const double2 *>(c->d_mathtab);
        tp.exptab = c->d_mathtab + 2 * kLogTab;
        tp.atantab = c->d_mathtab + 2 * kLogTab + kExpTab;
        { const char *dbg = std::getenv("BFG_DEBUG"); tp.debug = dbg ? std::atoi(dbg) : 0; }
        tp.out_zero = out_zero ? 1 : 0;
        tp.overwrite = overwrite ? 1 : 0;
        tp.blend = blend ? 1 : 0;
        tp.defer = (mode == MODE_PAINT) ? ts.d_defer : nullptr; tp.defer_count = ts.d_defer_count;
        tp.defer_cap_wg = (int)std::min<int64_t>(((int64_t)items_max * kDeferCap) / tile_grid, 1 << 20);   // the list, cut into one slice per workgroup
        tp.defer_tail = 1;                                                   // the workgroups add their deferred pixels themselves
        if (const char *e = std::getenv("BFG_FINAL_DRAIN")) {
            if (e[0] == 'i') tp.defer = nullptr;                            // "inline": every item drains its own queue
            if (e[0] == 'k') tp.defer_tail = 0;                             // "kernel": tile_deferred_kernel after the tile kernel
        }
        if (!c->tile_attr_set) {
            const int lp = (int)tile_lds_bytes<MODE_PAINT>() + (BFG_STAGE_TIMING == 4 ? 128 : 0), lb = (int)tile_lds_bytes<MODE_BARYONIFY>();
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_PAINT, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lp));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_PAINT, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lp));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_PAINT, true, 0, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lp));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_PAINT, true, 2, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lp));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_BARYONIFY, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(shell_tile_kernel<MODE_BARYONIFY, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lb));
            c->tile_attr_set = true;
        }
        const dim3 tgrid((unsigned)tile_grid), tblock(kTileThreads);
        tp.work_counter = persist > 0 ? ts.d_counters : nullptr;
        tp.n_counters = n_counters;
        const bool wl = win_nodes <= kWinLds;
        sp.only_flagged = 1;     // leftovers: halos the binning left to the global-atomic kernel
        sp.left = c->d_left; sp.left_n = pp.left_n;
        sp.pair_total_ptr = ts.d_tile_start + ts.geo.ntiles; sp.pair_cap = c->pair_cap;
        if (cuts.n > 0) {
            // sliced: left-overs first (into a cleared output, if there are any), then one tile launch per slice, each followed by
            // the caller's callback -- which typically starts the exchange of that part of the output on another stream
            // (only an output the call DEFINES is cleared: without BFG_SHELL_OUT_OVERWRITE the scatter kernel and the tiles add to what
            // the caller's buffer holds -- accumulate INTO -- or to the zeros the caller vouched for)
            if (overwrite) {
                hipLaunchKernelGGL(out_clear_if_left_kernel, dim3((unsigned)(4 * c->n_cu)), dim3(256), 0, c->stream, d_out, out_elems,
                                   pp.left_n, ts.d_tile_start + ts.geo.ntiles, (long long)c->pair_cap);
                HIP_TRY(hipGetLastError());
            }
            rc = launch_scatter();
            if (rc) return rc;
            tp.accum_left = pp.left_n;
        }
        const int n_launch = cuts.n > 0 ? cuts.n : 1;
        for (int islice = 0; islice < n_launch; ++islice) {
        if (cuts.n > 0) { tp.slice = cuts.range + 2 * islice; tp.work_counter = ts.d_counters + (size_t)(1 + islice) * kMaxCounters * kCounterStride; }
        timing_begin(c, 1);
        if (light && wl) {
            constexpr int ntp = TileCfg<MODE_PAINT, 1>::NT, ntb = TileCfg<MODE_BARYONIFY, 1>::NT;
            constexpr size_t ldp = tile_lds_bytes<MODE_PAINT, 1>(), ldb = tile_lds_bytes<MODE_BARYONIFY, 1>();
            if (mode == MODE_PAINT && blend && blend_rows)
                hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true, 3, true>), tgrid, dim3(ntp), ldp, c->stream, tp);
            else if (mode == MODE_PAINT && blend)
                hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true, 1, true>), tgrid, dim3(ntp), ldp, c->stream, tp);
            else if (mode == MODE_PAINT)
                hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true, 1>), tgrid, dim3(ntp), ldp, c->stream, tp);
            else
                hipLaunchKernelGGL((shell_tile_kernel<MODE_BARYONIFY, true, 1>), tgrid, dim3(ntb), ldb, c->stream, tp);
        } else if (mode == MODE_PAINT) {
            const size_t tlds = tile_lds_bytes<MODE_PAINT>() + (BFG_STAGE_TIMING == 4 ? 128 : 0);
            if (wl && blend && blend_rows) hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true, 2, true>), tgrid, tblock, tlds, c->stream, tp);
            else if (wl && blend) hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true, 0, true>), tgrid, tblock, tlds, c->stream, tp);
            else if (wl) hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, true>), tgrid, tblock, tlds, c->stream, tp);
            else hipLaunchKernelGGL((shell_tile_kernel<MODE_PAINT, false>), tgrid, tblock, tlds, c->stream, tp);
        } else {
            const size_t tlds = tile_lds_bytes<MODE_BARYONIFY>();
            if (wl) hipLaunchKernelGGL((shell_tile_kernel<MODE_BARYONIFY, true>), tgrid, tblock, tlds, c->stream, tp);
            else hipLaunchKernelGGL((shell_tile_kernel<MODE_BARYONIFY, false>), tgrid, tblock, tlds, c->stream, tp);
        }
        HIP_TRY(hipGetLastError());
        timing_end(c, 1);
        if (mode == MODE_PAINT && tp.defer && !tp.defer_tail) {
            timing_begin(c, 5);
            hipLaunchKernelGGL(tile_deferred_kernel, dim3((unsigned)std::min((tile_grid + 3) / 4, 8 * c->n_cu)), dim3(256), 0, c->stream, tp, tile_grid);
            HIP_TRY(hipGetLastError());
            timing_end(c, 5);
        }
        if (cuts.n > 0 && slice_fn(slice_user, islice, cuts.n, slice_elem[islice], slice_elem[islice + 1]) != 0) {
            g_last_error = "the slice callback failed";
            return BFG_ERR_INVALID;
        }
        }   // slices
        // what this call leaves behind is a plan (its own, or the one it ran on)
        c->plan.key = key; c->plan.tail = const_cast<int32_t *>(pp.left_n); c->plan.valid = (t->dev.hstride == 0);
        if (cuts.n > 0) return BFG_OK;
    }
    rc = launch_scatter();
    if (rc) { c->plan.valid = false; return rc; }
    return whole_output();
}

int bfg_paint_shell(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s, double *d_map)
{
    return run_shell(c, a, t, s, d_map, MODE_PAINT);
}

int bfg_baryonify_offsets(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s,
                          double *d_offsets)
{
    return run_shell(c, a, t, s, d_offsets, MODE_BARYONIFY);
}

int bfg_paint_shell_sliced(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s, double *d_map,
                           int n_slices, bfg_slice_fn fn, void *user)
{
    if (n_slices < 1 || !fn) return BFG_ERR_INVALID;
    return run_shell(c, a, t, s, d_map, MODE_PAINT, n_slices, fn, user);
}

int bfg_baryonify_offsets_sliced(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s,
                                 double *d_offsets, int n_slices, bfg_slice_fn fn, void *user)
{
    if (n_slices < 1 || !fn) return BFG_ERR_INVALID;
    return run_shell(c, a, t, s, d_offsets, MODE_BARYONIFY, n_slices, fn, user);
}

// Device -> page-locked host memory by a copy KERNEL (stores over PCIe) instead of the DMA engine: on this platform a
// host -> device and a device -> host DMA copy on two streams take turns (1.8 + 1.8 ms for two 101 MB maps), while a kernel's
// stores to mapped host memory overlap an incoming DMA copy.  32 workgroups: enough to fill the link, few enough to leave the
// CUs to the kernels it runs beside.
typedef double copy_v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void copy_to_mapped_kernel(copy_v2d *__restrict__ dst, const copy_v2d *__restrict__ src, int64_t n2,
                                                             double *__restrict__ dst1, const double *__restrict__ src1, int tail)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
        if (tail & 2) dst[i] = src[i]; else __builtin_nontemporal_store(src[i], dst + i);
    if ((tail & 1) && blockIdx.x == 0 && threadIdx.x == 0) *dst1 = *src1;
}

int bfg_copy_to_mapped_host(bfg_ctx *c, void *stream, void *host_dst, const void *d_src, size_t bytes)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (bytes == 0) return BFG_OK;
    if (!host_dst || !d_src || (bytes % sizeof(double)) || ((uintptr_t)host_dst % 16) || ((uintptr_t)d_src % 16)) return BFG_ERR_INVALID;
    void *dev_view = nullptr;
    if (hipHostGetDevicePointer(&dev_view, host_dst, 0) != hipSuccess || !dev_view) {
        (void)hipGetLastError();
        g_last_error = "bfg_copy_to_mapped_host: the destination is not page-locked (mapped) host memory";
        return BFG_ERR_INVALID;
    }
    const int64_t n = (int64_t)(bytes / sizeof(double)), n2 = n / 2;
    int cgrid = 32;
    if (const char *e = std::getenv("BFG_COPY_GRID")) cgrid = std::max(1, std::atoi(e));
    // (an explicit stream: re-binding the context to the copy stream and back would order the next kernels behind the copy)
    hipLaunchKernelGGL(copy_to_mapped_kernel, dim3((unsigned)cgrid), dim3(256), 0, stream ? (hipStream_t)stream : c->stream, reinterpret_cast<copy_v2d *>(dev_view),
                       reinterpret_cast<const copy_v2d *>(d_src), n2, reinterpret_cast<double *>(dev_view) + 2 * n2,
                       reinterpret_cast<const double *>(d_src) + 2 * n2, (int)(n & 1) | (std::getenv("BFG_COPY_PLAIN") ? 2 : 0));
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

// ---- models that are not tabulated: the geometry around a host-evaluated .projected / .displacement (bfg_enum.hpp) ----
static int enum_prep(const bfg_shell_args *a, const bfg_spline *s, PrepParams &pp)
{
    if (!a || !s) return BFG_ERR_INVALID;
    if (a->nside < 1 || a->nside > (1 << 20) || a->n_halo < 0 || (a->n_halo > 0 && !a->d_catalog)) return BFG_ERR_INVALID;
    if (a->cat_stride < 4 || !(a->epsilon_max >= 0)) return BFG_ERR_INVALID;
    if (a->n_halo > 0x7fffffffLL) return BFG_ERR_UNSUPPORTED;            // the entry lists carry 32-bit halo indices
    std::memset(&pp, 0, sizeof(pp));
    pp.hpx = make_hpx(a->nside);
    pp.n_halo = a->n_halo; pp.cap = a->n_halo;
    pp.cat = a->d_catalog; pp.cat_stride = a->cat_stride;
    pp.eps_run = a->epsilon_max;
    pp.md_run = a->runner_md; pp.md_model = a->runner_md;
    pp.spl_n = s->n; pp.spl_knots = s->d_knots; pp.spl_coef = s->d_coef;
    return BFG_OK;                                                       // tab.nouter = 0: no table, no cell search
}

int bfg_disc_enumerate_count(bfg_ctx *c, const bfg_shell_args *a, const bfg_spline *s, int fallback4, int64_t *d_counts)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    bfg::EnumParams ep;
    rc = enum_prep(a, s, ep.prep);
    if (rc) return rc;
    if (a->n_halo == 0) return BFG_OK;
    if (!d_counts) return BFG_ERR_INVALID;
    ep.fallback4 = fallback4 ? 1 : 0; ep.counts = d_counts; ep.base = nullptr; ep.pix = nullptr; ep.r_com = nullptr; ep.halo = nullptr;
    const unsigned grid = (unsigned)std::min<int64_t>((a->n_halo + 3) / 4, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL((bfg::disc_enum_kernel<true>), dim3(grid), dim3(256), 0, c->stream, ep);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_disc_enumerate(bfg_ctx *c, const bfg_shell_args *a, const bfg_spline *s, int fallback4, const int64_t *d_base,
                       int64_t *d_pix, double *d_r_com, int32_t *d_halo)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    bfg::EnumParams ep;
    rc = enum_prep(a, s, ep.prep);
    if (rc) return rc;
    if (a->n_halo == 0) return BFG_OK;
    if (!d_base || !d_pix || !d_r_com || !d_halo) return BFG_ERR_INVALID;
    ep.fallback4 = fallback4 ? 1 : 0; ep.counts = nullptr; ep.base = d_base; ep.pix = d_pix; ep.r_com = d_r_com; ep.halo = d_halo;
    const unsigned grid = (unsigned)std::min<int64_t>((a->n_halo + 3) / 4, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL((bfg::disc_enum_kernel<false>), dim3(grid), dim3(256), 0, c->stream, ep);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_map_add_values(bfg_ctx *c, double *d_map, const int64_t *d_pix, const double *d_val, int64_t n)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_map || !d_pix || !d_val))) return BFG_ERR_INVALID;
    if (n == 0) return BFG_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL(bfg::values_add_kernel, dim3(grid), dim3(256), 0, c->stream, d_map, d_pix, d_val, n);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_offsets_add_displacements(bfg_ctx *c, const bfg_shell_args *a, const bfg_spline *s, const int64_t *d_pix,
                                  const int32_t *d_halo, const double *d_disp, int64_t n, double *d_offsets)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    bfg::DispParams dp;
    rc = enum_prep(a, s, dp.prep);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!d_pix || !d_halo || !d_disp || !d_offsets))) return BFG_ERR_INVALID;
    if (n == 0) return BFG_OK;
    dp.pix = d_pix; dp.halo = d_halo; dp.disp = d_disp; dp.n = n; dp.out = d_offsets;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL(bfg::displacements_add_kernel, dim3(grid), dim3(256), 0, c->stream, dp);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

// the list of pixels the tile kernel leaves to regrid_list_kernel: one slot per pixel (every pixel could be one), grow-only;
// 4-byte entries while pixel indices fit (rg_slow_cap counts BYTES)
static inline bool regrid_idx32(int64_t npix) { return npix < ((int64_t)1 << 31); }
static int ensure_regrid_list(bfg_ctx *c, int64_t npix)
{
    if (!c->d_rg_slow_n) HIP_TRY(hipMalloc((void **)&c->d_rg_slow_n, sizeof(unsigned long long)));
    const int64_t want = npix * (int64_t)(regrid_idx32(npix) ? sizeof(int32_t) : sizeof(int64_t));
    if (want > c->rg_slow_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_rg_slow) (void)hipFree(c->d_rg_slow);
        c->d_rg_slow = nullptr; c->rg_slow_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_rg_slow, (size_t)want));
        c->rg_slow_cap = want;
    }
    return BFG_OK;
}

static int regrid_debug_mode()
{
    // A/B: BFG_REGRID=full: no exit for undisplaced pixels; =general: no differential path (every displaced pixel through the list
    // kernel); =all: neither
    const char *rg_env = std::getenv("BFG_REGRID");
    return !rg_env ? 0 : (rg_env[0] == 'f' ? 1 : (rg_env[0] == 'g' ? 2 : (rg_env[0] == 'a' ? 3 : 0)));
}

int bfg_regrid_shell(bfg_ctx *c, int64_t nside, const double *d_offsets, const double *d_in_map,
                     double *d_out_map, double *d_sums)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (nside < 1 || nside > (1 << 20) || !d_offsets || !d_in_map || !d_out_map) return BFG_ERR_INVALID;
    Hpx hp = make_hpx(nside);
    const char *rg_env = std::getenv("BFG_REGRID");                 // "pixel" forces the one-thread-per-pixel kernel
    const bool use_tiles = nside >= 8 && !(rg_env && rg_env[0] == 'p');
    if (use_tiles) {
        rc = ensure_tiles(c, kRegridSet, kRegridTR, kTileWidth, nside, 0);
        if (rc) return rc;
    }
    if (d_sums) HIP_TRY(hipMemsetAsync(d_sums, 0, 2 * sizeof(double), c->stream));
    if (use_tiles) {
        rc = ensure_regrid_list(c, hp.npix);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(c->d_rg_slow_n, 0, sizeof(unsigned long long), c->stream));
    }
    timing_begin(c, 2);
    if (use_tiles) {
        const dim3 tg((unsigned)c->tiles[kRegridSet].geo.ntiles), lg((unsigned)(2 * c->n_cu));
        if (regrid_idx32(hp.npix)) {
            hipLaunchKernelGGL(regrid_tile_kernel<int32_t>, tg, dim3(256), 0, c->stream, hp, c->tiles[kRegridSet].geo, d_offsets, d_in_map,
                               d_out_map, d_sums, regrid_debug_mode(), 0, (int32_t *)c->d_rg_slow, c->d_rg_slow_n);
            hipLaunchKernelGGL(regrid_list_kernel<int32_t>, lg, dim3(256), 0, c->stream, hp, (const int32_t *)c->d_rg_slow, c->d_rg_slow_n,
                               d_offsets, d_in_map, d_out_map, d_sums, 0, (double *)nullptr);
        } else {
            hipLaunchKernelGGL(regrid_tile_kernel<int64_t>, tg, dim3(256), 0, c->stream, hp, c->tiles[kRegridSet].geo, d_offsets, d_in_map,
                               d_out_map, d_sums, regrid_debug_mode(), 0, (int64_t *)c->d_rg_slow, c->d_rg_slow_n);
            hipLaunchKernelGGL(regrid_list_kernel<int64_t>, lg, dim3(256), 0, c->stream, hp, (const int64_t *)c->d_rg_slow, c->d_rg_slow_n,
                               d_offsets, d_in_map, d_out_map, d_sums, 0, (double *)nullptr);
        }
    } else
        hipLaunchKernelGGL(regrid_kernel, dim3((unsigned)((hp.npix + 255) / 256)), dim3(256), 0, c->stream, hp,
                           d_offsets, d_in_map, d_out_map, d_sums);
    HIP_TRY(hipGetLastError());
    timing_end(c, 2);
    return BFG_OK;
}

// The regrid of the SOURCE pixels of ring bands [band_lo, band_hi) only (bands of bfg_regrid_band_rings() rings, counted from the
// north pole): what lets a caller regrid a map slice by slice while the rest of it is still arriving over PCIe, and send a
// finished slice back while the next one is regridded (Runners/HealpixRunner.py:357-365 on a band of source pixels).
int bfg_regrid_band_rings(void) { return kRegridTR; }

int bfg_regrid_shell_bands(bfg_ctx *c, int64_t nside, const double *d_offsets, const double *d_in_map, double *d_out_map,
                           double *d_sums3, int band_lo, int band_hi, uint32_t flags)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (nside < 1 || nside > (1 << 20) || !d_offsets || !d_in_map || !d_out_map) return BFG_ERR_INVALID;
    if (nside < 8) return BFG_ERR_UNSUPPORTED;                       // (no tile geometry below NSIDE 8: use bfg_regrid_shell)
    const int nbands = (int)((4 * nside - 1 + kRegridTR - 1) / kRegridTR);
    if (band_lo < 0 || band_hi > nbands || band_lo > band_hi) return BFG_ERR_INVALID;
    rc = ensure_tiles(c, kRegridSet, kRegridTR, kTileWidth, nside, 0);
    if (rc) return rc;
    if (d_sums3 && (flags & 1u)) HIP_TRY(hipMemsetAsync(d_sums3, 0, 3 * sizeof(double), c->stream));
    if (band_lo == band_hi) return BFG_OK;
    int tile_lo = 0, tile_hi = 0;
    for (int b = 0; b < band_hi; ++b) {
        const int ns = band_sectors_host(nside, kRegridTR, kTileWidth, b);
        if (b < band_lo) tile_lo += ns;
        tile_hi += ns;
    }
    const Hpx hp = make_hpx(nside);
    rc = ensure_regrid_list(c, hp.npix);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_rg_slow_n, 0, sizeof(unsigned long long), c->stream));
    timing_begin(c, 2);
    const dim3 tg((unsigned)(tile_hi - tile_lo)), lg((unsigned)(2 * c->n_cu));
    double *const far = d_sums3 ? d_sums3 + 2 : (double *)nullptr;
    if (regrid_idx32(hp.npix)) {
        hipLaunchKernelGGL(regrid_tile_kernel<int32_t>, tg, dim3(256), 0, c->stream, hp, c->tiles[kRegridSet].geo, d_offsets, d_in_map,
                           d_out_map, d_sums3, regrid_debug_mode(), tile_lo, (int32_t *)c->d_rg_slow, c->d_rg_slow_n);
        hipLaunchKernelGGL(regrid_list_kernel<int32_t>, lg, dim3(256), 0, c->stream, hp, (const int32_t *)c->d_rg_slow, c->d_rg_slow_n,
                           d_offsets, d_in_map, d_out_map, d_sums3, kRegridTR, far);
    } else {
        hipLaunchKernelGGL(regrid_tile_kernel<int64_t>, tg, dim3(256), 0, c->stream, hp, c->tiles[kRegridSet].geo, d_offsets, d_in_map,
                           d_out_map, d_sums3, regrid_debug_mode(), tile_lo, (int64_t *)c->d_rg_slow, c->d_rg_slow_n);
        hipLaunchKernelGGL(regrid_list_kernel<int64_t>, lg, dim3(256), 0, c->stream, hp, (const int64_t *)c->d_rg_slow, c->d_rg_slow_n,
                           d_offsets, d_in_map, d_out_map, d_sums3, kRegridTR, far);
    }
    HIP_TRY(hipGetLastError());
    timing_end(c, 2);
    return BFG_OK;
}

int bfg_reduce_absmax_sum(bfg_ctx *c, int64_t n, const double *d_x, double *absmax, double *sum)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !d_x)) return BFG_ERR_INVALID;
    double h[2] = {0.0, 0.0};
    if (n > 0) {
        HIP_TRY(hipMemsetAsync(c->d_red, 0, 2 * sizeof(double), c->stream));
        unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(reduce_absmax_sum_kernel, dim3(grid), dim3(256), 0, c->stream, n, d_x, c->d_red);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, c->d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (sum) *sum = h[0];
    if (absmax) *absmax = h[1];
    return BFG_OK;
}

} // extern C (helper below has C++ linkage)
// grow-only workspace of the particle-grouping passes (tiled deposit, cell-grouped snapshot pass)
static int dep_workspace(bfg_ctx *c, const size_t want[5])
{
    for (int k = 0; k < 5; ++k) {
        if (want[k] > c->dep_cap[k]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->dep_buf[k]) (void)hipFree(c->dep_buf[k]);
            c->dep_buf[k] = nullptr; c->dep_cap[k] = 0;
            HIP_TRY(hipMalloc(&c->dep_buf[k], want[k]));
            c->dep_cap[k] = want[k];
        }
    }
    return BFG_OK;
}

extern "C" {
int bfg_baryonify_snapshot(bfg_ctx *c, const bfg_snapshot_args *a, const bfg_table *t, double *d_out)
{
    return bfg_baryonify_snapshot_strided(c, a, t, d_out, a ? a->ndim : 0, a ? a->ndim : 0);
}

int bfg_baryonify_snapshot_strided(bfg_ctx *c, const bfg_snapshot_args *a, const bfg_table *t, double *d_out,
                                   int64_t part_stride, int64_t out_stride)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!a || !t || (!d_out && a->n_part > 0) || (a->ndim != 2 && a->ndim != 3) || a->n_part < 0 || a->n_halo < 0 || !(a->L > 0) ||
        !(a->a > 0) || a->n_part >= (1ll << 31) || a->n_extra < 0 || a->halo_stride < 5 + a->n_extra)
        return BFG_ERR_INVALID;
    if (!t->d_blob) return BFG_ERR_UNSUPPORTED;     // (more p_keys axes than the kernels read: shell runners only)
    if (t->dev.ndim != 3 + a->n_extra || t->dev.log_values) return BFG_ERR_INVALID;       // linear displacement table
    if ((a->n_part > 0 && !a->d_part) || (a->n_halo > 0 && !a->d_halo)) return BFG_ERR_INVALID;
    if (a->n_part == 0) return BFG_OK;
    SnapParams P;
    std::memset(&P, 0, sizeof(P));
    P.ndim = a->ndim; P.rdelta = a->rdelta_sampling; P.n_part = a->n_part; P.n_halo = a->n_halo;
    P.L = a->L; P.a = a->a; P.eps_run = a->epsilon_max; P.eps_model = a->model_epsilon_max;
    P.md_run = a->runner_md; P.md_model = a->model_md;
    if (part_stride < a->ndim || out_stride < a->ndim) return BFG_ERR_INVALID;
    P.part = a->d_part; P.pstride = part_stride; P.ostride = out_stride; P.halo = a->d_halo; P.halo_stride = a->halo_stride; P.n_extra = a->n_extra;
    P.tab = t->dev; P.stats = c->d_stats; P.out = d_out;
    P.logtab = reinterpret_cast<const double2 *>(c->d_mathtab);
    // coarse cell grid for the halo-overlap lists: a few candidate halos per cell
    const int nmax = (a->ndim == 3) ? 128 : 2048;
    // ~8 cells per halo (4 until round 6: 2.1 of a particle's ~8 candidates were hits; with cells of 11 instead of 14 Mpc at BASELINE
    // configs[4] it tests ~6: particle kernel 4.70 -> 4.53 ms, 21.8 -> 20.2 ms with the particles in random order; 16: 4.47 ms, but the
    // overlap lists then cost what the kernel gains -- profiles/r06_snapshot_ab.txt)
    double cell_factor = 8.0;
    if (const char *e = std::getenv("BFG_SNAP_CELL_FACTOR")) cell_factor = std::max(0.25, std::atof(e));      // A/B switch
    int ncell = (int)std::floor(std::pow(cell_factor * (double)std::max<int64_t>(a->n_halo, 1), 1.0 / a->ndim));
    ncell = std::max(4, std::min(ncell, nmax));
    P.ncell = ncell;
    P.ncell_tot = (a->ndim == 3) ? (int64_t)ncell * ncell * ncell : (int64_t)ncell * ncell;
    const int64_t nblk = (P.ncell_tot + 1023) / 1024;
    // workspace kept in the context between calls; the candidate list is sized after the count pass
    const size_t want[8] = {(size_t)P.ncell_tot * sizeof(int32_t), (size_t)(P.ncell_tot + 1) * sizeof(int32_t),
                            (size_t)(std::max<int64_t>(a->n_halo, 1) + 1) * sizeof(int32_t),
                            (size_t)std::max<int64_t>(a->n_halo, 1) * sizeof(SnapHalo),
                            (size_t)std::max<int64_t>(a->n_halo, 1) * t->dev.nr * sizeof(double),
                            c->snap_cap[5], (size_t)nblk * sizeof(int32_t), sizeof(int32_t)};
    auto ensure = [&](int k, size_t bytes) -> int {
        if (bytes > c->snap_cap[k]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->snap_buf[k]) (void)hipFree(c->snap_buf[k]);
            c->snap_buf[k] = nullptr; c->snap_cap[k] = 0;
            HIP_TRY(hipMalloc(&c->snap_buf[k], bytes));
            c->snap_cap[k] = bytes;
        }
        return BFG_OK;
    };
    for (int k = 0; k < 8; ++k) { rc = ensure(k, want[k]); if (rc) return rc; }
    P.cell_count = (int32_t *)c->snap_buf[0]; P.cell_start = (int32_t *)c->snap_buf[1]; P.big = (int32_t *)c->snap_buf[2];
    P.hs = (SnapHalo *)c->snap_buf[3]; P.hrow = (double *)c->snap_buf[4];
    int32_t *d_bsum = (int32_t *)c->snap_buf[6], *d_total = (int32_t *)c->snap_buf[7];
    HIP_TRY(hipMemsetAsync(P.cell_count, 0, (size_t)P.ncell_tot * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(P.big, 0, sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(P.cell_start, 0, (size_t)(P.ncell_tot + 1) * sizeof(int32_t), c->stream));
    if (a->n_halo > 0) {
        const unsigned hgrid = (unsigned)((a->n_halo * kSnapOverlapLanes + 255) / 256);
        hipLaunchKernelGGL(snap_halo_kernel, dim3((unsigned)a->n_halo), dim3(64), 0, c->stream, P);
        if (a->ndim == 3) hipLaunchKernelGGL(snap_overlap_kernel<3>, dim3(hgrid), dim3(256), 0, c->stream, P, 0);
        else hipLaunchKernelGGL(snap_overlap_kernel<2>, dim3(hgrid), dim3(256), 0, c->stream, P, 0);
        hipLaunchKernelGGL(snap_scan_block_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, P.ncell_tot, P.cell_count,
                           P.cell_start, d_bsum);
        hipLaunchKernelGGL(snap_scan_sums_kernel, dim3(1), dim3(1024), 0, c->stream, (int)nblk, d_bsum, d_total);
        hipLaunchKernelGGL(snap_scan_add_kernel, dim3((unsigned)((P.ncell_tot + 255) / 256)), dim3(256), 0, c->stream,
                           P.ncell_tot, P.cell_start, d_bsum, P.cell_count, d_total);
        int32_t total = 0;                                      // size of the candidate list (a few entries per halo)
        HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        rc = ensure(5, (size_t)std::max<int32_t>(total, 1) * sizeof(SnapCand));
        if (rc) return rc;
        P.cand = (SnapCand *)c->snap_buf[5]; P.cand_cap = total;
        if (a->ndim == 3) hipLaunchKernelGGL(snap_overlap_kernel<3>, dim3(hgrid), dim3(256), 0, c->stream, P, 1);
        else hipLaunchKernelGGL(snap_overlap_kernel<2>, dim3(hgrid), dim3(256), 0, c->stream, P, 1);
    } else {
        P.cand = (SnapCand *)c->snap_buf[5]; P.cand_cap = 0;
    }
    // default: one thread per particle in the caller's order.  BFG_SNAPSHOT=cell: group the particle indices by cell,
    // then one wavefront per cell with wave-uniform candidate lists (measured equal at 512^3: 7.1 vs 6.2 ms; kept as the
    // variant for particle orders without any spatial coherence)
    bool grouped = false;
    if (const char *e = std::getenv("BFG_SNAPSHOT"))
        if (!std::strcmp(e, "cell")) grouped = a->n_part < (int64_t)0x7fffffff;
    if (!grouped) {
        const unsigned pgrid = (unsigned)std::min<int64_t>(((a->n_part + 255) / 256 + 7) / 8 * 8, 8192);       // grid-stride; a multiple of 8
        P.xcd_map = 1;
        if (const char *e = std::getenv("BFG_SNAP_XCD")) P.xcd_map = std::atoi(e) != 0;                 // A/B switch
        // (round 6, measured and not kept -- profiles/r06_snapshot_ab.txt, commits aa966fe / d66acd4: the hits of a wavefront spread
        // over its lanes through an LDS queue, 4.80 -> 4.73 ms; the candidate lists staged in LDS run by run, 4.75 -> 5.70 ms)
        timing_begin(c, 6);
        if (a->ndim == 3) hipLaunchKernelGGL(snap_particle_kernel<3>, dim3(pgrid), dim3(256), 0, c->stream, P);
        else hipLaunchKernelGGL(snap_particle_kernel<2>, dim3(pgrid), dim3(256), 0, c->stream, P);
        HIP_TRY(hipGetLastError());
        timing_end(c, 6);
        return BFG_OK;
    }
    const size_t dwant[5] = {(size_t)a->n_part * sizeof(int32_t), (size_t)a->n_part * sizeof(int32_t),
                             (size_t)P.ncell_tot * sizeof(int32_t), (size_t)(P.ncell_tot + 1) * sizeof(int32_t),
                             (size_t)(nblk + 1) * sizeof(int32_t)};
    rc = dep_workspace(c, dwant);
    if (rc) return rc;
    P.pkey = (int32_t *)c->dep_buf[0]; P.perm = (int32_t *)c->dep_buf[1]; P.pcount = (int32_t *)c->dep_buf[2];
    int32_t *d_pstart = (int32_t *)c->dep_buf[3], *d_pbsum = (int32_t *)c->dep_buf[4], *d_ptotal = d_pbsum + nblk;
    P.pstart = d_pstart;
    HIP_TRY(hipMemsetAsync(P.pcount, 0, (size_t)P.ncell_tot * sizeof(int32_t), c->stream));
    const unsigned kgrid = (unsigned)std::min<int64_t>((a->n_part + 255) / 256, 16384);         // grid-stride
    if (a->ndim == 3) hipLaunchKernelGGL(snap_key_kernel<3>, dim3(kgrid), dim3(256), 0, c->stream, P);
    else hipLaunchKernelGGL(snap_key_kernel<2>, dim3(kgrid), dim3(256), 0, c->stream, P);
    hipLaunchKernelGGL(snap_scan_block_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, P.ncell_tot, P.pcount, d_pstart,
                       d_pbsum);
    hipLaunchKernelGGL(snap_scan_sums_kernel, dim3(1), dim3(1024), 0, c->stream, (int)nblk, d_pbsum, d_ptotal);
    hipLaunchKernelGGL(snap_scan_add_kernel, dim3((unsigned)((P.ncell_tot + 255) / 256)), dim3(256), 0, c->stream, P.ncell_tot,
                       d_pstart, d_pbsum, P.pcount, d_ptotal);
    hipLaunchKernelGGL(group_fill_kernel, dim3(kgrid), dim3(256), 0, c->stream, a->n_part, P.pkey, P.pcount, P.pstart, P.perm);
    const unsigned cgrid = (unsigned)std::min<int64_t>((P.ncell_tot + 3) / 4, 8192);             // 4 cells per workgroup trip
    if (a->ndim == 3) hipLaunchKernelGGL(snap_cell_kernel<3>, dim3(cgrid), dim3(256), 0, c->stream, P);
    else hipLaunchKernelGGL(snap_cell_kernel<2>, dim3(cgrid), dim3(256), 0, c->stream, P);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

static int run_grid(bfg_ctx *c, const bfg_grid_args *a, const bfg_table *t, double *d_out, int mode)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!a || !t || !d_out || (a->ndim != 2 && a->ndim != 3) || a->n_halo < 0 || a->npix < 4 || !(a->a > 0) ||
        a->n_extra < 0 || a->halo_stride < 5 + a->n_extra || !a->d_bins)
        return BFG_ERR_INVALID;
    if (!t->d_blob) return BFG_ERR_UNSUPPORTED;     // (more p_keys axes than the kernels read: shell runners only)
    if (t->dev.ndim != 3 + a->n_extra) return BFG_ERR_INVALID;
    if ((mode == MODE_PAINT) != (t->dev.log_values != 0)) return BFG_ERR_INVALID;          // paint: ln T; baryonify: linear d
    if (a->n_halo == 0) return BFG_OK;
    if (!a->d_halo) return BFG_ERR_INVALID;
    const size_t want[2] = {(size_t)a->n_halo * sizeof(GridHalo), (size_t)a->n_halo * t->dev.nr * sizeof(double)};
    for (int k = 0; k < 2; ++k) {
        if (want[k] > c->grid_cap[k]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->grid_buf[k]) (void)hipFree(c->grid_buf[k]);
            c->grid_buf[k] = nullptr; c->grid_cap[k] = 0;
            HIP_TRY(hipMalloc(&c->grid_buf[k], want[k]));
            c->grid_cap[k] = want[k];
        }
    }
    double b01[2];
    HIP_TRY(hipMemcpyAsync(b01, a->d_bins, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    GridParams P;
    std::memset(&P, 0, sizeof(P));
    P.ndim = a->ndim; P.npix = a->npix; P.mode = mode; P.rdelta = a->rdelta_sampling; P.n_halo = a->n_halo;
    P.res = b01[1] - b01[0];                                                               // GriddedMap.res (io.py:455)
    if (!(P.res > 0)) return BFG_ERR_INVALID;
    P.a = a->a; P.eps_run = a->epsilon_max; P.eps_model = a->model_epsilon_max;
    P.md_run = a->runner_md; P.md_model = a->model_md;
    P.bins = a->d_bins; P.halo = a->d_halo; P.halo_stride = a->halo_stride; P.n_extra = a->n_extra;
    P.rmat = (a->ndim == 2) ? a->d_rmat : nullptr;
    P.tab = t->dev; P.gh = (GridHalo *)c->grid_buf[0]; P.hrow = (double *)c->grid_buf[1]; P.out = d_out; P.stats = c->d_stats;
    hipLaunchKernelGGL(grid_halo_kernel, dim3((unsigned)a->n_halo), dim3(64), 0, c->stream, P);
    // tile-privatised pass when the map has at least two tiles per axis (BFG_GRID=direct forces the atomic kernel,
    // BFG_GRID=small the small-tile instantiation used by the tests on the golden fixtures' small maps)
    bool small = false;
    bool direct = false;
    if (const char *e = std::getenv("BFG_GRID")) { direct = !std::strcmp(e, "direct"); small = !std::strcmp(e, "small"); }
    const int TS = (a->ndim == 2) ? (small ? GridTile<2>::TS_SMALL : GridTile<2>::TS) : (small ? GridTile<3>::TS_SMALL : GridTile<3>::TS);
    bool tiled = !direct && a->npix >= 2 * TS && a->n_halo < (int64_t)0x7fffffff;
    auto launch_direct = [&]() -> int {
        const dim3 g((unsigned)a->n_halo), b(256);
        if (mode == MODE_PAINT) {
            if (a->ndim == 2) hipLaunchKernelGGL((grid_window_kernel<2, MODE_PAINT>), g, b, 0, c->stream, P);
            else hipLaunchKernelGGL((grid_window_kernel<3, MODE_PAINT>), g, b, 0, c->stream, P);
        } else {
            if (a->ndim == 2) hipLaunchKernelGGL((grid_window_kernel<2, MODE_BARYONIFY>), g, b, 0, c->stream, P);
            else hipLaunchKernelGGL((grid_window_kernel<3, MODE_BARYONIFY>), g, b, 0, c->stream, P);
        }
        HIP_TRY(hipGetLastError());
        return BFG_OK;
    };
    if (!tiled) return launch_direct();
    GridBin B;
    std::memset(&B, 0, sizeof(B));
    B.nt = (a->npix + TS - 1) / TS;
    const int64_t ntile = (a->ndim == 2) ? (int64_t)B.nt * B.nt : (int64_t)B.nt * B.nt * B.nt;
    const int64_t nblk = (ntile + 1023) / 1024;
    auto ensure = [&](int k, size_t bytes) -> int {
        if (bytes > c->grid_cap[k]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->grid_buf[k]) (void)hipFree(c->grid_buf[k]);
            c->grid_buf[k] = nullptr; c->grid_cap[k] = 0;
            HIP_TRY(hipMalloc(&c->grid_buf[k], bytes));
            c->grid_cap[k] = bytes;
        }
        return BFG_OK;
    };
    if ((rc = ensure(2, (size_t)ntile * sizeof(int32_t)))) return rc;
    if ((rc = ensure(3, (size_t)(ntile + 1) * sizeof(int32_t)))) return rc;
    if ((rc = ensure(5, (size_t)(nblk + 1) * sizeof(int32_t)))) return rc;
    B.count = (int32_t *)c->grid_buf[2];
    int32_t *d_start = (int32_t *)c->grid_buf[3], *d_bsum = (int32_t *)c->grid_buf[5], *d_total = d_bsum + nblk;
    B.start = d_start;
    B.logtab = reinterpret_cast<const double2 *>(c->d_mathtab);
    B.exptab = c->d_mathtab + 2 * kLogTab;
    HIP_TRY(hipMemsetAsync(B.count, 0, (size_t)ntile * sizeof(int32_t), c->stream));
    B.winpix = c->d_pair_total;
    HIP_TRY(hipMemsetAsync(B.winpix, 0, sizeof(unsigned long long), c->stream));
    const unsigned hgrid = (unsigned)((a->n_halo + 255) / 256);
    auto launch_bin = [&]() {
        if (a->ndim == 2 && !small) hipLaunchKernelGGL((grid_bin_kernel<2, GridTile<2>::TS>), dim3(hgrid), dim3(256), 0, c->stream, P, B);
        else if (a->ndim == 2) hipLaunchKernelGGL((grid_bin_kernel<2, GridTile<2>::TS_SMALL>), dim3(hgrid), dim3(256), 0, c->stream, P, B);
        else if (!small) hipLaunchKernelGGL((grid_bin_kernel<3, GridTile<3>::TS>), dim3(hgrid), dim3(256), 0, c->stream, P, B);
        else hipLaunchKernelGGL((grid_bin_kernel<3, GridTile<3>::TS_SMALL>), dim3(hgrid), dim3(256), 0, c->stream, P, B);
    };
    B.fill = 0;
    launch_bin();
    hipLaunchKernelGGL(snap_scan_block_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, ntile, B.count, d_start, d_bsum);
    hipLaunchKernelGGL(snap_scan_sums_kernel, dim3(1), dim3(1024), 0, c->stream, (int)nblk, d_bsum, d_total);
    hipLaunchKernelGGL(snap_scan_add_kernel, dim3((unsigned)((ntile + 255) / 256)), dim3(256), 0, c->stream, ntile, d_start,
                       d_bsum, B.count, d_total);
    int32_t total = 0;                                            // (halo, tile) pairs: a few per halo
    unsigned long long winpix = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&winpix, B.winpix, sizeof(winpix), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // every touched tile costs an LDS clear and a sweep of its pixels, so tiles only pay when the windows cover the map
    // a few times over (2D 4096^2, 1e5 halos: 24x -> 2.1x / 3.4x faster; 3D 512^3, 2e4 halos: 0.08x -> 1.4x / 2.6x slower)
    const double npix_tot = std::pow((double)a->npix, a->ndim);
    if (!small && !std::getenv("BFG_GRID") && (double)winpix < 2.0 * npix_tot) return launch_direct();
    if ((rc = ensure(4, (size_t)std::max<int32_t>(total, 1) * sizeof(int32_t)))) return rc;
    B.pairs = (int32_t *)c->grid_buf[4]; B.pair_cap = total;
    B.fill = 1;
    launch_bin();
    const int na = (mode == MODE_PAINT) ? 1 : a->ndim;
    const int nc = (a->ndim == 2) ? TS * TS : TS * TS * TS;
    const size_t lds = (size_t)nc * na * sizeof(double) + kLogTab * sizeof(double2) + kExpTab * sizeof(double);
    if (!c->grid_attr_set) {
        const int big = 128 * 1024;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(grid_tile_kernel<2, MODE_BARYONIFY, GridTile<2>::TS>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(grid_tile_kernel<3, MODE_BARYONIFY, GridTile<3>::TS>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
        c->grid_attr_set = true;
    }
    const dim3 g((unsigned)ntile), b(256);
#define BFG_GRID_LAUNCH(ND, MD)                                                                                          \
    do {                                                                                                                  \
        if (!small) hipLaunchKernelGGL((grid_tile_kernel<ND, MD, GridTile<ND>::TS>), g, b, lds, c->stream, P, B);         \
        else hipLaunchKernelGGL((grid_tile_kernel<ND, MD, GridTile<ND>::TS_SMALL>), g, b, lds, c->stream, P, B);          \
    } while (0)
    if (mode == MODE_PAINT) {
        if (a->ndim == 2) BFG_GRID_LAUNCH(2, MODE_PAINT); else BFG_GRID_LAUNCH(3, MODE_PAINT);
    } else {
        if (a->ndim == 2) BFG_GRID_LAUNCH(2, MODE_BARYONIFY); else BFG_GRID_LAUNCH(3, MODE_BARYONIFY);
    }
#undef BFG_GRID_LAUNCH
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_paint_grid(bfg_ctx *c, const bfg_grid_args *a, const bfg_table *t, double *d_map)
{
    return run_grid(c, a, t, d_map, MODE_PAINT);
}

int bfg_baryonify_grid_offsets(bfg_ctx *c, const bfg_grid_args *a, const bfg_table *t, double *d_offsets)
{
    return run_grid(c, a, t, d_offsets, MODE_BARYONIFY);
}

int bfg_regrid_grid(bfg_ctx *c, int ndim, int npix, const double *d_offsets, const double *d_in_map, double *d_out_map)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if ((ndim != 2 && ndim != 3) || npix < 1 || !d_offsets || !d_in_map || !d_out_map) return BFG_ERR_INVALID;
    const int64_t ntot = (ndim == 2) ? (int64_t)npix * npix : (int64_t)npix * npix * npix;
    const unsigned grid = (unsigned)((ntot + 255) / 256);
    if (ndim == 2) hipLaunchKernelGGL(grid_regrid_kernel<2>, dim3(grid), dim3(256), 0, c->stream, npix, d_offsets, d_in_map, d_out_map);
    else hipLaunchKernelGGL(grid_regrid_kernel<3>, dim3(grid), dim3(256), 0, c->stream, npix, d_offsets, d_in_map, d_out_map);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_deposit_grid(bfg_ctx *c, int ndim, int64_t n_part, const double *d_pos, const double *d_mass, double L,
                     int n_grid, int mode, double *d_grid)
{
    return bfg_deposit_grid_strided(c, ndim, n_part, d_pos, ndim, d_mass, 1, L, n_grid, mode, d_grid);
}

int bfg_deposit_grid_strided(bfg_ctx *c, int ndim, int64_t n_part, const double *d_pos, int64_t pos_stride,
                             const double *d_mass, int64_t mass_stride, double L, int n_grid, int mode, double *d_grid)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if ((ndim != 2 && ndim != 3) || n_part < 0 || !(L > 0) || n_grid < 1 || (mode != BFG_DEPOSIT_NGP && mode != BFG_DEPOSIT_CIC) ||
        !d_grid || (n_part > 0 && !d_pos))
        return BFG_ERR_INVALID;
    if (n_part == 0) return BFG_OK;
    DepositParams P;
    P.ndim = ndim; P.mode = mode; P.N = n_grid; P.n_part = n_part; P.L = L; P.pos = d_pos; P.mass = d_mass; P.grid = d_grid;
    if (pos_stride < ndim || mass_stride < 1) return BFG_ERR_INVALID;
    P.pstride = pos_stride; P.mstride = mass_stride;
    // large particle sets go through the tile-privatised path (BFG_DEPOSIT=direct / tile forces either one)
    const int T = (ndim == 3) ? DepTile<3>::T : DepTile<2>::T;
    const int nt = (n_grid + T - 1) / T;
    const int64_t ntile = (ndim == 3) ? (int64_t)nt * nt * nt : (int64_t)nt * nt;
    bool tiled = n_part >= (1 << 18) && n_part < (int64_t)0x7fffffff && ntile < (1 << 26);
    if (const char *e = std::getenv("BFG_DEPOSIT")) {
        if (!std::strcmp(e, "direct")) tiled = false;
        else if (!std::strcmp(e, "tile")) tiled = n_part < (int64_t)0x7fffffff && ntile < (1 << 26);
    }
    if (!tiled) {
        const unsigned grid = (unsigned)((n_part + 255) / 256);
        if (ndim == 3) hipLaunchKernelGGL(deposit_kernel<3>, dim3(grid), dim3(256), 0, c->stream, P);
        else hipLaunchKernelGGL(deposit_kernel<2>, dim3(grid), dim3(256), 0, c->stream, P);
        HIP_TRY(hipGetLastError());
        return BFG_OK;
    }
    // slots per tile: twice the mean occupancy (a uniform 512^3 load has 4096 +- 64 per 16^3 tile); denser tiles spill
    int64_t cap = 64;
    while (cap < 2 * n_part / std::max<int64_t>(ntile, 1) && cap < (1 << 20)) cap *= 2;
    while (cap > 64 && cap * ntile > ((int64_t)1 << 30)) cap /= 2;
    if (const char *e = std::getenv("BFG_DEPOSIT_CAP")) cap = std::max<int64_t>(1, std::atoll(e));       // test hook
    const int64_t ovf_cap = n_part;
    const size_t want[5] = {(size_t)(ovf_cap + 2) * sizeof(int32_t), (size_t)(cap * ntile) * sizeof(int32_t),
                            (size_t)ntile * sizeof(int32_t), 0, 0};
    rc = dep_workspace(c, want);
    if (rc) return rc;
    DepSortParams S;
    S.d = P; S.nt = nt; S.cap = (int)cap;
    S.ovf_n = (unsigned long long *)c->dep_buf[0]; S.ovf = (int32_t *)c->dep_buf[0] + 2; S.ovf_cap = ovf_cap;
    S.perm = (int32_t *)c->dep_buf[1]; S.count = (int32_t *)c->dep_buf[2];
    HIP_TRY(hipMemsetAsync(S.count, 0, (size_t)ntile * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemsetAsync(S.ovf_n, 0, sizeof(unsigned long long), c->stream));
    const unsigned pgrid = (unsigned)std::min<int64_t>((n_part + 255) / 256, 16384);          // grid-stride
#define BFG_DEP_LAUNCH(KERNEL, GRID, THREADS)                                                                     \
    do {                                                                                                           \
        if (ndim == 3 && mode == BFG_DEPOSIT_CIC) hipLaunchKernelGGL((KERNEL<3, BFG_DEPOSIT_CIC>), dim3(GRID), dim3(THREADS), 0, c->stream, S); \
        else if (ndim == 3) hipLaunchKernelGGL((KERNEL<3, BFG_DEPOSIT_NGP>), dim3(GRID), dim3(THREADS), 0, c->stream, S);                       \
        else if (mode == BFG_DEPOSIT_CIC) hipLaunchKernelGGL((KERNEL<2, BFG_DEPOSIT_CIC>), dim3(GRID), dim3(THREADS), 0, c->stream, S);         \
        else hipLaunchKernelGGL((KERNEL<2, BFG_DEPOSIT_NGP>), dim3(GRID), dim3(THREADS), 0, c->stream, S);                                      \
    } while (0)
    timing_begin(c, 7);
    BFG_DEP_LAUNCH(dep_key_kernel, pgrid, 256);
    BFG_DEP_LAUNCH(dep_tile_kernel, (unsigned)ntile, kDepThreads);
#undef BFG_DEP_LAUNCH
    if (ndim == 3) hipLaunchKernelGGL(dep_overflow_kernel<3>, dim3(1024), dim3(256), 0, c->stream, S);
    else hipLaunchKernelGGL(dep_overflow_kernel<2>, dim3(1024), dim3(256), 0, c->stream, S);
    HIP_TRY(hipGetLastError());
    timing_end(c, 7);
    return BFG_OK;
}

int bfg_build_displacement_table(bfg_ctx *c, int geometry, int n_rows, int n_int, const double *r_int,
                                 const double *d_dens_dmo, const double *d_dens_dmb, int nr, const double *r,
                                 const double *rdelta, const double *rdelta_range, double *d_out, int32_t *status)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if ((geometry != 2 && geometry != 3) || n_rows < 0 || n_int < 3 || nr < 2 || !r_int || !r || !d_out || !status ||
        (n_rows > 0 && (!d_dens_dmo || !d_dens_dmb)) || ((rdelta != nullptr) != (rdelta_range != nullptr)))
        return BFG_ERR_INVALID;
    if (n_rows == 0) return BFG_OK;
    const size_t lds = build_lds_bytes(n_int, nr);
    if (lds > c->max_dyn_lds) return BFG_ERR_UNSUPPORTED;
    // small host arrays -> one device blob: r_int | ln r_int | r | ln r | rdelta | rdelta_range
    std::vector<double> blob;
    blob.reserve((size_t)2 * n_int + 3 * nr + n_rows);
    for (int k = 0; k < n_int; ++k) blob.push_back(r_int[k]);
    for (int k = 0; k < n_int; ++k) blob.push_back(std::log(r_int[k]));
    for (int k = 0; k < nr; ++k) blob.push_back(r[k]);
    for (int k = 0; k < nr; ++k) blob.push_back(std::log(r[k]));
    if (rdelta) {
        for (int k = 0; k < n_rows; ++k) blob.push_back(rdelta[k]);
        for (int k = 0; k < nr; ++k) blob.push_back(rdelta_range[k]);
    }
    double *d_blob = nullptr;
    int32_t *d_status = nullptr;
    HIP_TRY(hipMalloc((void **)&d_blob, blob.size() * sizeof(double)));
    if (hipMalloc((void **)&d_status, (size_t)n_rows * sizeof(int32_t)) != hipSuccess) { (void)hipFree(d_blob); return BFG_ERR_NOMEM; }
    BuildParams P;
    P.n_rows = n_rows; P.n_int = n_int; P.nr = nr; P.geometry = geometry;
    P.dens_dmo = d_dens_dmo; P.dens_dmb = d_dens_dmb;
    P.r_int = d_blob; P.lnr_int = d_blob + n_int; P.r = d_blob + 2 * n_int; P.lnr = P.r + nr;
    P.dlnr = std::log(r_int[1] / r_int[0]);                                                // get_masses :675
    P.rdelta = rdelta ? P.lnr + nr : nullptr; P.rdelta_range = rdelta ? P.rdelta + n_rows : nullptr;
    P.out = d_out; P.status = d_status;
    hipError_t e = hipMemcpyAsync(d_blob, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && lds > 64 * 1024)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(table_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(table_build_kernel, dim3((unsigned)n_rows), dim3(256), lds, c->stream, P);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(status, d_status, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_blob); (void)hipFree(d_status);
    if (e != hipSuccess) { g_last_error = std::string("bfg_build_displacement_table: ") + hipGetErrorString(e); return BFG_ERR_HIP; }
    return BFG_OK;
}

int bfg_plan_reuses(bfg_ctx *c, int64_t *count)
{
    if (!c || !count) return BFG_ERR_INVALID;
    *count = (int64_t)c->plan_reuses;
    return BFG_OK;
}

int bfg_stats_reset(bfg_ctx *c)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof(bfg_stats), c->stream));
    return BFG_OK;
}

int bfg_stats_read(bfg_ctx *c, bfg_stats *out)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!out) return BFG_ERR_INVALID;
    HIP_TRY(hipMemcpyAsync(out, c->d_stats, sizeof(bfg_stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (out->warn_mask & 0x80000000u) {       // a tile kernel refused to run (its LDS layout assumption was violated)
        g_last_error = "shell_tile_kernel: dynamic LDS does not start at address 0; results of the tile variant are invalid";
        return BFG_ERR_UNSUPPORTED;
    }
    return BFG_OK;
}

int bfg_timing_enable(bfg_ctx *c, int enable)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->timing = enable != 0;
    c->timing_mask = ~0u;
    for (int k = 0; k < kTimingSlots; ++k) { c->t_ms[k] = 0; c->t_n[k] = 0; c->ev_used[k] = 0; }
    return BFG_OK;
}

int bfg_timing_select(bfg_ctx *c, unsigned which_mask)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));       // no begin without its end
    c->timing_mask = which_mask;
    return BFG_OK;
}

int bfg_timing_read(bfg_ctx *c, int which, double *ms_total, int64_t *launches)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (which < 0 || which >= kTimingSlots) return BFG_ERR_INVALID;
    timing_fold(c, which);
    if (ms_total) *ms_total = c->t_ms[which];
    if (launches) *launches = c->t_n[which];
    return BFG_OK;
}

}  // extern "C"

#include "bfg_comm.hpp"

#if BFG_STAGE_TIMING
// profiling build only (not declared in include/bfg_mi355.h)
extern "C" int bfg_debug_stage_cycles(bfg_ctx *c, unsigned long long *out16, int reset)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    (void)hipStreamSynchronize(c->stream);
    if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(bfg::g_stage_cycles), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(bfg::g_stage_cycles), z, sizeof(z)); }
    return 0;
}
#endif

// ---- tables with more p_keys axes than the shell kernels read (bfg_ndtable.hpp) ------------------------------------------
struct bfg_ndtable {
    bfg::NdTable dev;
    double *d_blob;
};

static const double *ndtable_raxis(const bfg_ndtable *t) { return t->dev.raxis; }

// bfg_paint_shell* / bfg_baryonify_offsets* with a table of more than BFG_MAX_DIM dimensions (ParamTabulatedProfile /
// BaryonificationClass with more than three p_keys: utils/Tabulate.py:497-650, Profiles/BaryonCorrection.py:211-227, :404-408).
// All non-radial coordinates of a (halo, pixel) query are the halo's, so the multilinear read-out factors: nd_rows_kernel blends
// the 2^(n+2) corners of the halo's (z, M, p_1 ... p_n) cell into ITS radial row once (NaN rows outside the hull of any axis), and
// the shell kernels -- prep, binning, tile kernel, left-over scatter kernel: the same code as for a 3-D table -- run on those rows
// as a table without outer axes whose values start at j * hstride for halo j.  Rows are float64[n][nr]: catalogs whose rows pass
// BFG_ND_ROW_BYTES (default 4 GiB) are painted in batches of halos, each batch accumulating into the output of the one before;
// a sliced call in several batches reports its slices after the last batch (they are not final earlier).
// Does a shell call with this table go through the per-halo rows?  A table the kernels cannot read themselves: always.  A 6-D
// displacement table: always (32 corners per window node in the kernels: 8.07 ms against 3.64 on rows even with every halo
// blending its own).  Other tables that keep both forms: where a table cell holds eight halos or more on average -- the grouped
// blend then shares every corner row among the halos of a wavefront --, else the kernels' own corner blend.
// BFG_ND_FROM_DIM (A/B): every table that has the N-dimensional form takes the rows.
static bool nd_rows_pay(const bfg_table *t, const bfg_shell_args *a)
{
    if (!t->nd) return false;
    if (!t->d_blob) return true;
    if (std::getenv("BFG_ND_FROM_DIM")) return true;
    if (t->dev.ndim == BFG_MAX_DIM && !t->dev.log_values) return true;
    if (!a) return false;
    const bfg::NdTable &N = t->nd->dev;
    double cells = 1.0;
    for (int k = 0; k < N.nouter; ++k) cells *= (double)std::max(1, N.oshape[k] - 1);
    return (double)a->n_halo >= 8.0 * cells;
}

static int run_shell_nd(bfg_ctx *c, const bfg_shell_args *a, const bfg_table *t, const bfg_spline *s, double *d_out, int mode,
                        int n_slices, bfg_slice_fn slice_fn, void *slice_user)
{
    if (!a || !t || !s || !d_out) return BFG_ERR_INVALID;
    const bfg::NdTable &N = t->nd->dev;
    if (a->n_extra != N.nouter - 2 || a->cat_stride < 4 + a->n_extra || a->n_halo < 0) return BFG_ERR_INVALID;
    if (a->n_halo > 0 && !a->d_catalog) return BFG_ERR_INVALID;
    double cap_bytes = 4294967296.0;
    if (const char *e = std::getenv("BFG_ND_ROW_BYTES")) cap_bytes = std::max(1.0, std::atof(e));
    const int64_t per = std::max<int64_t>(1, (int64_t)(cap_bytes / (8.0 * (double)N.nr)));
    const int64_t nb_max = std::min<int64_t>(std::max<int64_t>(a->n_halo, 1), per);
    if (nb_max * N.nr > c->ndrows_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_ndrows) (void)hipFree(c->d_ndrows);
        c->d_ndrows = nullptr; c->ndrows_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->d_ndrows, (size_t)(nb_max * N.nr) * sizeof(double)));
        c->ndrows_cap = nb_max * N.nr;
    }
    // the rows of halos that share a table cell are blended together (BFG_ND_ROWS=plain: every halo by itself, the A/B; also the
    // path of tables with more than kNdMaxCells cells)
    int64_t ncell = 1;
    for (int k = 0; k < N.nouter && ncell <= bfg::kNdMaxCells; ++k) ncell *= std::max(1, N.oshape[k] - 1);
    bool blocked = ncell <= bfg::kNdMaxCells && nb_max < ((int64_t)1 << 31);
    int rshift = 0;                                             // R = 2^rshift counters per cell (<= 32, cells x R <= kNdMaxCells)
    while (blocked && rshift < 5 && (ncell << (rshift + 1)) <= bfg::kNdMaxCells) ++rshift;
    if (blocked) ncell <<= rshift;                              // from here on: the number of sort keys
    if (const char *e = std::getenv("BFG_ND_ROWS")) if (!std::strcmp(e, "plain")) blocked = false;
    const int64_t nblk = (ncell + 1023) / 1024;
    size_t y_off = 0;
    if (blocked) {
        y_off = ((size_t)(3 * nb_max + 2 * ncell + 1 + nblk + 1) * sizeof(int32_t) + 15) / 16 * 16;
        const size_t want = y_off + (size_t)nb_max * N.nouter * sizeof(double);
        if (want > c->ndsort_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_ndsort) (void)hipFree(c->d_ndsort);
            c->d_ndsort = nullptr; c->ndsort_cap = 0;
            HIP_TRY(hipMalloc(&c->d_ndsort, want));
            c->ndsort_cap = want;
        }
    }
    bfg_table tv;
    tv.dev = t->dev; tv.d_blob = nullptr; tv.nd = nullptr;
    tv.dev.ndim = 3; tv.dev.nouter = 0; tv.dev.values = c->d_ndrows; tv.dev.hstride = N.nr;
    const int64_t n_batches = std::max<int64_t>(1, (a->n_halo + nb_max - 1) / nb_max);
    const bool sliced_here = slice_fn && n_batches == 1;
    for (int64_t b = 0; b < n_batches; ++b) {
        const int64_t j0 = b * nb_max, nb = std::min(nb_max, a->n_halo - j0);
        bfg_shell_args ab = *a;
        ab.d_catalog = a->d_catalog ? a->d_catalog + j0 * (int64_t)a->cat_stride : nullptr;
        ab.n_halo = std::max<int64_t>(nb, 0);
        ab.n_extra = 0;
        if (b > 0) ab.flags &= ~(uint32_t)(BFG_SHELL_OUT_OVERWRITE | BFG_SHELL_OUT_IS_ZERO);      // accumulate into the batches before
        if (nb > 0 && blocked) {
            // halos grouped by table cell, eight of a cell per wavefront (bfg_ndtable.hpp)
            int32_t *const d_cell = (int32_t *)c->d_ndsort, *const d_perm = d_cell + nb_max, *const d_rank = d_perm + nb_max;
            int32_t *const d_count = d_rank + nb_max;
            int32_t *const d_start = d_count + ncell, *const d_bsum = d_start + ncell + 1, *const d_total = d_bsum + nblk;
            double *const d_y = (double *)((char *)c->d_ndsort + y_off);
            HIP_TRY(hipMemsetAsync(d_count, 0, (size_t)ncell * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(bfg::nd_cell_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, N, ab.d_catalog, nb,
                               a->cat_stride, rshift, d_cell, d_rank, d_y, d_count, c->d_ndrows, c->d_stats);
            hipLaunchKernelGGL(bfg::snap_scan_block_kernel, dim3((unsigned)nblk), dim3(256), 0, c->stream, ncell, d_count, d_start, d_bsum);
            hipLaunchKernelGGL(bfg::snap_scan_sums_kernel, dim3(1), dim3(1024), 0, c->stream, (int)nblk, d_bsum, d_total);
            hipLaunchKernelGGL(bfg::snap_scan_add_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, ncell, d_start,
                               d_bsum, d_count, d_total);
            hipLaunchKernelGGL(bfg::nd_fill_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, nb, d_cell, d_rank, d_start,
                               d_perm);
            const unsigned grid = (unsigned)std::min<int64_t>((nb + 4 * bfg::kNdBlockHalos - 1) / (4 * bfg::kNdBlockHalos), (int64_t)c->n_cu * 16);
            hipLaunchKernelGGL(bfg::nd_rows_blocked_kernel, dim3(grid), dim3(256), 0, c->stream, N, rshift, d_cell, d_perm, d_start + ncell,
                               d_y, c->d_ndrows);
            HIP_TRY(hipGetLastError());
        } else if (nb > 0) {
            const unsigned grid = (unsigned)std::min<int64_t>((nb + 3) / 4, (int64_t)c->n_cu * 16);
            hipLaunchKernelGGL(bfg::nd_rows_kernel, dim3(grid), dim3(256), 0, c->stream, N, ab.d_catalog, nb, a->cat_stride, c->d_ndrows,
                               c->d_stats);
            HIP_TRY(hipGetLastError());
        }
        const int rc = run_shell(c, &ab, &tv, s, d_out, mode, sliced_here ? n_slices : 1, sliced_here ? slice_fn : nullptr, slice_user);
        if (rc) return rc;
    }
    if (slice_fn && !sliced_here) {
        int64_t cuts[kMaxSlices + 1];
        const int K = shell_slice_cuts(a->nside, mode, n_slices, cuts);
        for (int k = 0; k < K; ++k)
            if (slice_fn(slice_user, k, K, cuts[k], cuts[k + 1]) != 0) { g_last_error = "the slice callback failed"; return BFG_ERR_INVALID; }
    }
    return BFG_OK;
}

int bfg_ndtable_create(bfg_ctx *c, int n_outer, const int64_t *outer_shape, const double *const *outer_axes, int64_t nr,
                       const double *raxis, const double *values, bfg_ndtable **out)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!outer_shape || !outer_axes || !raxis || !values || !out) return BFG_ERR_INVALID;
    if (n_outer < 2 || nr < 2 || nr > (1 << 24)) return BFG_ERR_INVALID;
    if (n_outer > bfg::kNdMaxOuter) return BFG_ERR_UNSUPPORTED;
    size_t n_axes = (size_t)nr;
    int64_t rows = 1;
    for (int k = 0; k < n_outer; ++k) {
        if (outer_shape[k] < 2 || outer_shape[k] > (1 << 24) || !outer_axes[k]) return BFG_ERR_INVALID;
        for (int64_t i = 1; i < outer_shape[k]; ++i) if (!(outer_axes[k][i] > outer_axes[k][i - 1])) return BFG_ERR_INVALID;
        n_axes += (size_t)outer_shape[k];
        rows *= outer_shape[k];
        if (rows > ((int64_t)1 << 40) / nr) return BFG_ERR_UNSUPPORTED;
    }
    for (int64_t i = 1; i < nr; ++i) if (!(raxis[i] > raxis[i - 1])) return BFG_ERR_INVALID;
    const size_t total = (size_t)rows * (size_t)nr;
    bfg_ndtable *t = new bfg_ndtable();
    std::memset(&t->dev, 0, sizeof(t->dev));
    t->d_blob = nullptr;
    if (hipMalloc((void **)&t->d_blob, (n_axes + total) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); delete t; return BFG_ERR_NOMEM; }
    std::vector<double> ax(n_axes);
    size_t pos = 0;
    bfg::NdTable &D = t->dev;
    D.nouter = n_outer; D.nr = (int)nr;
    { int64_t st = nr; for (int k = n_outer - 1; k >= 0; --k) { D.ostride[k] = st; st *= outer_shape[k]; } }
    for (int k = 0; k < n_outer; ++k) {
        std::copy(outer_axes[k], outer_axes[k] + outer_shape[k], ax.begin() + pos);
        D.oaxis[k] = t->d_blob + pos; D.oshape[k] = (int)outer_shape[k];
        pos += (size_t)outer_shape[k];
    }
    std::copy(raxis, raxis + nr, ax.begin() + pos);
    D.raxis = t->d_blob + pos; pos += (size_t)nr;
    D.values = t->d_blob + pos;
    if (hipMemcpyAsync(t->d_blob, ax.data(), n_axes * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(t->d_blob + n_axes, values, total * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        g_last_error = std::string("bfg_ndtable_create upload: ") + hipGetErrorString(hipGetLastError());
        (void)hipFree(t->d_blob); delete t;
        return BFG_ERR_HIP;
    }
    *out = t;
    return BFG_OK;
}

int bfg_ndtable_destroy(bfg_ctx *c, bfg_ndtable *t)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!t) return BFG_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(t->d_blob);
    delete t;
    return BFG_OK;
}

int bfg_ndtable_rows(bfg_ctx *c, const bfg_ndtable *t, const double *d_catalog, int64_t n_halo, int cat_stride, double *d_rows)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!t || n_halo < 0 || (n_halo > 0 && (!d_catalog || !d_rows)) || cat_stride < 2 + t->dev.nouter) return BFG_ERR_INVALID;
    if (n_halo == 0) return BFG_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n_halo + 3) / 4, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL(bfg::nd_rows_kernel, dim3(grid), dim3(256), 0, c->stream, t->dev, d_catalog, n_halo, cat_stride, d_rows,
                       (bfg_stats *)nullptr);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}

int bfg_ndtable_read(bfg_ctx *c, const bfg_ndtable *t, const double *d_rows, int64_t n, const int32_t *d_halo, const double *d_r_com,
                     const double *d_shift, const double *d_rcut, const double *d_scale, int exp_values, double *d_out,
                     unsigned int *d_r_oob)
{
    DeviceGuard dg_;
    int rc = ctx_enter(c, dg_);
    if (rc) return rc;
    if (!t || n < 0 || (n > 0 && (!d_rows || !d_halo || !d_r_com || !d_out))) return BFG_ERR_INVALID;
    if (n == 0) return BFG_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)c->n_cu * 16);
    hipLaunchKernelGGL(bfg::nd_read_kernel, dim3(grid), dim3(256), 0, c->stream, d_rows, t->dev.nr, t->dev.raxis, n, d_halo, d_r_com,
                       d_shift, d_rcut, d_scale, exp_values ? 1 : 0, d_out, d_r_oob);
    HIP_TRY(hipGetLastError());
    return BFG_OK;
}
