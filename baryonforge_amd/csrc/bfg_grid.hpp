// bfg_grid.hpp -- periodic Cartesian grid runners (BaryonForge/Runners/Map2DRunner.py) on the GPU:
// PaintProfilesGrid.process (:676-829), BaryonifyGrid.process (:431-621) and regrid_pixels_2D / _3D (:14-162).
//
// The reference cuts a (2w)^d window of pixels out of the periodic map around every halo (w from eps R / res, even,
// clipped), evaluates radii on a *stretched* offset grid np.linspace(-w, w, 2w) * res shifted by the sub-pixel offset of
// the halo from its nearest pixel centre, and pairs those offsets with the window's pixels in np.meshgrid(..., 'xy')
// order against inds[x_inds, :][:, y_inds] -- so the first window axis carries the y offsets and the second the x
// offsets.  All of that is reproduced literally (the oracle restates it line by line and is pinned by the reference's
// own output).  One workgroup per halo walks its window; values / offsets go to the map with f64 global atomics.
// Included by bfg_mi355.hip after DevTable / massdef_radius / SnapHalo helpers are defined.
#pragma once

namespace bfg {

struct __align__(16) GridHalo {
    double d[3];                     // bins[cen] - halo coordinate
    double rmask;                    // paint: eps R_com (pixels at or beyond it are masked, :815)
    double xcut;                     // baryonify: model.epsilon_max * R_model_com
    double lnshift;                  // ln(R_model_com) for Rdelta_sampling tables, else 0
    int32_t cen[3];
    int32_t nsize, flags, pad;
};

struct GridParams {
    int ndim, npix, mode, rdelta;    // mode: MODE_PAINT / MODE_BARYONIFY
    int64_t n_halo;
    double res, a, eps_run, eps_model;
    bfg_massdef md_run, md_model;
    const double *bins;              // [npix] pixel centres
    const double *halo;              // [n_halo][stride]: M, lnM (table coordinate), x, y, z, extras...
    const double *rmat;              // [n_halo][4] shear matrices of the 2D ellipticity option, or nullptr
    int halo_stride, n_extra;
    DevTable tab;
    GridHalo *gh;                    // [n_halo]
    double *hrow;                    // [n_halo][tab.nr] blended radial rows (ln T for paint, d for baryonify)
    double *out;                     // paint: map [npix^ndim]; baryonify: offsets [npix^ndim][ndim] (pixel widths)
    bfg_stats *stats;
};

// np.argmin(np.abs(bins - h)): nearest pixel centre, the lowest index on ties
__device__ inline int grid_nearest_bin(const double *bins, int n, double res, double h)
{
    // a coordinate at +-infinity (or NaN) is equally far from every pixel centre: np.argmin returns index 0
    // (pinned by tests/golden/grid.npz x2 / x3: the reference cuts such a halo's window around pixel 0)
    if (!(fabs(h) < __builtin_huge_val())) return 0;
    int i = (int)floor((h - bins[0]) / res + 0.5);
    i = min(max(i, 0), n - 1);
    int best = i;
    double dbest = fabs(bins[i] - h);
    for (int c = max(i - 1, 0); c <= min(i + 1, n - 1); ++c) {
        const double dc = fabs(bins[c] - h);
        if (dc < dbest || (dc == dbest && c < best)) { best = c; dbest = dc; }
    }
    return best;
}

__global__ __launch_bounds__(64) void grid_halo_kernel(const GridParams P)
{
    const int64_t j = blockIdx.x;
    const int lane = threadIdx.x;
    const double *c = P.halo + j * P.halo_stride;
    const double M = c[0], lnM = c[1];
    const DevTable &T = P.tab;
    __shared__ double s_w[kMaxCorner];
    __shared__ int64_t s_off[kMaxCorner];
    __shared__ int s_oob;
    if (lane == 0) {
        const double R = massdef_radius(P.md_run, M, P.a);                          // physical Mpc (:471 / :708)
        const double Rcom = R / P.a;
        const double Rm = massdef_radius(P.md_model, M, P.a) / P.a;
        double rcut;
        if (P.mode == MODE_PAINT) rcut = P.eps_run * Rcom;                          // :718
        else {
            double bmax = P.bins[0];
            bmax = fmax(bmax, P.bins[P.npix - 1]);
            rcut = fmin(fmax(P.eps_run * R / P.a, 0.0), 0.5 * bmax);                // :473-474 (bins ascend: max = last)
        }
        double ns = 2.0 * rcut / P.res;                                             // :484-486 / :718-720
        long long nsz = (long long)floor(ns / 2.0) * 2;
        if (!(ns == ns)) nsz = 2;
        nsz = std::min<long long>(std::max<long long>(nsz, 2), P.npix / 2);
        GridHalo h;
        h.nsize = (int)nsz; h.flags = 0; h.pad = 0;
        for (int k = 0; k < 3; ++k) { h.cen[k] = 0; h.d[k] = 0.0; }
        for (int k = 0; k < P.ndim; ++k) {
            h.cen[k] = grid_nearest_bin(P.bins, P.npix, P.res, c[2 + k]);           // :492-494
            h.d[k] = P.bins[h.cen[k]] - c[2 + k];                                   // :500-502
        }
        h.rmask = Rcom * P.eps_run;                                                 // :815
        h.xcut = P.eps_model * Rm;                                                  // BaryonCorrection.py:410
        h.lnshift = P.rdelta ? log(Rm) : 0.0;
        bool oob = false;
        uint32_t warn = 0;
        int idx[BFG_MAX_DIM];
        double wt[BFG_MAX_DIM];
        for (int k = 0; k < T.nouter; ++k) {
            const double x = (k == 0) ? log(1.0 / P.a) : (k == 1) ? lnM : c[5 + (k - 2)];
            const double *g = T.oaxis[k];
            const int n = T.oshape[k];
            if (!(x >= g[0]) || !(x <= g[n - 1])) {
                oob = true;
                if (k == 0) warn |= BFG_WARN_Z_RANGE;
                if (k == 1) warn |= BFG_WARN_M_RANGE;
            }
            idx[k] = find_interval(g, n, x);
            wt[k] = (x - g[idx[k]]) / (g[idx[k] + 1] - g[idx[k]]);
        }
        const int ncorner = 1 << T.nouter;
        for (int cc = 0; cc < ncorner; ++cc) {
            double w = 1.0;
            int64_t off = 0;
            for (int k = 0; k < T.nouter; ++k) {
                const int bit = (cc >> (T.nouter - 1 - k)) & 1;
                w *= bit ? wt[k] : 1.0 - wt[k];
                off += (int64_t)(idx[k] + bit) * T.ostride[k];
            }
            s_w[cc] = w; s_off[cc] = off;
        }
        s_oob = oob ? 1 : 0;
        if (oob) {
            h.flags = HF_OOB;
            atomicAdd((unsigned long long *)&P.stats->halos_out_of_table, 1ull);
            atomicOr(&P.stats->warn_mask, warn);
        }
        P.gh[j] = h;
    }
    __syncthreads();
    const int ncorner = 1 << T.nouter;
    for (int i = lane; i < T.nr; i += 64) {
        double b = 0.0;
        for (int cc = 0; cc < ncorner; ++cc) b = fma(T.values[s_off[cc] + i], s_w[cc], b);
        P.hrow[j * T.nr + i] = s_oob ? nan("") : b;
    }
}

// offset m of np.linspace(-n/2, n/2, n) * res  (numpy: arange(n) * step + start, the last element set to stop)
__device__ inline double grid_linspace(int m, int n, double res)
{
    const double start = -0.5 * (double)n, stop = 0.5 * (double)n;
    const double step = (stop - start) / (double)(n - 1);
    const double v = (m == n - 1) ? stop : (double)m * step + start;
    return v * res;
}

template <int NDIM, int MODE>
__global__ __launch_bounds__(256) void grid_window_kernel(const GridParams P)
{
    const int64_t j = blockIdx.x;
    const GridHalo h = P.gh[j];
    const DevTable &T = P.tab;
    const int n = h.nsize, w = n / 2, N = P.npix;
    const int64_t ncell = (NDIM == 2) ? (int64_t)n * n : (int64_t)n * n * n;
    const double *row = P.hrow + j * T.nr;
    const double r_lo = T.raxis[0], r_hi = T.raxis[T.nr - 1];
    unsigned long long cnt = 0, n_oob = 0;
    for (int64_t q = threadIdx.x; q < ncell; q += blockDim.x) {
        // window element (a, b[, c]) in C order: map pixel (x_inds[a], y_inds[b][, z_inds[c]]), offsets of the
        // 'xy' meshgrid: x offset = x[b], y offset = x[a], z offset = x[c]   (:507-516 / :545-556)
        int ia, ib, ic = 0;
        if (NDIM == 2) { ia = (int)(q / n); ib = (int)(q % n); }
        else { ia = (int)(q / ((int64_t)n * n)); ib = (int)((q / n) % n); ic = (int)(q % n); }
        int pa = h.cen[0] - w + ia; pa = (pa < 0) ? pa + N : pa; pa = (pa >= N) ? pa - N : pa;          // pick_indices
        int pb = h.cen[1] - w + ib; pb = (pb < 0) ? pb + N : pb; pb = (pb >= N) ? pb - N : pb;
        int64_t flat = (int64_t)pa * N + pb;
        double comp[3];
        comp[0] = grid_linspace(ib, n, P.res) + h.d[0];
        comp[1] = grid_linspace(ia, n, P.res) + h.d[1];
        double r2 = comp[0] * comp[0] + comp[1] * comp[1];
        if (NDIM == 3) {
            int pc = h.cen[2] - w + ic; pc = (pc < 0) ? pc + N : pc; pc = (pc >= N) ? pc - N : pc;
            flat = flat * N + pc;
            comp[2] = grid_linspace(ic, n, P.res) + h.d[2];
            r2 += comp[2] * comp[2];
        }
        const double r = sqrt(r2);                              // circular radius: the unit vectors always use it
        double rm = r;                                          // radius handed to the model
        if (NDIM == 2 && P.rmat) {                              // (x, y) @ Rmat  (:520-524 / :753-757)
            const double *Rm = P.rmat + j * 4;
            const double xe = comp[0] * Rm[0] + comp[1] * Rm[2], ye = comp[0] * Rm[1] + comp[1] * Rm[3];
            rm = sqrt(xe * xe + ye * ye);
        }
        ++cnt;
        // table read-out on the halo's blended row: NaN outside the radial hull
        const double rin = log(rm) - ((MODE == MODE_BARYONIFY) ? h.lnshift : 0.0);
        double val = nan("");
        if ((rin >= r_lo) && (rin <= r_hi)) {
            const int i = find_interval(T.raxis, T.nr, rin);
            const double f = (rin - T.raxis[i]) / (T.raxis[i + 1] - T.raxis[i]);
            val = row[i] * (1.0 - f) + row[i + 1] * f;
        } else if (!(h.flags & HF_OOB)) ++n_oob;
        if (MODE == MODE_PAINT) {
            const double Pv = exp(val);                                              // Tabulate.py:319
            if ((fabs(Pv) <= 1.797e308) && (rm < h.rmask)) unsafeAtomicAdd(P.out + flat, Pv);         // :812-823
        } else {
            // BaryonCorrection.py:410-411: 0 at or beyond eps R, else the (possibly NaN) table value; the reference adds
            // NaN contributions too -- they poison the pixel's offset, which is zeroed before the regrid (:591 / :603)
            const double off = (rm < h.xcut) ? val / P.res : 0.0;                    // :530 / :570, in pixel widths
            for (int k = 0; k < NDIM; ++k) unsafeAtomicAdd(P.out + flat * NDIM + k, off * (comp[k] / r));
        }
    }
    __shared__ unsigned long long s_red[2][4];
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_down(cnt, o, 64); n_oob += __shfl_down(n_oob, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = cnt; s_red[1][threadIdx.x >> 6] = n_oob; }
    __syncthreads();
    if (threadIdx.x == 0) {                                    // one pair of same-address atomics per workgroup (= halo)
        cnt = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        n_oob = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (cnt) atomicAdd((unsigned long long *)&P.stats->pixel_updates, cnt);
        if (n_oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_oob);
            if (!(MODE == MODE_BARYONIFY && P.rdelta)) atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);
        }
    }
}

// ---- tile-privatised window pass --------------------------------------------------------------------------------------
// grid_window_kernel pays one global f64 atomic per window pixel (1 per pixel for paint, ndim for the offsets) plus libm
// sqrt / log / exp and a bisection of the radial axis.  Here the map is cut into tiles of 64^2 / 16^3 pixels; the halos
// are binned to the tiles their window touches (count -> scan -> fill, as the shell path does for sky tiles), one
// workgroup per tile accumulates in LDS (ds_add_f64) -- each of its 4 wavefronts takes one (halo, tile) pair at a
// time and walks the intersection of the halo's window with the tile -- and the tile is added to the map once with
// plain coalesced read-add-writes (every pixel belongs to exactly one tile).  Same per-pixel formulas as above
// (stretched linspace offsets, 'xy' pairing, pick_indices wrap); ln / exp from the shell kernels' LDS tables.
// Needs npix >= 2 tile sides (a window, at most npix / 2 wide, then meets a tile in at most one interval per axis).
template <int NDIM> struct GridTile {
    static constexpr int TS = (NDIM == 2) ? 64 : 16;                 // production tile side
    static constexpr int TS_SMALL = (NDIM == 2) ? 16 : 8;            // BFG_GRID=small: the same kernels on the small
                                                                     // maps of the golden fixtures (test hook)
};

struct GridBin {
    int nt;                          // tiles per axis = ceil(npix / TS)
    int fill;                        // 0: count pass, 1: fill pass
    int32_t *count;                  // [nt^ndim] counts, then fill cursors
    const int32_t *start;            // [nt^ndim + 1]
    int32_t *pairs;                  // [pair_cap] halo ids grouped by tile
    int64_t pair_cap;
    const double2 *logtab;           // [kLogTab]
    const double *exptab;            // [kExpTab]
    unsigned long long *winpix;      // count pass: sum of window sizes n^ndim over the halos
};

template <int NDIM, int TS>
__global__ __launch_bounds__(256) void grid_bin_kernel(const GridParams P, const GridBin B)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.n_halo) return;
    const GridHalo h = P.gh[j];
    const int N = P.npix, n = h.nsize, w = n / 2;
    int lo[NDIM][2], cnt[NDIM][2], tot[NDIM];
    int64_t total = 1;
    for (int k = 0; k < NDIM; ++k) {
        int s = (h.cen[k] - w) % N; if (s < 0) s += N;          // first window pixel (pick_indices wrap)
        const int e = s + n;
        lo[k][0] = s / TS; cnt[k][0] = (min(e, N) - 1) / TS - lo[k][0] + 1;
        lo[k][1] = 0; cnt[k][1] = (e > N) ? (e - N - 1) / TS + 1 : 0;
        tot[k] = cnt[k][0] + cnt[k][1];
        total *= tot[k];
    }
    if (!B.fill) {                                             // window pixels of this wavefront's halos -> one atomic
        unsigned long long wp = (NDIM == 2) ? (unsigned long long)n * n : (unsigned long long)n * n * n;
        for (int o = 32; o > 0; o >>= 1) wp += __shfl_down(wp, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(B.winpix, wp);
    }
    for (int64_t q = 0; q < total; ++q) {
        int64_t rem = q, tile = 0, mul = 1;
        for (int k = NDIM - 1; k >= 0; --k) {
            const int i = (int)(rem % tot[k]); rem /= tot[k];
            const int t = (i < cnt[k][0]) ? lo[k][0] + i : lo[k][1] + (i - cnt[k][0]);
            tile += (int64_t)t * mul; mul *= B.nt;
        }
        if (!B.fill) atomicAdd(&B.count[tile], 1);
        else {
            const int64_t pos = (int64_t)B.start[tile] + atomicAdd(&B.count[tile], 1);
            if (pos < B.pair_cap) B.pairs[pos] = (int32_t)j;
        }
    }
}

template <int NDIM, int MODE, int TS>
__global__ __launch_bounds__(256) void grid_tile_kernel(const GridParams P, const GridBin B)
{
    constexpr int NC = (NDIM == 2) ? TS * TS : TS * TS * TS;
    constexpr int NA = (MODE == MODE_PAINT) ? 1 : NDIM;
    extern __shared__ double smem_gt[];
    double *acc = smem_gt;                                               // [NC][NA]
    double2 *logtab = reinterpret_cast<double2 *>(acc + NC * NA);
    double *exptab = reinterpret_cast<double *>(logtab + kLogTab);
    const int tile = blockIdx.x;
    const int p0 = B.start[tile], p1 = (int)min((int64_t)B.start[tile + 1], B.pair_cap);
    if (p0 >= p1) return;                                                // nothing lands in this tile
    const DevTable &T = P.tab;
    const int N = P.npix;
    int t0[NDIM], tlen[NDIM];
    { int rem = tile; for (int k = NDIM - 1; k >= 0; --k) { t0[k] = (rem % B.nt) * TS; rem /= B.nt; tlen[k] = min(TS, N - t0[k]); } }
    for (int e = threadIdx.x; e < NC * NA; e += 256) acc[e] = 0.0;
    if (threadIdx.x < kLogTab) logtab[threadIdx.x] = B.logtab[threadIdx.x];
    if (threadIdx.x < kExpTab) exptab[threadIdx.x] = B.exptab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double r_lo = T.raxis[0], r_hi = T.raxis[T.nr - 1];
    unsigned long long cnt = 0, n_oob = 0;
    for (int pi = p0 + (int)(threadIdx.x >> 6); pi < p1; pi += 4) {
        const int j = __builtin_amdgcn_readfirstlane(B.pairs[pi]);
        const GridHalo &h = P.gh[j];                                     // wave-uniform: scalar loads
        const int n = h.nsize, w = n / 2;
        int ia0[NDIM], l0[NDIM], len[NDIM];
        int ncell = 1;
        for (int k = 0; k < NDIM; ++k) {
            int s = (h.cen[k] - w) % N; if (s < 0) s += N;
            int a0 = t0[k] - s; if (a0 < 0) a0 += N;                     // window index of the tile's first pixel
            if (a0 < n) { ia0[k] = a0; l0[k] = 0; len[k] = min(n - a0, tlen[k]); }
            else {
                int b0 = s - t0[k]; if (b0 < 0) b0 += N;                 // tile-local index of the window's first pixel
                if (b0 < tlen[k]) { ia0[k] = 0; l0[k] = b0; len[k] = min(n, tlen[k] - b0); }
                else { ia0[k] = 0; l0[k] = 0; len[k] = 0; }
            }
            ncell *= len[k];
        }
        const double *row = P.hrow + (int64_t)j * T.nr;
        const float inv_last = 1.0f / (float)max(len[NDIM - 1], 1);
        const float inv_mid = (NDIM == 3) ? 1.0f / (float)max(len[1], 1) : 0.0f;
        for (int q = lane; q < ncell; q += 64) {
            // (qa, qb[, qc]) = position inside the intersection, last axis fastest; quotients by float reciprocal
            // (q < 4096: (q + 0.5) / len is at least 0.5 / 64 away from an integer)
            int qa, qb, qc = 0;
            if (NDIM == 2) {
                qa = (int)(((float)q + 0.5f) * inv_last);
                qb = q - qa * len[1];
            } else {
                const int qab = (int)(((float)q + 0.5f) * inv_last);
                qc = q - qab * len[2];
                qa = (int)(((float)qab + 0.5f) * inv_mid);
                qb = qab - qa * len[1];
            }
            const int ia = ia0[0] + qa, ib = ia0[1] + qb;
            int e = (l0[0] + qa) * TS + (l0[1] + qb);
            // offsets of the 'xy' meshgrid: x offset = x[b], y offset = x[a], z offset = x[c]   (:507-516 / :545-556)
            double comp[3];
            comp[0] = grid_linspace(ib, n, P.res) + h.d[0];
            comp[1] = grid_linspace(ia, n, P.res) + h.d[1];
            double r2 = comp[0] * comp[0] + comp[1] * comp[1];
            if (NDIM == 3) {
                const int ic = ia0[2] + qc;
                e = e * TS + (l0[2] + qc);
                comp[2] = grid_linspace(ic, n, P.res) + h.d[2];
                r2 += comp[2] * comp[2];
            }
            const double r = sqrt(r2);                                   // circular radius: the unit vectors always use it
            double rm = r, rm2 = r2;                                     // radius handed to the model
            if (NDIM == 2 && P.rmat) {                                   // (x, y) @ Rmat  (:520-524 / :753-757)
                const double *Rm = P.rmat + (int64_t)j * 4;
                const double xe = comp[0] * Rm[0] + comp[1] * Rm[2], ye = comp[0] * Rm[1] + comp[1] * Rm[3];
                rm2 = xe * xe + ye * ye;
                rm = sqrt(rm2);
            }
            ++cnt;
            double rin;
            if (rm2 >= 1e-290 && rm2 <= 1e290) {
                rin = 0.5 * fast_log(rm2, logtab);
                if (MODE == MODE_BARYONIFY) rin -= h.lnshift;
                if (fabs(rin - r_lo) < 1e-9 || fabs(rin - r_hi) < 1e-9) rin = log(rm) - ((MODE == MODE_BARYONIFY) ? h.lnshift : 0.0);
            } else rin = log(rm) - ((MODE == MODE_BARYONIFY) ? h.lnshift : 0.0);
            double val = nan("");
            if ((rin >= r_lo) && (rin <= r_hi)) {
                int i;
                double f;
                if (T.r_uniform) {
                    const double t = (rin - T.r0) * T.inv_dr;
                    i = min(max((int)t, 0), T.nr - 2);
                    f = t - (double)i;
                } else {
                    i = find_interval(T.raxis, T.nr, rin);
                    f = (rin - T.raxis[i]) / (T.raxis[i + 1] - T.raxis[i]);
                }
                val = row[i] * (1.0 - f) + row[i + 1] * f;
            } else if (!(h.flags & HF_OOB)) ++n_oob;
            if (MODE == MODE_PAINT) {
                const double Pv = (T.hot || !(fabs(val) < 700.0)) ? exp(val) : fast_exp(val, exptab);      // Tabulate.py:319
                if ((fabs(Pv) <= 1.797e308) && (rm < h.rmask)) unsafeAtomicAdd(&acc[e], Pv);               // :812-823
            } else {
                const double off = (rm < h.xcut) ? val / P.res : 0.0;                                      // :530 / :570
                const double s = off / r;
                for (int k = 0; k < NDIM; ++k) unsafeAtomicAdd(&acc[e * NDIM + k], s * comp[k]);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NC; e += 256) {
        int rem = e;
        int64_t flat = 0, mul = 1;
        bool inside = true;
        for (int k = NDIM - 1; k >= 0; --k) {
            const int l = rem % TS; rem /= TS;
            inside = inside && (l < tlen[k]);
            flat += (int64_t)(t0[k] + l) * mul; mul *= N;
        }
        if (!inside) continue;
        for (int k = 0; k < NA; ++k) {
            const double v = acc[e * NA + k];
            if (v != 0.0) P.out[flat * NA + k] += v;                     // this tile owns the pixel; NaN != 0 is added too
        }
    }
    __shared__ unsigned long long s_red[2][4];
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_down(cnt, o, 64); n_oob += __shfl_down(n_oob, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = cnt; s_red[1][threadIdx.x >> 6] = n_oob; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        n_oob = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (cnt) atomicAdd((unsigned long long *)&P.stats->pixel_updates, cnt);
        if (n_oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_oob);
            if (!(MODE == MODE_BARYONIFY && P.rdelta)) atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);
        }
    }
}

// regrid_pixels_2D / _3D: one thread per source pixel
template <int NDIM>
__global__ __launch_bounds__(256) void grid_regrid_kernel(int N, const double *__restrict__ offsets,
                                                          const double *__restrict__ in_map, double *__restrict__ out_map)
{
    const int64_t ntot = (NDIM == 2) ? (int64_t)N * N : (int64_t)N * N * N;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ntot) return;
    const double val = in_map[q];
    if (val == 0.0) return;                                   // overlap * 0 adds nothing
    // np.meshgrid(arange, ..., indexing='xy') flattened: x = second index, y = first index, z = third (:588-606)
    int gi, gj, gk = 0;
    if (NDIM == 2) { gi = (int)(q / N); gj = (int)(q % N); }
    else { gi = (int)(q / ((int64_t)N * N)); gj = (int)((q / N) % N); gk = (int)(q % N); }
    const double g[3] = {(double)gj, (double)gi, (double)gk};
    double start[3], end[3];
    int base[3], last[3];
    for (int k = 0; k < NDIM; ++k) {
        double o = offsets[q * NDIM + k];
        if (!(fabs(o) <= 1.797e308)) o = 0.0;                 // :591 / :603
        double s = fmod(o + g[k], (double)N);                 // python %: result in [0, N)
        if (s < 0.0) s += (double)N;
        if (s >= (double)N) s -= (double)N;
        start[k] = s; end[k] = s + 1.0;
        base[k] = (int)s - 2;                                 // range(int(start) - 2, int(end) + 2)
        last[k] = (int)end[k] + 2;
    }
    auto overlap = [&](int k, int cc, int &cell) -> double {
        int c = cc;
        if (c < 0) c += N;
        if (c + 1 > N) c = c % N;
        cell = c;
        double d = fmin((double)c + 1.0, end[k]) - fmax((double)c, start[k]);
        if (d < 0) d = fmin((double)c + 1.0, end[k] + N) - fmax((double)c, start[k] + N);
        if (d < 0) d = fmin((double)c + 1.0, end[k] - N) - fmax((double)c, start[k] - N);
        return d;
    };
    // The reference scans range(int(start) - 2, int(end) + 2) on every axis (5-6 cells, 6^d overlap evaluations).  A unit
    // cube starting at s in [0, N) can only overlap cells int(s) and int(s) + 1, and for N >= 7 the scanned range never
    // visits a cell twice through the periodic wrap, so evaluating the SAME overlap expression on just those two cells
    // gives the same deposits bit for bit; smaller grids keep the literal scan (with its double visits).
    const bool narrow = N >= 7;
    int lo[3], hi[3];
    for (int k = 0; k < NDIM; ++k) { lo[k] = narrow ? (int)start[k] : base[k]; hi[k] = narrow ? (int)start[k] + 2 : last[k]; }
    for (int ci = lo[1]; ci < hi[1]; ++ci) {                  // i: y range
        int ii; const double dy = overlap(1, ci, ii);
        if (!(dy > 0)) continue;
        for (int cj = lo[0]; cj < hi[0]; ++cj) {              // j: x range
            int jj; const double dx = overlap(0, cj, jj);
            if (!(dx > 0)) continue;
            if (NDIM == 2) unsafeAtomicAdd(out_map + (int64_t)ii * N + jj, dx * dy * val);
            else {
                for (int ck = lo[2]; ck < hi[2]; ++ck) {
                    int kk; const double dz = overlap(2, ck, kk);
                    if (!(dz > 0)) continue;
                    unsafeAtomicAdd(out_map + ((int64_t)ii * N + jj) * N + kk, dx * dy * dz * val);
                }
            }
        }
    }
}

}  // namespace bfg
