// bfg_ndtable.hpp -- tables with MORE p_keys axes than the shell kernels read (BFG_MAX_EXTRA = 3), on the device.
//
// ParamTabulatedProfile / BaryonificationClass tables are N-dimensional in the reference (utils/Tabulate.py:497-650,
// Profiles/BaryonCorrection.py:211-227, :404-408): (z, M, r, p_1 ... p_n) with one scipy RegularGridInterpolator call per halo.
// A multilinear read-out factors: all the non-radial coordinates of a (halo, pixel) query are the HALO's, so the 2^(n+2) corner
// blend over (z, M, p_1 ... p_n) is done ONCE per halo -- nd_rows_kernel: one wavefront per halo, lanes over the radial nodes --
// and what is left per pixel is a linear interpolation along r of the halo's row (nd_read_kernel).  The pixels come from the disc
// enumeration of bfg_enum.hpp and go back through its scatter-add kernels, so the hot tile kernels are not touched; the host
// evaluates nothing (round 3 sent these tables through scipy, per halo, on the host).
// Corner order and products as in halo_row_kernel / scipy's _evaluate_linear: w = prod_k (bit_k ? y_k : 1 - y_k), corners in
// index order, NaN where the halo lies outside the hull of any axis (fill_value = nan).
#pragma once

namespace bfg {

constexpr int kNdMaxOuter = 12;      // z, M and up to 10 p_keys axes

struct NdTable {
    int nouter, nr;
    int oshape[kNdMaxOuter];
    int64_t ostride[kNdMaxOuter];    // in doubles; the radial axis is the fastest
    const double *oaxis[kNdMaxOuter];
    const double *raxis;
    const double *values;            // [z][M][p_1]...[p_n][r]
};

// stats (may be null): halos outside the hull of an axis are counted, and the z / M axes raise the range warnings the 3-D path raises
// (BaryonCorrection.py:382-394).  The corner weights and row offsets of a halo are formed ONCE per chunk of 256 corners by the
// wavefront's lanes together (LDS) instead of by every lane for itself: 2^(n+2) x n_outer products per halo, not per (halo, node).
__global__ __launch_bounds__(256) void nd_rows_kernel(const NdTable T, const double *__restrict__ cat, int64_t n_halo, int cat_stride,
                                                      double *__restrict__ rows, bfg_stats *stats)
{
    constexpr int kChunk = 256;
    __shared__ double s_y[4][kNdMaxOuter];
    __shared__ int32_t s_ci[4][kNdMaxOuter];
    __shared__ double s_w[4][kChunk];
    __shared__ int64_t s_off[4][kChunk];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int ncorner = 1 << T.nouter;
    for (int64_t j = (int64_t)blockIdx.x * 4 + grp; j < n_halo; j += (int64_t)gridDim.x * 4) {
        const double *c = cat + j * (int64_t)cat_stride;
        bool oob = false;
        if (lane < T.nouter) {
            const double a = 1.0 / (1.0 + c[1]);
            const double x = (lane == 0) ? log(1.0 / a) : (lane == 1) ? log(c[0]) : c[4 + (lane - 2)];   // Tabulate.py:308, :312, :620
            const double *g = T.oaxis[lane];
            const int n = T.oshape[lane];
            oob = !(x >= g[0]) || !(x <= g[n - 1]);
            const int i = find_interval(g, n, x);
            s_ci[grp][lane] = i;
            s_y[grp][lane] = (x - g[i]) / (g[i + 1] - g[i]);
        }
        const unsigned long long oob_mask = __ballot(oob);
        const bool any_oob = oob_mask != 0ull;
        if (any_oob && stats && lane == 0) {
            atomicAdd((unsigned long long *)&stats->halos_out_of_table, 1ull);
            const unsigned warn = ((oob_mask & 1ull) ? BFG_WARN_Z_RANGE : 0u) | ((oob_mask & 2ull) ? BFG_WARN_M_RANGE : 0u);
            if (warn) atomicOr(&stats->warn_mask, warn);
        }
        __builtin_amdgcn_wave_barrier();
        // two neighbouring nodes per lane (one 16-byte load per corner row: the table lives in the Infinity Cache for all but the
        // smallest tables, where 8-byte accesses run at 0.5-0.7 of the 16-byte rate), 128 nodes per pass
        typedef double nd_v2d __attribute__((ext_vector_type(2), aligned(8)));
        for (int r0 = 0; r0 < T.nr; r0 += 128) {
            const int ir = r0 + 2 * lane;
            const bool two = ir + 1 < T.nr, one = ir < T.nr;
            double acc0 = 0.0, acc1 = 0.0;
            if (any_oob) { acc0 = __builtin_nan(""); acc1 = acc0; }
            else {
                for (int c0 = 0; c0 < ncorner; c0 += kChunk) {
                    const int nc = min(kChunk, ncorner - c0);
                    for (int q = lane; q < nc; q += 64) {
                        const int cc = c0 + q;
                        double w = 1.0;
                        int64_t off = 0;
                        for (int k = 0; k < T.nouter; ++k) {
                            const int bit = (cc >> (T.nouter - 1 - k)) & 1;
                            const double y = s_y[grp][k];
                            w = w * (bit ? y : 1.0 - y);
                            off += (int64_t)(s_ci[grp][k] + bit) * T.ostride[k];
                        }
                        s_w[grp][q] = w; s_off[grp][q] = off;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (two) {
                        for (int q = 0; q < nc; ++q) {
                            const nd_v2d v = *reinterpret_cast<const nd_v2d *>(T.values + s_off[grp][q] + ir);
                            const double w = s_w[grp][q];
                            acc0 = fma(v.x, w, acc0); acc1 = fma(v.y, w, acc1);
                        }
                    } else if (one) {
                        for (int q = 0; q < nc; ++q) acc0 = fma(T.values[s_off[grp][q] + ir], s_w[grp][q], acc0);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (two) { nd_v2d o; o.x = acc0; o.y = acc1; *reinterpret_cast<nd_v2d *>(rows + j * (int64_t)T.nr + ir) = o; }
            else if (one) rows[j * (int64_t)T.nr + ir] = acc0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the rows again, halos grouped by table cell ---------------------------------------------------------------------------------
// nd_rows_kernel reads the 2^(n+2) corner rows of every halo by itself: 51 / 102 KB per halo at four / five extra axes, out of a
// table that only fits the Infinity Cache -- 2.8 / 7.1 ms per 1e6 halos, three to seven times the painting.  But a table has few
// cells (nodes per parameter axis: a handful) and a catalog many halos per cell, and the corner ROWS are the cell's, only the
// weights are the halo's.  So: nd_cell_kernel finds every halo's cell (and its interpolation weights per axis), a counting sort
// groups the halo indices by cell (the scan / fill kernels of bfg_snapshot.hpp), and nd_rows_blocked_kernel gives kNdBlockHalos
// consecutive halos of the sorted list to one wavefront: every corner row is loaded ONCE (two nodes per lane, as above) and added
// into the rows of all the unit's halos of that cell (see the kernel for how the weights get there).  Tables with more than
// kNdMaxCells cells keep nd_rows_kernel.
// The sort key is cell * R + (halo index mod R), R a power of two: R counters per cell, because a catalog crowds into a few
// (z, M) cells and the counting atomics of one address serialise (one counter per cell: 0.38 + 0.30 ms for the count and fill
// passes of 1e6 halos, the hottest cell holding 1e4 of them).
constexpr int kNdBlockHalos = 8;
constexpr int kNdBlockChunk = 128;           // corners per pass (32 groups of four: their offsets and high-axes weights in LDS)
constexpr int64_t kNdMaxCells = 1 << 22;     // (cells x R counters)

// per halo: key[j] = cell * R + j % R, cell = flattened cell index over the outer axes (-1: outside the hull of an axis -> a NaN row,
// written here, and the counters / warnings of nd_rows_kernel), y[j][k] = interpolation weight on axis k, count[key] += 1
__global__ __launch_bounds__(256) void nd_cell_kernel(const NdTable T, const double *__restrict__ cat, int64_t n_halo, int cat_stride,
                                                      int rshift, int32_t *__restrict__ key, int32_t *__restrict__ rank,
                                                      double *__restrict__ y, int32_t *__restrict__ count, double *__restrict__ rows,
                                                      bfg_stats *stats)
{
    // the outer axes in LDS while they fit (the bisections are chains of dependent loads: as in halo_prep_kernel)
    constexpr int kAxisLds = 1024;
    __shared__ double s_axis[kAxisLds];
    __shared__ int s_axis0[kNdMaxOuter + 1];
    if (threadIdx.x == 0) {
        int pos = 0;
        for (int k = 0; k < T.nouter; ++k) { s_axis0[k] = pos; pos += T.oshape[k]; }
        s_axis0[T.nouter] = pos;
    }
    __syncthreads();
    const bool axes_lds = s_axis0[T.nouter] <= kAxisLds;
    if (axes_lds)
        for (int k = 0; k < T.nouter; ++k)
            for (int i = threadIdx.x; i < T.oshape[k]; i += blockDim.x) s_axis[s_axis0[k] + i] = T.oaxis[k][i];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_halo) return;
    const double *c = cat + j * (int64_t)cat_stride;
    const double a = 1.0 / (1.0 + c[1]);
    int64_t id = 0;
    bool oob = false;
    unsigned warn = 0;
    for (int k = 0; k < T.nouter; ++k) {
        const double x = (k == 0) ? log(1.0 / a) : (k == 1) ? log(c[0]) : c[4 + (k - 2)];   // Tabulate.py:308, :312, :620
        const double *g = axes_lds ? s_axis + s_axis0[k] : T.oaxis[k];
        const int n = T.oshape[k];
        if (!(x >= g[0]) || !(x <= g[n - 1])) {
            oob = true;
            if (k == 0) warn |= BFG_WARN_Z_RANGE;
            if (k == 1) warn |= BFG_WARN_M_RANGE;
        }
        const int i = find_interval(g, n, x);
        y[j * T.nouter + k] = (x - g[i]) / (g[i + 1] - g[i]);
        id = id * (n - 1) + i;
    }
    if (oob) {
        key[j] = -1;
        if (stats) {
            atomicAdd((unsigned long long *)&stats->halos_out_of_table, 1ull);
            if (warn) atomicOr(&stats->warn_mask, warn);
        }
        for (int ir = 0; ir < T.nr; ++ir) rows[j * (int64_t)T.nr + ir] = __builtin_nan("");
    } else {
        const int32_t kk = (int32_t)((id << rshift) | (j & ((1 << rshift) - 1)));
        key[j] = kk;
        rank[j] = atomicAdd(&count[kk], 1);             // the halo's place among those of its key: the fill pass needs no second atomic
    }
}

// perm[start[key] + rank] = halo index (the rank is the value nd_cell_kernel's counting atomic returned)
__global__ __launch_bounds__(256) void nd_fill_kernel(int64_t n, const int32_t *__restrict__ key, const int32_t *__restrict__ rank,
                                                      const int32_t *__restrict__ start, int32_t *__restrict__ perm)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int k = key[j];
    if (k >= 0) perm[start[k] + rank[j]] = (int32_t)j;
}

__device__ inline double nd_uniform(double v)             // a wave-uniform value into scalar registers (an FMA takes one scalar operand)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// The weight of corner c of halo h is a product over the axes; with the LAST TWO axes split off, w[h][c] = whi[h][c >> 2] * wlo[h][c & 3]:
// the four wlo of each halo sit in scalar registers, the 2^n whi go through LDS (a quarter of the corners), and four consecutive
// corner rows -- one (c >> 2) -- are first combined with wlo, then added with whi: one LDS broadcast per sixteen FMAs instead of
// one per four, which was what bound this kernel (five extra axes: 1.5 -> 0.9 ms per 1e6 halos).  The sums are associated differently from
// nd_rows_kernel's (rows equal to ~1e-16 relative, not bit for bit).
__global__ __launch_bounds__(256) void nd_rows_blocked_kernel(const NdTable T, int rshift, const int32_t *__restrict__ key,
                                                              const int32_t *__restrict__ perm, const int32_t *__restrict__ n_sorted_ptr,
                                                              const double *__restrict__ y, double *__restrict__ rows)
{
    constexpr int kGroups = kNdBlockChunk / 4;                            // groups of four corners per pass
    __shared__ double s_y[4][kNdBlockHalos][kNdMaxOuter];
    __shared__ int32_t s_ci[4][kNdMaxOuter];
    __shared__ int32_t s_j[4][kNdBlockHalos];
    __shared__ int64_t s_off[4][kGroups];
    __shared__ __align__(16) double s_w[4][kGroups][kNdBlockHalos];
    typedef double nd_v2d __attribute__((ext_vector_type(2), aligned(8)));
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n_sorted = *n_sorted_ptr;
    const int nhi = T.nouter - 2;                                         // axes whose weights go into whi
    const int ngroup = 1 << nhi;
    const int64_t st2 = T.ostride[T.nouter - 2], st1 = T.ostride[T.nouter - 1];
    for (int64_t u = (int64_t)blockIdx.x * 4 + grp; u * kNdBlockHalos < n_sorted; u += (int64_t)gridDim.x * 4) {
        const int p0 = (int)(u * kNdBlockHalos), np = min(kNdBlockHalos, n_sorted - p0);
        int mycell = -1;
        if (lane < np) { const int jj = perm[p0 + lane]; s_j[grp][lane] = jj; mycell = key[jj] >> rshift; }
        __builtin_amdgcn_wave_barrier();
        for (int run0 = 0; run0 < np;) {                                   // runs of halos that share a cell (sorted: contiguous)
            const int cid = __shfl(mycell, run0, 64);
            const int m = __popcll(__ballot(lane >= run0 && lane < np && mycell == cid));
            if (lane == 0) {
                int rem = cid;
                for (int k = T.nouter - 1; k >= 0; --k) { const int nk = T.oshape[k] - 1; s_ci[grp][k] = rem % nk; rem /= nk; }
            }
            for (int t = lane; t < m * T.nouter; t += 64) {
                const int h = t / T.nouter, k = t - h * T.nouter;
                s_y[grp][h][k] = y[(int64_t)s_j[grp][run0 + h] * T.nouter + k];
            }
            __builtin_amdgcn_wave_barrier();
            // the last two axes: the four row offsets (the cell's) and the halos' four weights each, wave-uniform
            const int64_t base2 = (int64_t)s_ci[grp][T.nouter - 2] * st2 + (int64_t)s_ci[grp][T.nouter - 1] * st1;
            double wlo[kNdBlockHalos][4];
#pragma unroll
            for (int h = 0; h < kNdBlockHalos; ++h) {
                const bool on = h < m;
                const double ya = on ? s_y[grp][h][T.nouter - 2] : 0.0, yb = on ? s_y[grp][h][T.nouter - 1] : 0.0;
                const double a0 = on ? 1.0 - ya : 0.0, b0 = 1.0 - yb;
                wlo[h][0] = nd_uniform(a0 * b0); wlo[h][1] = nd_uniform(a0 * yb);
                wlo[h][2] = nd_uniform(ya * b0); wlo[h][3] = nd_uniform(ya * yb);
            }
            for (int r0 = 0; r0 < T.nr; r0 += 128) {
                const int ir = r0 + 2 * lane;
                const bool two = ir + 1 < T.nr, one = ir < T.nr;
                double acc0[kNdBlockHalos], acc1[kNdBlockHalos];
#pragma unroll
                for (int h = 0; h < kNdBlockHalos; ++h) { acc0[h] = 0.0; acc1[h] = 0.0; }
                for (int g0 = 0; g0 < ngroup; g0 += kGroups) {
                    const int ng = min(kGroups, ngroup - g0);
                    // this pass's groups: the row offset of each and its whi for every halo -- one (group, halo) product per lane and trip
                    // (one GROUP per lane left three quarters of the lanes idle at four extra axes: this set-up was as many VALU
                    // instructions as the blend below)
                    for (int q = lane; q < ng; q += 64) {
                        const int gg = g0 + q;
                        int64_t off = base2;
                        for (int k = 0; k < nhi; ++k) off += (int64_t)(s_ci[grp][k] + ((gg >> (nhi - 1 - k)) & 1)) * T.ostride[k];
                        s_off[grp][q] = off;
                    }
                    for (int t = lane; t < ng * kNdBlockHalos; t += 64) {
                        const int q = t / kNdBlockHalos, h = t - q * kNdBlockHalos, gg = g0 + q;
                        double w = 0.0;
                        if (h < m) {
                            w = 1.0;
                            for (int k = 0; k < nhi; ++k) {
                                const double yy = s_y[grp][h][k];
                                w = w * (((gg >> (nhi - 1 - k)) & 1) ? yy : 1.0 - yy);
                            }
                        }
                        s_w[grp][q][h] = w;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (one) {
                        for (int q = 0; q < ng; ++q) {
                            const double *r00 = T.values + s_off[grp][q] + ir;
                            nd_v2d v0, v1, v2, v3;
                            if (two) {
                                v0 = *reinterpret_cast<const nd_v2d *>(r00); v1 = *reinterpret_cast<const nd_v2d *>(r00 + st1);
                                v2 = *reinterpret_cast<const nd_v2d *>(r00 + st2); v3 = *reinterpret_cast<const nd_v2d *>(r00 + st2 + st1);
                            } else {
                                v0.x = r00[0]; v1.x = r00[st1]; v2.x = r00[st2]; v3.x = r00[st2 + st1];
                                v0.y = 0.0; v1.y = 0.0; v2.y = 0.0; v3.y = 0.0;
                            }
                            const double4 wa = *reinterpret_cast<const double4 *>(&s_w[grp][q][0]);
                            const double4 wb = *reinterpret_cast<const double4 *>(&s_w[grp][q][4]);
                            const double whi[kNdBlockHalos] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                            for (int h = 0; h < kNdBlockHalos; ++h) {
                                double px = v0.x * wlo[h][0], py = v0.y * wlo[h][0];
                                px = fma(v1.x, wlo[h][1], px); py = fma(v1.y, wlo[h][1], py);
                                px = fma(v2.x, wlo[h][2], px); py = fma(v2.y, wlo[h][2], py);
                                px = fma(v3.x, wlo[h][3], px); py = fma(v3.y, wlo[h][3], py);
                                acc0[h] = fma(px, whi[h], acc0[h]); acc1[h] = fma(py, whi[h], acc1[h]);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int h = 0; h < kNdBlockHalos; ++h) {
                    if (h < m) {
                        double *row = rows + (int64_t)s_j[grp][run0 + h] * T.nr + ir;
                        if (two) { nd_v2d o; o.x = acc0[h]; o.y = acc1[h]; *reinterpret_cast<nd_v2d *>(row) = o; }
                        else if (one) row[0] = acc0[h];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            run0 += m;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One (halo, pixel) entry: the halo's row at ln(r_com) [- shift_j: Rdelta_sampling tables, BaryonCorrection.py:406-408], NaN outside the
// radial axis.  exp_values (paint, Tabulate.py:640-650 + HealpixRunner.py:473, :478): exp of it, non-finite -> 0, times scale_j
// (pixarea D_j^2).  Otherwise (displacement, BaryonCorrection.py:410-411): the value itself, 0 where r_com >= rcut_j; NaN is
// left for the offsets kernel, which zeroes non-finite offsets (:347).
__global__ __launch_bounds__(256) void nd_read_kernel(const double *__restrict__ rows, int nr, const double *__restrict__ raxis, int64_t n,
                                                      const int32_t *__restrict__ halo, const double *__restrict__ r_com,
                                                      const double *__restrict__ shift, const double *__restrict__ rcut,
                                                      const double *__restrict__ scale, int exp_values, double *__restrict__ out,
                                                      unsigned int *__restrict__ r_oob)
{
    unsigned int n_out = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = halo[e];
        const double rc = r_com[e];
        const double x = log(rc) - (shift ? shift[j] : 0.0);
        const bool inside = (x >= raxis[0]) && (x <= raxis[nr - 1]);
        double v = __builtin_nan("");
        if (inside) {
            const int i = find_interval(raxis, nr, x);
            const double f = (x - raxis[i]) / (raxis[i + 1] - raxis[i]);
            const double *row = rows + j * (int64_t)nr;
            const double b0 = row[i], b1 = row[i + 1];
            v = b0 * (1.0 - f) + b1 * f;
        } else n_out += 1;
        if (exp_values) {
            v = exp(v);
            if (!isfinite(v)) v = 0.0;
            if (scale) v *= scale[j];
        } else if (rcut && !(rc < rcut[j])) v = 0.0;
        out[e] = v;
    }
    if (r_oob && n_out) atomicAdd(r_oob, n_out);
}

}  // namespace bfg
