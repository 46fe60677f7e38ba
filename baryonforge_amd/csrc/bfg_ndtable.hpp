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
