// bfg_tablebuild.hpp -- the displacement-table builder on the GPU.
//
// BaryonificationClass.setup_interpolator (BaryonForge/Profiles/BaryonCorrection.py:225-304) with
// get_masses (:669-691 for Sigma(r), :552-575 for rho(r)) inlined, for all (z, params, M) rows of a table at once:
//   density on the integration grid  -> clip negatives -> integrand * dlnr -> scipy cumulative_simpson (+ first term)
//   -> PCHIP of (ln r_int, ln M_enc) over the usable points -> ln M(r) on the table radii              [DMO and DMB]
//   -> iterative monotonic mask of ln M_DMB (:243-274) -> PCHIP r(ln M_DMB), PCHIP ln M_DMO(ln r)
//   -> d(r) = exp(r_DMB(ln M_DMO(ln r))) - r, non-finite -> 0 (:285-288) -> optional resampling on r / R_delta (:293-295).
// One workgroup per row.  The data-parallel parts (logs, PCHIP slopes and coefficients, evaluations) run on all
// threads; the inherently sequential ones (the running sum of the Simpson sub-integrals, in numpy's order, and the
// iterative mask) run on thread 0 over arrays in LDS -- a few hundred to a few thousand steps per row.
// The third-party primitives are restated from scipy (the oracle for them): cumulative_simpson's
// equal-interval formula, PchipInterpolator's three-point harmonic-mean slopes with its one-sided edge rule, and
// PPoly's local power basis; np.interp for the resampling.
#pragma once

namespace bfg {

struct BuildParams {
    int n_rows, n_int, nr, geometry;     // geometry 2: 2 pi r^2 Sigma, 3: 4 pi r^3 rho
    const double *dens_dmo, *dens_dmb;   // [n_rows][n_int]
    const double *r_int, *lnr_int;       // [n_int]
    double dlnr;
    const double *r, *lnr;               // [nr]
    const double *rdelta;                // [n_rows] R_delta (comoving) or nullptr
    const double *rdelta_range;          // [nr] (Rdelta_sampling)
    double *out;                         // [n_rows][nr]
    int32_t *status;                     // [n_rows] BFG_BUILD_* bits
};

// status bits (include/bfg_mi355.h): BFG_BUILD_CONSTANT "nearly constant over radius" (> 30 mask iterations),
// BFG_BUILD_FEW "less than 5 datapoints are usable", BFG_BUILD_ZERO the displacement defaulted to 0 for this row,
// BFG_BUILD_ERROR a PCHIP had < 2 points or a non-increasing / non-finite abscissa (scipy raises ValueError there)

__device__ inline double pchip_sign(double v) { return (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : ((v == 0.0) ? 0.0 : v)); }

// scipy.interpolate.PchipInterpolator._edge_case
__device__ inline double pchip_edge(double h0, double h1, double m0, double m1)
{
    double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    const bool flip = pchip_sign(d) != pchip_sign(m0);
    const bool big = (pchip_sign(m0) != pchip_sign(m1)) && (fabs(d) > 3.0 * fabs(m0));
    if (flip) d = 0.0;
    else if (big) d = 3.0 * m0;
    return d;
}

// PCHIP through (x[0..m), y[0..m)) in LDS: slopes into dk, then the PPoly coefficients c0..c2 (c3 = y) of every interval.
// All threads of the block call this; returns false (uniformly) if scipy would refuse the input.
__device__ inline bool pchip_build(int m, const double *x, const double *y, double *dk, double *c0, double *c1, int *s_bad)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) *s_bad = (m < 2) ? 1 : 0;
    __syncthreads();
    if (m >= 2) {
        for (int k = tid; k < m; k += nt) {
            if (!(fabs(x[k]) <= 1.797e308) || !(fabs(y[k]) <= 1.797e308)) *s_bad = 1;     // scipy: finite check
            if (k + 1 < m && !(x[k + 1] > x[k])) *s_bad = 1;                               // strictly increasing
        }
    }
    __syncthreads();
    if (*s_bad) return false;
    for (int k = tid; k < m; k += nt) {
        double d;
        if (m == 2) d = (y[1] - y[0]) / (x[1] - x[0]);
        else if (k == 0) {
            const double h0 = x[1] - x[0], h1 = x[2] - x[1];
            d = pchip_edge(h0, h1, (y[1] - y[0]) / h0, (y[2] - y[1]) / h1);
        } else if (k == m - 1) {
            const double h0 = x[m - 1] - x[m - 2], h1 = x[m - 2] - x[m - 3];
            d = pchip_edge(h0, h1, (y[m - 1] - y[m - 2]) / h0, (y[m - 2] - y[m - 3]) / h1);
        } else {
            const double hl = x[k] - x[k - 1], hr = x[k + 1] - x[k];
            const double ml = (y[k] - y[k - 1]) / hl, mr = (y[k + 1] - y[k]) / hr;
            const bool flat = (pchip_sign(mr) != pchip_sign(ml)) || (mr == 0.0) || (ml == 0.0);
            const double w1 = 2.0 * hr + hl, w2 = hr + 2.0 * hl;
            const double whmean = (w1 / ml + w2 / mr) / (w1 + w2);
            d = flat ? 0.0 : 1.0 / whmean;
        }
        dk[k] = d;
    }
    __syncthreads();
    for (int k = tid; k + 1 < m; k += nt) {            // CubicHermiteSpline.__init__
        const double dx = x[k + 1] - x[k];
        const double slope = (y[k + 1] - y[k]) / dx;
        const double t = (dk[k] + dk[k + 1] - 2.0 * slope) / dx;
        c0[k] = t / dx;
        c1[k] = (slope - dk[k]) / dx - t;
    }
    __syncthreads();
    return true;
}

// PPoly evaluation, extrapolate = False: NaN outside [x[0], x[m-1]] and for NaN arguments
__device__ inline double pchip_eval(int m, const double *x, const double *y, const double *dk, const double *c0,
                                    const double *c1, double v)
{
    if (!(v >= x[0]) || !(v <= x[m - 1])) return nan("");
    int lo = 0, hi = m - 1;                            // interval i with x[i] <= v < x[i+1]; the last one closed
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x[mid] <= v) lo = mid; else hi = mid; }
    const double s = v - x[lo];
    return ((c0[lo] * s + c1[lo]) * s + dk[lo]) * s + y[lo];
}

// LDS layout (doubles): xa[n_int] ya[n_int] da[n_int] ca[n_int] cb[n_int] | lnM[2][nr] | xb[nr] yb[nr] db[nr] cc[nr] cd[nr]
//                       xc[nr] yc[nr] dc[nr] ce[nr] cf[nr] | off[nr]; then int flags
__host__ __device__ inline size_t build_lds_bytes(int n_int, int nr)
{
    return (size_t)(5 * n_int + 13 * nr) * sizeof(double) + 16 * sizeof(int);
}

__global__ __launch_bounds__(256) void table_build_kernel(const BuildParams P)
{
    extern __shared__ double smem_tb[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int row = blockIdx.x;
    const int n = P.n_int, nr = P.nr;
    double *xa = smem_tb, *ya = xa + n, *da = ya + n, *ca = da + n, *cb = ca + n;
    double *lnM = cb + n;                              // [2][nr]: ln M_DMO, ln M_DMB on the table radii
    double *xb = lnM + 2 * nr, *yb = xb + nr, *db = yb + nr, *cc = db + nr, *cd = cc + nr;
    double *xc = cd + nr, *yc = xc + nr, *dc = yc + nr, *ce = dc + nr, *cf = ce + nr;
    double *off = cf + nr;
    int *flags = reinterpret_cast<int *>(off + nr);    // [0] m, [1] bad, [2] status, [3] mb, [4] mc
    if (tid == 0) flags[2] = 0;
    const double geo = (P.geometry == 2) ? 2.0 * 3.141592653589793 : 4.0 * 3.141592653589793;

    for (int which = 0; which < 2; ++which) {
        const double *dens = (which == 0 ? P.dens_dmo : P.dens_dmb) + (size_t)row * n;
        // integrand (get_masses :679-683): where(dens < 0, 0, dens) * 2 pi r^2 [4 pi r^3] * dlnr   -> ya; clipped density -> da
        for (int k = tid; k < n; k += nt) {
            double s = dens[k];
            s = (s < 0.0) ? 0.0 : s;
            const double rr = P.r_int[k];
            const double g = (P.geometry == 2) ? geo * (rr * rr) : geo * (rr * rr * rr);
            da[k] = s;
            ya[k] = g * s * P.dlnr;
        }
        __syncthreads();
        // scipy cumulative_simpson, equal intervals (dx = 1): sub-integral of [x_k, x_k+1]
        for (int k = tid; k < n - 1; k += nt) {
            double v;
            if ((k & 1) == 0 && k < n - 2) v = 1.0 / 3.0 * (5.0 * ya[k] / 4.0 + 2.0 * ya[k + 1] - ya[k + 2] / 4.0);
            else v = 1.0 / 3.0 * (5.0 * ya[k + 1] / 4.0 + 2.0 * ya[k] - ya[k - 1] / 4.0);
            ca[k] = v;
        }
        __syncthreads();
        if (tid == 0) {                                // running sum in numpy's order; compaction of the usable points
            const double first = ya[0];
            double run = 0.0;
            int m = 0;
            for (int k = 0; k < n; ++k) {
                const double Menc = run + first;       // cumulative_simpson(..., initial = 0) + intgd[:, [0]]
                if (k < n - 1) run += ca[k];
                if (da[k] > 0.0 && fabs(Menc) <= 1.797e308) { cb[m] = Menc; xa[m] = P.lnr_int[k]; ++m; }
            }
            flags[0] = m;
        }
        __syncthreads();
        const int m = flags[0];
        for (int k = tid; k < m; k += nt) ya[k] = log(cb[k]);
        __syncthreads();
        const bool ok = pchip_build(m, xa, ya, da, ca, cb, &flags[1]);
        for (int i = tid; i < nr; i += nt) {
            // M_f = exp(pchip(ln r)); the caller takes its logarithm again (:238-239)
            lnM[which * nr + i] = ok ? log(exp(pchip_eval(m, xa, ya, da, ca, cb, P.lnr[i]))) : nan("");
        }
        if (!ok && tid == 0) flags[2] |= BFG_BUILD_ERROR;
        __syncthreads();
    }

    const double *lnO = lnM, *lnB = lnM + nr;
    // iterative monotonic mask of ln M_DMB (:243-274); keep[] lives in off[] as 0/1 for now
    if (tid == 0) {
        int nkeep = nr;
        for (int i = 0; i < nr; ++i) off[i] = 1.0;
        double min_diff = -INFINITY;
        int iterate = 0;
        while (min_diff < 1e-5 && nkeep > 5) {
            double prev = 0.0;                         // np.diff(..., prepend = 0)
            for (int i = 0; i < nr; ++i) {
                if (off[i] == 0.0) continue;
                const double cur = lnB[i];
                const bool rise = (cur - prev) > 1e-5;
                const bool differs = (fabs(cur - lnO[i]) > 1e-6) || (lnO[i] != lnO[i]);
                const bool fin = fabs(cur) <= 1.797e308;
                prev = cur;                            // the difference is taken over the points kept BEFORE this round
                if (!(rise && differs && fin)) off[i] = 0.0;
            }
            off[0] = 1.0;
            ++iterate;
            nkeep = 0;
            for (int i = 0; i < nr; ++i) nkeep += (off[i] != 0.0);
            if (iterate > 30) {
                for (int i = 0; i < nr; ++i) off[i] = 0.0;
                nkeep = 0;
                flags[2] |= BFG_BUILD_CONSTANT;
                break;
            }
            if (nkeep < 5) { flags[2] |= BFG_BUILD_FEW; break; }
            min_diff = INFINITY;
            bool first = true;
            prev = 0.0;
            for (int i = 0; i < nr; ++i) {
                if (off[i] == 0.0) continue;
                if (!first) { const double dlt = lnB[i] - prev; min_diff = (dlt < min_diff || dlt != dlt) ? dlt : min_diff; }
                prev = lnB[i]; first = false;
            }
        }
        // compaction: PCHIP A  x = ln M_DMB[keep], y = ln r[keep];  PCHIP B  x = ln r[dmo_ok], y = ln M_DMO[dmo_ok] (:279-283)
        int mb = 0, mc = 0;
        if (nkeep > 5) {
            double prevO = 0.0;
            for (int i = 0; i < nr; ++i) {
                if (off[i] != 0.0) { xb[mb] = lnB[i]; yb[mb] = P.lnr[i]; ++mb; }
                const bool rise = (lnO[i] - prevO) > 1e-5;
                prevO = lnO[i];
                const bool differs = (fabs(lnB[i] - lnO[i]) > 1e-6) || (lnB[i] != lnB[i]);
                if (rise && differs && fabs(lnO[i]) <= 1.797e308) { xc[mc] = P.lnr[i]; yc[mc] = lnO[i]; ++mc; }
            }
        } else flags[2] |= BFG_BUILD_ZERO;
        flags[3] = mb; flags[4] = mc;
    }
    __syncthreads();
    const int mb = flags[3], mc = flags[4];
    bool built = (flags[2] & BFG_BUILD_ZERO) == 0 && (flags[2] & BFG_BUILD_ERROR) == 0;
    if (built) {
        const bool okb = pchip_build(mb, xb, yb, db, cc, cd, &flags[1]);
        const bool okc = pchip_build(mc, xc, yc, dc, ce, cf, &flags[1]);
        if (!(okb && okc)) { built = false; if (tid == 0) flags[2] |= BFG_BUILD_ERROR; }
    }
    __syncthreads();
    for (int i = tid; i < nr; i += nt) {
        double o = 0.0;
        if (built) {
            const double lnm = pchip_eval(mc, xc, yc, dc, ce, cf, P.lnr[i]);
            o = exp(pchip_eval(mb, xb, yb, db, cc, cd, lnm)) - P.r[i];
            if (!(fabs(o) <= 1.797e308)) o = 0.0;      // np.where(np.isfinite(offset), offset, 0)
        }
        off[i] = o;
    }
    __syncthreads();
    for (int i = tid; i < nr; i += nt) {
        double o = off[i];
        if (P.rdelta && built) {                       // np.interp(rdelta_range, r / Rdelta, offset) (:293-295)
            const double Rd = P.rdelta[row];
            const double v = P.rdelta_range[i];
            const double x0 = P.r[0] / Rd, x1 = P.r[nr - 1] / Rd;
            if (v != v) o = nan("");
            else if (v <= x0) o = off[0];
            else if (v >= x1) o = off[nr - 1];
            else {
                int lo = 0, hi = nr - 1;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.r[mid] / Rd <= v) lo = mid; else hi = mid; }
                const double xl = P.r[lo] / Rd, xh = P.r[lo + 1] / Rd;
                const double slope = (off[lo + 1] - off[lo]) / (xh - xl);
                o = slope * (v - xl) + off[lo];
            }
        }
        P.out[(size_t)row * nr + i] = o;
    }
    if (tid == 0) P.status[row] = flags[2];
}

}  // namespace bfg
