// bfg_snapshot.hpp -- BaryonifySnapshot (BaryonForge/Runners/SnapshotRunner.py:176-275) on the GPU.
//
// The reference finds, halo by halo, the particles inside R_q = min(eps R / a, L / 2) with a periodic scipy KDTree
// (:99, :225 / :240), reads the radial displacement off the model's table and accumulates offset * unit-vector on
// every particle found (:232 / :249); at the end positions are shifted and wrapped into the box once (:260-273).
//
// Here the search is turned around (particles outnumber halos by 10^3): a coarse periodic cell grid holds, per cell,
// the list of halos whose sphere can reach the cell (built halo by halo: count -> scan -> fill over each sphere's
// bounding box; spheres spanning more than kSnapBigCells cells go to one global list instead).  Then ONE pass over the
// particles, one thread per particle in whatever order they come: look up the particle's cell, test its handful of
// candidate halos (+ the global list), read the displacement off the halo's blended radial row (hrow), sum the
// offsets in registers, shift, wrap, write.  No particle sort, no atomics on particle data, one read and one write of
// every particle.
// Included by bfg_mi355.hip after DevTable / massdef_radius are defined.
#pragma once

namespace bfg {

constexpr int kSnapBigCells = 4096;

struct __align__(16) SnapHalo {      // per-halo constants (written by snap_halo_kernel)
    double x, y, z, rq;              // centre [comoving Mpc], query radius
    double xcut;                     // model.epsilon_max * R_model_com: no displacement at or beyond it
    double lnshift;                  // ln(R_model_com) for Rdelta_sampling tables, else 0
    int32_t flags, pad;              // HF_OOB: (z, M, extras) outside the table hull -> contributes nothing
    double pad2;
};

struct __align__(16) SnapCand {      // one (cell, halo) entry of the overlap lists
    double x, y, z, rq;
    int32_t halo, pad[3];
};

struct SnapParams {
    int ndim;                        // 2 or 3
    int rdelta;
    int64_t n_part, n_halo;
    double L, a, eps_run, eps_model;
    bfg_massdef md_run, md_model;
    const double *part;              // [n_part][ndim]
    const double *halo;              // [n_halo][stride]: M, lnM (table coordinate), x, y, z, extras...
    int halo_stride, n_extra;
    DevTable tab;
    int ncell;                       // cells per dimension
    int64_t ncell_tot;
    int32_t *cell_count;             // [ncell_tot] (count, then fill cursor)
    int32_t *cell_start;             // [ncell_tot + 1]
    SnapCand *cand;                  // [cand_cap] candidates grouped by cell: test data inline, no second indirection
    int64_t cand_cap;
    int32_t *big;                    // [0] = count, [1..] halos whose sphere spans too many cells (tested by every particle)
    SnapHalo *hs;                    // [n_halo]
    double *hrow;                    // [n_halo][tab.nr] blended radial rows
    double *out;                     // [n_part][ndim] displaced, wrapped coordinates
    bfg_stats *stats;
};

__device__ inline int snap_cell_of(double x, double inv_cell, int n)
{
    int i = (int)floor(x * inv_cell);
    return min(max(i, 0), n - 1);    // x in [0, L]; x == L lands in the last cell
}

// exclusive scan of n int32 in three phases (block sums -> scan of the sums -> add), 1024 elements per block
__global__ __launch_bounds__(256) void snap_scan_block_kernel(int64_t n, const int32_t *in, int32_t *out, int32_t *bsum)
{
    __shared__ int32_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int incl = s;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    int run = woff + incl - s;
    for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 255) bsum[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(1024) void snap_scan_sums_kernel(int nb, int32_t *bsum, int32_t *total)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < nb) ? bsum[i] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int c = carry;
        if (i < nb) bsum[i] = c + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void snap_scan_add_kernel(int64_t n, int32_t *out, const int32_t *bsum, int32_t *count,
                                                            const int32_t *total)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { out[i] += bsum[i >> 10]; count[i] = 0; }      // count becomes the fill cursor
    if (i == 0) out[n] = *total;
}

// per halo: radii, table cell in the outer dimensions, full blended radial row
__global__ __launch_bounds__(64) void snap_halo_kernel(const SnapParams P)
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
        const double R = massdef_radius(P.md_run, M, P.a);                          // physical Mpc (:222)
        double rq = P.eps_run * R / P.a;                                            // comoving (:223)
        rq = fmin(fmax(rq, 0.0), 0.5 * P.L);                                        // :224 (NaN -> NaN: finds nothing)
        const double Rm = massdef_radius(P.md_model, M, P.a) / P.a;                 // BaryonCorrection.py:399
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
            atomicAdd((unsigned long long *)&P.stats->halos_out_of_table, 1ull);
            atomicOr(&P.stats->warn_mask, warn);
        }
        SnapHalo h;
        h.x = c[2]; h.y = c[3]; h.z = (P.ndim == 3) ? c[4] : 0.0; h.rq = rq;
        h.xcut = P.eps_model * Rm;
        h.lnshift = P.rdelta ? log(Rm) : 0.0;
        h.flags = oob ? HF_OOB : 0; h.pad = 0; h.pad2 = 0.0;
        P.hs[j] = h;
    }
    __syncthreads();
    const int ncorner = 1 << T.nouter;
    for (int i = lane; i < T.nr; i += 64) {
        double b = 0.0;
        for (int cc = 0; cc < ncorner; ++cc) b = fma(T.values[s_off[cc] + i], s_w[cc], b);
        P.hrow[j * T.nr + i] = s_oob ? nan("") : b;
    }
}

// cells of the (periodic) grid a halo's sphere can reach: count pass (fill = false) / list fill (fill = true);
// spheres spanning more than kSnapBigCells cells are appended to the global list in the count pass
template <int NDIM>
__global__ __launch_bounds__(256) void snap_overlap_kernel(const SnapParams P, int fill)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.n_halo) return;
    const SnapHalo h = P.hs[j];
    if ((h.flags & HF_OOB) || !(h.rq > 0.0)) return;           // NaN displacement everywhere -> 0 (:231 / :248)
    const int n = P.ncell;
    const double cell = P.L / (double)n, inv_cell = (double)n / P.L;
    const double hc[3] = {h.x, h.y, h.z};
    int lo[3] = {0, 0, 0}, cnt[3] = {1, 1, 1};
    for (int k = 0; k < NDIM; ++k) {
        // 1e-9 cells of slack: a particle's own cell index is floor(x / cell) of its wrapped coordinate
        const int a = (int)floor((hc[k] - h.rq) * inv_cell - 1e-9), b = (int)floor((hc[k] + h.rq) * inv_cell + 1e-9);
        lo[k] = a; cnt[k] = min(b - a + 1, n);                 // at most every cell once
    }
    const int64_t nbox = (int64_t)cnt[0] * cnt[1] * cnt[2];
    if (nbox > kSnapBigCells) {
        if (!fill) P.big[1 + atomicAdd(&P.big[0], 1)] = (int32_t)j;
        return;
    }
    const double rq2 = h.rq * h.rq * (1.0 + 1e-12);
    for (int64_t ci = 0; ci < nbox; ++ci) {
        int64_t rem = ci, cid = 0;
        int ic[3];
        for (int k = NDIM - 1; k >= 0; --k) { ic[k] = lo[k] + (int)(rem % cnt[k]); rem /= cnt[k]; }
        double dmin2 = 0.0;
        for (int k = 0; k < NDIM; ++k) {
            const double c0 = (double)ic[k] * cell, c1 = c0 + cell;
            double gap = 0.0;
            if (cnt[k] < n - 1) {                              // (nearly) full wrap: another image of the cell may be closer
                if (hc[k] < c0) gap = c0 - hc[k]; else if (hc[k] > c1) gap = hc[k] - c1;
            }
            dmin2 += gap * gap;
            int w = ic[k] % n; if (w < 0) w += n;
            cid = cid * n + w;
        }
        if (dmin2 > rq2) continue;                             // the sphere cannot reach this cell
        if (!fill) atomicAdd(&P.cell_count[cid], 1);
        else {
            const int64_t pos = (int64_t)P.cell_start[cid] + atomicAdd(&P.cell_count[cid], 1);
            if (pos < P.cand_cap) {
                SnapCand e;
                e.x = h.x; e.y = h.y; e.z = h.z; e.rq = h.rq; e.halo = (int32_t)j; e.pad[0] = e.pad[1] = e.pad[2] = 0;
                P.cand[pos] = e;
            }
        }
    }
}

// one thread per particle: candidates of its cell (+ the global list) -> summed offset -> shift, wrap, write
template <int NDIM>
__global__ __launch_bounds__(256) void snap_particle_kernel(const SnapParams P)
{
    // grid-stride over the particles: a few thousand fat workgroups, so that the two statistics counters see a few
    // thousand same-address atomics instead of one per wavefront (2e6 of those cost 20 ms)
    unsigned long long hits = 0, n_oob = 0;
    for (int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ip < P.n_part; ip += (int64_t)gridDim.x * blockDim.x) {
        const DevTable &T = P.tab;
        const double L = P.L, halfL = 0.5 * P.L;
        const int n = P.ncell;
        const double inv_cell = (double)n / L;
        double p[3] = {0.0, 0.0, 0.0}, off[3] = {0.0, 0.0, 0.0};
        int64_t cid = 0;
        for (int k = 0; k < NDIM; ++k) { p[k] = P.part[ip * NDIM + k]; cid = cid * n + snap_cell_of(p[k], inv_cell, n); }
        const double r_lo = T.raxis[0], r_hi = T.raxis[T.nr - 1];
        // a candidate: periodic distance test (compute_distance / enforce_periodicity, :104-158; KDTree radius :225 / :240)
        auto test = [&](double hx_, double hy_, double hz_, double rq_, double *dd, double &d) -> bool {
            const double hc[3] = {hx_, hy_, hz_};
            double d2 = 0.0;
            for (int k = 0; k < NDIM; ++k) {
                double dx = p[k] - hc[k];
                dx = (dx > halfL) ? dx - L : dx;
                dx = (dx < -halfL) ? dx + L : dx;
                dd[k] = dx; d2 += dx * dx;
            }
            d = sqrt(d2);
            return d <= rq_;
        };
        // a hit: BaryonificationClass._readout (BaryonCorrection.py:331-419): linear table, NaN outside the hull, 0 at or
        // beyond epsilon_max * R; non-finite offsets contribute nothing (:231 / :248)
        auto apply = [&](int j, const double *dd, double d) {
            ++hits;
            const double rin = log(d) - P.hs[j].lnshift;
            if (!(rin >= r_lo) || !(rin <= r_hi)) { ++n_oob; return; }
            if (!(d < P.hs[j].xcut)) return;
            const double *row = P.hrow + (int64_t)j * T.nr;
            int i;
            if (T.r_uniform) {                                 // geomspace radial axis: the cell by arithmetic, then one fix-up
                i = min(max((int)((rin - T.r0) * T.inv_dr), 0), T.nr - 2);
                if (rin < T.raxis[i]) --i; else if (rin >= T.raxis[i + 1] && i < T.nr - 2) ++i;
            } else i = find_interval(T.raxis, T.nr, rin);
            const double f = (rin - T.raxis[i]) / (T.raxis[i + 1] - T.raxis[i]);
            const double val = row[i] * (1.0 - f) + row[i + 1] * f;
            if (!(fabs(val) < 1.0e300)) return;
            const double s = val / d;
            for (int k = 0; k < NDIM; ++k) off[k] += s * dd[k];
        };
        const int c0 = P.cell_start[cid], c1 = (int)min((int64_t)P.cell_start[cid + 1], P.cand_cap);
        for (int q = c0; q < c1; q += 2) {                     // two candidates per trip: their loads are independent
            const SnapCand ea = P.cand[q];
            const SnapCand eb = P.cand[min(q + 1, c1 - 1)];
            double da[3] = {0, 0, 0}, db[3] = {0, 0, 0}, ra, rb;
            const bool ha = test(ea.x, ea.y, ea.z, ea.rq, da, ra);
            const bool hb = (q + 1 < c1) && test(eb.x, eb.y, eb.z, eb.rq, db, rb);
            if (ha) apply(ea.halo, da, ra);
            if (hb) apply(eb.halo, db, rb);
        }
        const int nbig = P.big[0];
        for (int q = 0; q < nbig; ++q) {
            const int j = P.big[1 + q];
            const double4 hx = *reinterpret_cast<const double4 *>(&P.hs[j]);
            double dd[3] = {0, 0, 0}, d;
            if (test(hx.x, hx.y, hx.z, hx.w, dd, d)) apply(j, dd, d);
        }
        for (int k = 0; k < NDIM; ++k) {
            double v = p[k] + off[k];                          // :262-265
            v = (v > L) ? v - L : v;                           // :268-273
            v = (v < 0.0) ? v + L : v;
            P.out[ip * NDIM + k] = v;
        }
    }
    __shared__ unsigned long long s_red[2][4];
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o, 64); n_oob += __shfl_down(n_oob, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = hits; s_red[1][threadIdx.x >> 6] = n_oob; }
    __syncthreads();
    if (threadIdx.x == 0) {
        hits = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        n_oob = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        if (hits) atomicAdd((unsigned long long *)&P.stats->pixel_updates, hits);
        if (n_oob) {
            atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, n_oob);
            if (!P.rdelta) atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);       // BaryonCorrection.py:391-394
        }
    }
}

// Mass deposit of particles on a periodic N^ndim grid: mode 0 = nearest grid point with numpy.histogramdd's bin
// rule on edges linspace(0, L, N + 1) (ParticleSnapshot.make_map, io.py:629-677: right-open bins, the last one
// closed); mode 1 = cloud-in-cell on cell centres (i + 1/2) L / N with periodic wrap.  mass == nullptr: unit masses.
struct DepositParams {
    int ndim, mode, N;
    int64_t n_part;
    double L;
    const double *pos;               // [n_part][ndim]
    const double *mass;              // [n_part] or nullptr
    double *grid;                    // [N^ndim], C order (x slowest), accumulated into
};

__device__ inline int ngp_bin(double x, double step, int N, double L)
{
    // np.histogramdd: searchsorted(edges, x, 'right') - 1, with x == edges[-1] put into the last bin; outside -> dropped
    if (!(x >= 0.0) || !(x <= L)) return -1;
    int i = (int)floor(x / step);
    i = min(max(i, 0), N - 1);
    // edges as numpy.linspace builds them: i * step, the last one exactly L
    const double e0 = (double)i * step, e1 = (i + 1 == N) ? L : (double)(i + 1) * step;
    if (x < e0) --i; else if (x >= e1 && i + 1 < N) ++i;
    return i;
}

template <int NDIM>
__global__ __launch_bounds__(256) void deposit_kernel(const DepositParams P)
{
    const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ip >= P.n_part) return;
    const double m = P.mass ? P.mass[ip] : 1.0;
    const double step = P.L / (double)P.N;
    if (P.mode == 0) {
        int64_t c = 0;
        for (int k = 0; k < NDIM; ++k) {
            const int i = ngp_bin(P.pos[ip * NDIM + k], step, P.N, P.L);
            if (i < 0) return;
            c = c * P.N + i;
        }
        unsafeAtomicAdd(P.grid + c, m);
    } else {
        int i0[NDIM];
        double w1[NDIM];
        for (int k = 0; k < NDIM; ++k) {
            const double u = P.pos[ip * NDIM + k] / step - 0.5;     // in units of cells, relative to cell centres
            const double f = floor(u);
            w1[k] = u - f;
            int i = (int)f % P.N; if (i < 0) i += P.N;
            i0[k] = i;
        }
        for (int corner = 0; corner < (1 << NDIM); ++corner) {
            double w = m;
            int64_t c = 0;
            for (int k = 0; k < NDIM; ++k) {
                const int bit = (corner >> k) & 1;
                w *= bit ? w1[k] : 1.0 - w1[k];
                int i = i0[k] + bit; if (i >= P.N) i -= P.N;
                c = c * P.N + i;
            }
            if (w != 0.0) unsafeAtomicAdd(P.grid + c, w);
        }
    }
}

}  // namespace bfg
