// bfg_tile.hpp -- sky-tile privatised variant of the paint kernel (BFG_VARIANT_TILE_LDS).
//
// Why: measured on MI355X (profiles/r01_atomic_microbench.txt) global f64 atomics top out at
// 1.0-1.3e11 adds/s for the 8-16 pixel ring segments a disc produces, i.e. >= 2 ms for the
// 2.8e8 pixel-updates of the headline workload, while LDS f64 atomics (ds_add_f64) run at
// 2.0e12 adds/s.  So the map is cut into sky tiles (kTileRings consecutive rings x one phi
// sector, <= kTileWidth pixels per ring row); every (halo, tile) overlap is binned on the device;
// one workgroup per tile accumulates all its halos into an LDS copy of the tile and writes the
// tile back with plain coalesced stores (each pixel belongs to exactly one tile: no global
// atomics on this path).
//
// Work decomposition inside a tile workgroup (512 threads = 8 wavefronts), per chunk of pairs:
//   stage a  one thread per (halo, tile) pair  : halo record -> ring range inside the band
//   stage b  one thread per (pair, ring)       : query_disc ring window, clipped to the sector
//                                                -> 64-byte segment record in LDS
//   stage c  one lane per segment, 64 segments per wave batch: wave prefix scan of the pixel
//            counts, then a flattened, fully populated loop over the pixels of the batch:
//            sin^2(dphi/2) polynomial -> chord^2 = A + B sin^2 -> table-driven ln -> cell and
//            fraction on the uniform ln r axis -> the halo's pre-blended profile row (value,
//            forward difference; built once per halo by halo_row_kernel) -> table-driven exp
//            -> ds_add_f64 into the tile.
// All arithmetic is float64; the ln / exp kernels are table + short polynomial
// (|rel err| < 1e-12), not libm calls.
#pragma once
#include "bfg_device.hpp"

namespace bfg {

// ---- table-driven ln / exp, series sin^2 (float64) ------------------------------------------------
// Accuracy budget: the painted value is exp(B_i + f (B_{i+1} - B_i)), f = frac((ln x) * t_m + t_c); the parity
// bar is 1e-5 relative.  These kernels are built for <= 1e-9 relative on the painted value:
//   ln : x = 2^e m, m in [1, 2); c = 1 + (idx + 0.5)/128 from the top 7 mantissa bits;
//        ln x = e ln2 + ln c + log1p(m/c - 1), |m/c - 1| < 2^-8, series to r^3 (abs err < 6e-11)
//   exp: L = k ln2/64 + r, |r| <= ln2/128; exp L = 2^(k>>6) 2^((k&63)/64) (1 + r + r^2/2 + r^3/6) (rel err < 4e-11)
//   sin^2(h): series to h^8 for h^2 <= 0.04 (rel err < 4e-10); wider angles (polar caps only) go through
//        h/16 and four angle doublings sin^2(2a) = 4 sin^2(a) (1 - sin^2(a)).
__device__ inline double fast_log(double x, const double2 *__restrict__ tab)
{
    const int hi = __double2hiint(x);
    const int lo = __double2loint(x);
    const int e = (hi >> 20) - 1023;                 // positive finite normal x; anything else is filtered by the caller
    const int idx = (hi >> 13) & (kLogTab - 1);
    const double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, lo);
    const double2 t = tab[idx];                      // {1/c, ln c}
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 1.0 / 3.0, -0.5);
    p = fma(r, p, 1.0);
    return fma((double)e, 0.693147180559945309417232, fma(r, p, t.y));
}

// ln(x) + 1023 ln 2 (the exponent bias is left in; callers fold it into their additive constant)
constexpr double kLogBias = 1023.0 * 0.693147180559945309417232;
__device__ inline double fast_log_biased(double x, const double2 *__restrict__ tab)
{
    const int hi = __double2hiint(x);
    const int lo = __double2loint(x);
    const int eb = hi >> 20;                         // biased exponent of a positive finite normal x
    const int idx = (hi >> 13) & (kLogTab - 1);
    const double m = __hiloint2double((hi & 0x000FFFFF) | (0x3FF00000 & ~0x000FFFFF), lo);
    const double2 t = tab[idx];                      // {1/c, ln c}
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 1.0 / 3.0, -0.5);
    p = fma(r, p, 1.0);
    return fma((double)eb, 0.693147180559945309417232, fma(r, p, t.y));
}

__device__ inline double fast_exp(double L, const double *__restrict__ tab)
{
    // round L * 64 / ln 2 to an integer with the 1.5 * 2^52 trick: the integer sits in the low word (|L| < 709 here)
    const double km = fma(L, 92.332482616893656877476, 6755399441055744.0);
    const int k = __double2loint(km);
    const double kf = km - 6755399441055744.0;
    const double r = fma(kf, -0.010830424696249145459412, L);      // ln2 / 64
    double p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    return ldexp(tab[k & (kExpTab - 1)] * p, k >> 6);
}

constexpr double kSinSmall = 0.04;                   // h^2 limit of the short series
__device__ inline double sin_squared_small(double q2)
{
    double p = fma(q2, -1.0 / 315.0, 2.0 / 45.0);
    p = fma(q2, p, -1.0 / 3.0);
    p = fma(q2, p, 1.0);
    return q2 * p;
}

// sin^2(h) for |h| <= 3.2 without libm (h = dphi/2 lies in [-pi/2, pi/2]; large |h| only near the poles)
__device__ inline double sin_squared_wide(double h)
{
    const double q = 0.0625 * h;
    double s = sin_squared_small(q * q);
    s = 4.0 * s * (1.0 - s);
    s = 4.0 * s * (1.0 - s);
    s = 4.0 * s * (1.0 - s);
    return 4.0 * s * (1.0 - s);
}

__device__ inline double sin_squared(double h)
{
    const double h2 = h * h;
    return (h2 <= kSinSmall) ? sin_squared_small(h2) : sin_squared_wide(h);
}

// Per-halo record of the tile path (one 128-byte line per halo, written by halo_prep_kernel)
struct __align__(16) HaloTile {
    double st, ct, pphi, S;          // sin/cos of the halo colatitude, longitude, (D/a)^2
    double cosr, z0, xa, pixfac;     // query_disc constants, pixarea * D^2 (or 1)
    int32_t rfirst, rlast, irmin, irmax;
    int32_t win_lo, flags, ci0, ci1; // ci0 / ci1: the halo's cell on the z and M axes of the table (3-D tables)
    double spare[4];                 // [0] = ln(pixfac), [1] / [2] = the halo's weights on the z / M axes (3-D tables)
};
static_assert(sizeof(HaloTile) == 128, "HaloTile must be one 128-byte line");

// The halo's cell (index, weight) on non-radial table axis k: from its HaloTile line for 3-D tables (the SoA arrays are only
// written for halos the scatter kernel / the fill pass read: PrepParams::lazy_soa), from the SoA arrays for tables with extra
// p_keys axes.  Scalar selects on purpose: private arrays indexed by the axis number end up in scratch memory.
__device__ inline int halo_cell_index(const HaloTile *ht, const int32_t *cidx, int64_t cap, int nouter, int64_t j, int k)
{
    return (nouter == 2) ? (k == 0 ? ht[j].ci0 : ht[j].ci1) : cidx[k * cap + j];
}
__device__ inline double halo_cell_weight(const HaloTile *ht, const double *cw, int64_t cap, int nouter, int64_t j, int k)
{
    return (nouter == 2) ? (k == 0 ? ht[j].spare[1] : ht[j].spare[2]) : cw[k * cap + j];
}

// Nodes i, i + 1 of the blended row of halo j straight from the table, corners in index order (the arithmetic of halo_row_kernel)
// ONLY2: the caller knows the table is 3-D (the BLEND instantiation of the tile kernel): the general corner loop is not compiled in
template <bool ONLY2 = false>
__device__ inline void halo_row_pair(const DevTable &T, const HaloTile *ht, const int32_t *cidx, const double *cw, int64_t cap,
                                     int64_t j, int i, double &c0v, double &c1v)
{
    c0v = 0.0; c1v = 0.0;
    if (ONLY2 || T.nouter == 2) {
        const HaloTile &h = ht[j];
        const double y0 = h.spare[1], y1 = h.spare[2];
        const double *r0 = T.values + (int64_t)h.ci0 * T.ostride[0] + (int64_t)h.ci1 * T.ostride[1] + i;
        const double *r1 = r0 + T.ostride[1], *r2 = r0 + T.ostride[0], *r3 = r2 + T.ostride[1];
        const double a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1], a2 = r2[0], b2 = r2[1], a3 = r3[0], b3 = r3[1];
        const double w0 = (1.0 * (1.0 - y0)) * (1.0 - y1), w1 = (1.0 * (1.0 - y0)) * y1;
        const double w2 = (1.0 * y0) * (1.0 - y1), w3 = (1.0 * y0) * y1;
        c0v = fma(a3, w3, fma(a2, w2, fma(a1, w1, fma(a0, w0, 0.0))));
        c1v = fma(b3, w3, fma(b2, w2, fma(b1, w1, fma(b0, w0, 0.0))));
        return;
    }
    if constexpr (!ONLY2) {
        const int ncorner = 1 << T.nouter;
        for (int c = 0; c < ncorner; ++c) {
            double w = 1.0; int64_t off = j * (int64_t)T.hstride;
            for (int k = 0; k < T.nouter; ++k) {
                const int bit = (c >> (T.nouter - 1 - k)) & 1;
                const double y = cw[k * cap + j];
                w = w * (bit ? y : 1.0 - y);
                off += (int64_t)(cidx[k * cap + j] + bit) * T.ostride[k];
            }
            c0v = fma(T.values[off + i], w, c0v);
            c1v = fma(T.values[off + i + 1], w, c1v);
        }
    }
}

struct __align__(16) Seg {           // one ring segment of one halo inside one tile (48 bytes)
    int32_t excl;                    // offset of the segment's first pixel in the chunk's flattened pixel list
    int32_t abyte;                   // LDS byte offset of the accumulator of the segment's first pixel
    int32_t wbyte;                   // LDS byte offset of node -1 of the pair's row window (node i at wbyte + 8 (i + 1));
                                     // pair slot when the windows are not staged in LDS
    int32_t pk;                      // paint: win_lo + 1 ; baryonify: pair slot | ring row << 6 | (win_lo + 1) << 12
    double hstep, c0, Aq, Bq;        // k-th pixel: h = k hstep + c0 ;  r_com^2 = Aq + Bq sin^2(h)
};
static_assert(sizeof(Seg) == 48, "Seg must be 48 bytes");

// extra per-halo constants of the baryonify offsets path (written by halo_prep_kernel next to HaloTile)
struct __align__(16) HaloDisp {
    double cp0, sp0;                 // cos / sin of the halo longitude
    double a, D;                     // scale factor, angular diameter distance [phys Mpc]
    double xcut;                     // (model.epsilon_max * R_model_com)^2: displacement is zero at r_com^2 >= xcut
    double tshift;                   // -ln(R_model_com) / dlnr for Rdelta_sampling tables, else 0
    double pad[2];
};
static_assert(sizeof(HaloDisp) == 64, "HaloDisp must be 64 bytes");

struct __align__(16) PairInfo {      // per (halo, tile) pair of the current chunk, paint (32 bytes)
    double lnpf;                     // ln(pixarea D^2), already folded into the row window (used by the rare direct read-out)
    int64_t hoff;                    // index of the halo's row window in hwin
    int32_t win_lo, halo, ra, pad;
};

struct __align__(16) PairInfoDisp {  // ... baryonify (96 bytes)
    double cp0, sp0, st, ct;         // halo unit vector = (st cp0, st sp0, ct)
    double a, D, xcut, tshift;
    int64_t hoff;
    int32_t win_lo, halo, ra, pad;
    double a_over_D;                 // a / D
};
static_assert(sizeof(PairInfoDisp) == 96, "PairInfoDisp must be 96 bytes");

struct __align__(16) DeferredPixel { // a pixel whose table cell lies outside the staged row window (queued, see drain)
    int32_t halo, abyte;             // halo index, LDS byte offset of the pixel's accumulator
    double t;                        // cell coordinate on the radial axis
};

// A deferred pixel that is still queued when its work item ends: handed to tile_deferred_kernel (a follow-up launch)
// instead of being drained by the tile workgroup itself -- the drain is two dependent rounds of global loads on the
// critical path of every tile (6 % of the kernel at 1e6 halos, 33 % at 1e5: profiles/r01_v25_stage_cycles.txt).
constexpr int kDeferCap = 40;        // >= the largest QCAP
struct __align__(16) DeferredOut {
    int64_t pix;                     // RING pixel index
    double t;                        // cell coordinate on the radial axis
    int32_t halo, pad[3];
};
static_assert(sizeof(DeferredOut) == 32, "DeferredOut must be 32 bytes");

struct __align__(16) RingRow {       // one ring of the tile's band (computed once per work item)
    double z, sth, phistep, phioff;
    int32_t nr, k0, k1, rowoff;      // rowoff = row * TW - k0
    int64_t start, pad;              // RING index of the ring's first pixel (the write-back needs no ring arithmetic of its own)
};
static_assert(sizeof(RingRow) == 64, "RingRow must be 64 bytes");

struct TileParams {
    Hpx hpx;
    int64_t n_halo, cap;
    const HaloTile *ht;
    const HaloDisp *hd;              // baryonify only
    const int32_t *cidx;             // outer-cell indices / weights (slow path only)
    const double *cw;
    DevTable tab;
    TileGeom geo;
    const int32_t *tile_start;       // [ntiles+1]
    const int4 *work;                // [n_work][2] (tile, first pair, last pair + 1 [positions in pairs], shared), (band, sector, NS, -) -- see tile_scan_kernel
    const int32_t *n_work;
    const int32_t *pairs;            // halo ids grouped by tile
    const double *hwin;              // [n_halo][win_nodes] blended row values B_i, i = win_lo + e
    int win_nodes;
    int win_table;                   // 1: no per-halo row windows at all -- the pixel stage blends the halo's corner rows
                                     // straight from the (L2-resident) table; for finely sampled radial axes (see below)
    double *out;
    bfg_stats *stats;
    const double2 *logtab;           // [128] {1/c, ln c}
    const double *exptab;            // [64]  2^(j/64)
    const double *atantab;           // [72]  atan(k/64), k = 0 .. 64
    long long pair_cap;              // capacity of pairs[]; a larger total means the binning fell back to scatter
    int debug;                       // ablation switches for profiling (BFG_DEBUG env; 0 in production)
    int out_zero;                    // BFG_SHELL_OUT_IS_ZERO: the caller cleared `out`; tiles are stored, not read-modify-written
    int overwrite;                   // BFG_SHELL_OUT_OVERWRITE: `out` is uninitialised; every tile (also one without halos) is stored in full
    DeferredOut *defer;              // [grid][defer_cap_wg] pixels a workgroup's items left queued (paint): added by the workgroup itself
                                     // after its last item (defer_tail) or by tile_deferred_kernel; null: every item drains its own
    int32_t *defer_count;            // [grid] entries in the workgroup's slice (for tile_deferred_kernel)
    int defer_cap_wg, defer_tail;
    int blend;                       // 1: no pre-blended row windows in HBM -- stage b blends every pair's 32-node window from the four
                                     // corner rows of the halo's (z, M) cell, straight from the (L2-resident) table, into LDS
    int32_t *work_counter;           // persistent grid: the item counters of this launch ([n_counters], kCounterStride apart; see the kernel);
                                     // null: one item per workgroup
    int n_counters;
    // Sliced calls (bfg_*_sliced): this launch takes the work items [slice[0], slice[1]) only -- the tiles of one band range, written
    // by tile_scan_kernel -- so that the caller can start exchanging that part of the map while the next slice is painted.
    const int32_t *slice;            // null: the whole work list
    // Sliced calls run the left-over scatter kernel BEFORE the tile kernels (a slice must be final when its launch ends).  If there
    // are left-over halos (*accum_left > 0, or the binning gave up) the output was cleared and scattered into beforehand, and the tiles
    // are added to it instead of stored.
    const int32_t *accum_left;       // null: not a sliced call
};

// sectors of band b whose phi range can intersect the disc: [s_lo, s_lo + n) modulo NS
__device__ inline void band_sectors(const TileGeom &G, int b, double pphi, double dphi_bound, int &s_lo, int &n)
{
    const int NS = G.band_ns[b];
    const double half = dphi_bound + 2.0 * kTwoPi / (double)G.band_nrmin[b];   // two pixels of slack
    if (!(half < kPi) || NS == 1) { s_lo = 0; n = NS; return; }
    const double f = (double)NS * kInvTwoPi;
    const int a = (int)floor((pphi - half) * f);
    const int e = (int)floor((pphi + half) * f);
    n = e - a + 1;
    if (n >= NS) { s_lo = 0; n = NS; return; }
    s_lo = ((a % NS) + NS) % NS;
}

// counter[idx] += 1 for the lanes that execute this call together (it may sit in divergent code: __ballot only sees the
// active lanes); returns the lane's slot, i.e. the value an atomicAdd of its own would have returned in some order.
// Lanes with the same idx are merged into one atomic (up to 8 groups per call, the rest go one by one), and all group
// leaders issue theirs in the same instruction.  Catalogs that arrive sorted by sky position put neighbouring halos in
// neighbouring lanes, which then hit the same tile counter: unmerged, those same-address atomics made the binning passes
// 4-6x slower (tools/shard_scale.py, LAYOUT=contiguous).
__device__ inline int wave_merged_inc(int32_t *counter, int idx)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(1);
    unsigned long long mine = 1ull << lane;
    int leader = lane;
    bool merged = false;
    for (int round = 0; round < 8 && todo; ++round) {
        const int l0 = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(idx, l0, 64);
        const unsigned long long same = __ballot(idx == k0) & todo;
        if ((same >> lane) & 1ull) { mine = same; leader = l0; }
        todo &= ~same;
        merged = merged || (same & (same - 1)) != 0;
        if (round == 1 && !merged) break;          // two singleton groups in a row: an unsorted catalog, stop looking
    }
    int base = 0;
#ifndef BFG_ABLATE_BIN_ATOMIC          // (profiling build: what the binning costs without its returning atomics; wrong results)
    if (lane == leader) base = atomicAdd(&counter[idx], __popcll(mine));
#endif
    base = __shfl(base, leader, 64);
    return base + __popcll(mine & ((1ull << lane) - 1ull));
}

// The same in two halves, so that several returning atomics of a lane can be in flight at once (the count pass of the binning
// issues a halo's pairs four at a time: a halo of the headline catalog overlaps 4-5 tiles, and one round trip after the other was
// 0.04 ms of the 0.155 ms prep kernel): `issue` groups the lanes that hit the same counter and lets the group leaders send their
// atomics, `finish` -- the first use of the returned value -- turns it into the lane's slot.  valid = false: the lane has no pair
// in this round (it still takes part in the ballots and shuffles).
struct MergedInc { int base, leader; unsigned long long mine; };
__device__ inline MergedInc wave_merged_inc_issue(int32_t *counter, int idx, bool valid)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    MergedInc m;
    m.mine = 1ull << lane; m.leader = lane; m.base = 0;
    bool merged = false;
    for (int round = 0; round < 8 && todo; ++round) {
        const int l0 = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(idx, l0, 64);
        const unsigned long long same = __ballot(valid && idx == k0) & todo;
        if ((same >> lane) & 1ull) { m.mine = same; m.leader = l0; }
        todo &= ~same;
        merged = merged || (same & (same - 1)) != 0;
        if (round == 1 && !merged) break;          // two singleton groups in a row: an unsorted catalog, stop looking
    }
#ifndef BFG_ABLATE_BIN_ATOMIC
    if (valid && lane == m.leader) m.base = atomicAdd(&counter[idx], __popcll(m.mine));
#endif
    return m;
}
__device__ inline int wave_merged_inc_finish(const MergedInc &m)
{
    const int lane = threadIdx.x & 63;
    const int base = __shfl(m.base, m.leader, 64);
    return base + __popcll(m.mine & ((1ull << lane) - 1ull));
}

// Bin one halo into the tiles its disc's bounding box (ring band x longitude extent) overlaps.
// fill = false: count pass (runs inside halo_prep_kernel).  The atomic that counts a pair also gives the pair its rank in
// the tile; ranks below cap_direct are the pair's slot in the tile's fixed region of pairs[] and the halo id is stored
// there at once -- for a catalog spread over the sky that is every pair, and the fill pass (one more returning atomic and
// one more scattered store per pair) has nothing left to do.  Pairs that find the slots full set their bit in `mask`.
// Returns the halo's flags, with HF_SCATTER set if it overlaps too many tiles or lies outside the table hull -- such
// halos are left to the global-atomic scatter kernel.
// fill = true: second pass, only for halos with a non-zero mask: appends the masked pairs to the tiles' overflow lists.
// If the overflow lists (sized on the device by the scan) exceed the pair buffer, the fill pass flags EVERY halo
// HF_SCATTER and the tile kernel exits: the call degrades to the scatter kernel instead of overflowing.
__device__ inline int tile_bin_halo(const BinCtx &B, bool fill, int64_t j, int flags, int rfirst, int rlast,
                                    int irmin, int irmax, double ptheta, double pphi, double radius,
                                    unsigned long long &mask)
{
    if (flags & HF_SKIP) return flags;
    if (fill && (flags & HF_SCATTER)) return flags;
    bool to_scatter = (flags & HF_OOB) != 0;
    int b0 = 0, b1 = -1;
    if (rlast >= rfirst) { b0 = (rfirst - 1) / B.geo.tr; b1 = (rlast - 1) / B.geo.tr; }
    else if (B.mode == MODE_BARYONIFY) to_scatter = true;        // empty disc -> 4-neighbour fallback
    // longitude half-extent of the disc: asin(sin r / sin theta0), or everything if a pole is inside
    double dphi_bound = kPi;
    const bool pole_inside = (rfirst < irmin) || (rlast > irmax) || !(radius < kPi) || (ptheta - radius <= 0) ||
                             (ptheta + radius >= kPi);
    if (!pole_inside) {
        const double q = sin_range(radius) / sin_range(ptheta);
        // an UPPER bound of asin(q) is all that is needed (band_sectors adds two pixels of slack on top): the series to q^5 plus the
        // sum of all its remaining (positive) coefficients on q^7; beyond q = 0.7 (discs next to a pole) libm
        if (q < 0.7) { const double q2 = q * q; dphi_bound = q * (1.0 + q2 * (1.0 / 6.0 + q2 * (0.075 + q2 * 0.3292))); }
        else dphi_bound = (q < 1.0) ? asin(q) : kPi;
    }
    if (!fill) {
        int npairs = 0;
        for (int b = b0; b <= b1; ++b) { int s_lo, n; band_sectors(B.geo, b, pphi, dphi_bound, s_lo, n); npairs += n; }
        if (npairs > kMaxPairsPerHalo) { to_scatter = true; flags |= HF_SLOW; }   // a disc over more than 64 tiles
        if (to_scatter) return flags | HF_SCATTER;
    }
    const int64_t ovf_base = (int64_t)B.geo.ntiles * B.cap_direct;
    bool crowded = false;                                        // some tile of this halo holds more pairs than one plain work item takes
    int ipair = 0;                                               // the halo's pairs in enumeration order (< kMaxPairsPerHalo = 64)
    if (!fill) {
        // count pass: the halo's pairs four at a time -- four counting atomics in flight per lane, then their slots
        constexpr int kGroup = 4;
        int b = b0, i = 0, s_lo = 0, n = 0, NS = 1, t0 = 0;
        bool have = b0 <= b1;
        if (have) { band_sectors(B.geo, b, pphi, dphi_bound, s_lo, n); NS = B.geo.band_ns[b]; t0 = B.geo.band_tile0[b]; }
        while (__any(have)) {
            int tl[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                tl[k] = -1;
                if (have) {
                    int sct = s_lo + i; if (sct >= NS) sct -= NS;
                    tl[k] = t0 + sct;
                    if (++i >= n) {
                        if (++b > b1) have = false;
                        else { band_sectors(B.geo, b, pphi, dphi_bound, s_lo, n); NS = B.geo.band_ns[b]; t0 = B.geo.band_tile0[b]; i = 0; }
                    }
                }
            }
            MergedInc mk[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; ++k) mk[k] = wave_merged_inc_issue(B.tile_count, tl[k] < 0 ? 0 : tl[k], tl[k] >= 0);
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                const int pos = wave_merged_inc_finish(mk[k]);                  // the pair's rank in its tile
                if (tl[k] >= 0) {
                    if (pos < B.cap_direct) B.pairs[(int64_t)tl[k] * B.cap_direct + pos] = (int32_t)j;
                    else mask |= 1ull << ipair;
                    crowded = crowded || pos >= B.direct_limit;
                    ++ipair;
                }
            }
        }
        if (crowded) *B.needs_scan = 1;                          // same value from every writer
        return flags;
    }
    for (int b = b0; b <= b1; ++b) {
        int s_lo, n;
        band_sectors(B.geo, b, pphi, dphi_bound, s_lo, n);
        const int NS = B.geo.band_ns[b], t0 = B.geo.band_tile0[b];
        for (int i = 0; i < n; ++i, ++ipair) {
            int s = s_lo + i; if (s >= NS) s -= NS;
            const int tile = t0 + s;
            if (!fill) {
                const int pos = wave_merged_inc(B.tile_count, tile);            // the pair's rank in its tile
                if (pos < B.cap_direct) B.pairs[(int64_t)tile * B.cap_direct + pos] = (int32_t)j;
                else mask |= 1ull << ipair;
                crowded = crowded || pos >= B.direct_limit;
            } else if ((mask >> ipair) & 1ull) {
                const int pos = wave_merged_inc(B.tile_count, tile);            // cursor of the tile's overflow list
                B.pairs[ovf_base + B.tile_start[tile] + pos] = (int32_t)j;
            }
        }
    }
    if (crowded) *B.needs_scan = 1;                              // same value from every writer
    return flags;
}

struct FillParams {
    bfg_stats *stats;
    int overwrite, nacc;             // overwrite: clear the tiles the tile kernel will add to with atomics (shared_flag)
    const int32_t *shared_flag;
    double *out;
    Hpx hpx;
    int64_t n_halo, cap;
    const double *rec;
    int32_t *irec;
    HaloTile *ht;
    BinCtx bin;
    PrepParams prep;                 // lazy_soa: what halo_calc needs to rebuild a halo's SoA row
};

// Clear the tiles the tile kernel will add to with atomics (those cut into several work items: crowded catalogs, compact
// multi-GPU shards) -- overwrite mode only.  Tile t belongs to workgroup t mod gridDim, so the clearing is spread over the whole
// grid; a workgroup reads the flags of up to 256 of its tiles in one go, one per thread (a small grid walking its flags one by
// one took 78 us at 1e4 halos; 25 workgroups clearing 256 tiles each took 1 ms on a half-sky shard).  256 threads per workgroup.
__device__ inline void clear_shared_tiles(const TileGeom &G, const Hpx &hpx, const int32_t *shared_flag, double *out, int nacc)
{
    __shared__ unsigned long long s_mask[4];
    for (int base = 0; base < G.ntiles; base += 256 * (int)gridDim.x) {
        const int tt = base + (int)threadIdx.x * (int)gridDim.x + (int)blockIdx.x;
        const int f = (tt < G.ntiles) ? shared_flag[tt] : 0;
        const unsigned long long m = __ballot(f != 0);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_mask[threadIdx.x >> 6] = m;
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
          unsigned long long todo = s_mask[w];               // workgroup-uniform; zero for a catalog spread over the sky
          while (todo) {
            const int i = 64 * w + __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int t = base + i * (int)gridDim.x + (int)blockIdx.x;
            const int band = G.tile_band[t], sector = t - G.band_tile0[band], NS = G.band_ns[band];
            const int ring_lo = 1 + band * G.tr;
            for (int e = threadIdx.x; e < G.tr * G.tw; e += blockDim.x) {
                const int row = e / G.tw, col = e % G.tw;
                const int64_t ring = ring_lo + row;
                if (ring > 4 * hpx.nside - 1) continue;
                int64_t start, nr; bool shifted;
                ring_info_small(hpx, ring, start, nr, shifted);
                const int k0 = (int)(((int64_t)sector * nr) / NS), k1 = (int)(((int64_t)(sector + 1) * nr) / NS);
                if (k0 + col < k1) for (int c = 0; c < nacc; ++c) out[nacc * (start + k0 + col) + c] = 0.0;
            }
          }
        }
    }
}

__global__ __launch_bounds__(256) void tile_fill_kernel(const FillParams P)
{
    // no tile beyond its fixed slots (the count pass says so): no overflow lists to fill, no tile cut into shared items
    if (!*P.bin.needs_scan) return;
    if (P.overwrite) clear_shared_tiles(P.bin.geo, P.hpx, P.shared_flag, P.out, P.nacc);
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.n_halo) return;
    const int64_t cap = P.cap;
    if ((long long)P.bin.tile_start[P.bin.geo.ntiles] > P.bin.pair_cap) {        // overflow lists too long: all to scatter
        const int f0 = P.ht[j].flags, f = f0 | HF_SCATTER;
        if (P.prep.lazy_soa) {
            // the prep kernel wrote the SoA rows of the halos it knew would need them; now every halo does
            HaloCalc hc;
            halo_calc(P.prep, j, P.prep.spl_knots, [&](int k) -> const double * { return P.prep.tab.oaxis[k]; },
                      [&](int k, int i, double y) { P.prep.cidx[k * cap + j] = i; P.prep.cw[k * cap + j] = y; }, hc);
            halo_write_soa(P.prep, j, hc, f);
        } else P.irec[I_FLAGS * cap + j] = f;
        P.ht[j].flags = f;
        if (!(f0 & (HF_SCATTER | HF_SKIP))) { atomicAdd(&P.stats->halos_scatter_fallback, 1u); atomicAdd(&P.prep.left_n[kPlanFallback], 1); }
        return;
    }
    unsigned long long mask = P.bin.ovf_mask[j];
    if (mask == 0ull) return;                                                     // every pair found a slot in the count pass
    tile_bin_halo(P.bin, true, j, P.irec[I_FLAGS * cap + j], P.irec[I_RFIRST * cap + j], P.irec[I_RLAST * cap + j],
                  P.irec[I_IRMIN * cap + j], P.irec[I_IRMAX * cap + j], P.rec[F_PTHETA * cap + j],
                  P.rec[F_PPHI * cap + j], P.rec[F_RADIUS * cap + j], mask);
}

// Per tile: the first cap_direct pairs sit in the tile's fixed slots (written by the count pass), the rest go to an
// overflow list: exclusive scan of the overflow lengths into tile_start[ntiles+1] (tile_start[ntiles] = their total), then
// the work list of the tile kernel; single workgroup.
// A work item is two int4: (tile, first pair, last pair + 1 -- positions in pairs[], so the tile kernel needs no tile_start
// lookup on its start-up chain --, shared) and (band, sector, sectors in the band, -): a tile's slot region and its overflow list are separate items, and either is cut
// into equal slices of at most S pairs; items of one tile are accumulated by different workgroups (shared = 1: the
// write-back uses atomics).  S = max(256, pairs / 4096), so a full-sky catalog gives one item per tile while a catalog
// that crowds into part of the sky (an octant light cone, a compact multi-GPU shard: 1/8 of the tiles with 8x the pairs)
// still yields a few thousand items of similar size instead of 784 heavy ones on 512 workgroup slots.
constexpr int kWorkExtra = 4096;

// Sliced calls: the work list is cut at tile boundaries slices.tile[0] = 0 < tile[1] < ... < tile[n] = ntiles (whole bands); the scan
// kernel writes the item range of slice k to slices.range[2 k .. 2 k + 1], which the k-th launch of the tile kernel reads.
constexpr int kMaxSlices = 16;
struct SliceCuts {
    int n;                           // 0: not a sliced call
    int tile[kMaxSlices + 1];
    int32_t *range;                  // [2 n]
};

// Item counters of the persistent tile kernel: set 0 for a whole-list launch, set 1 + k for the launch of slice k; K counters per
// set, kCounterStride ints apart.  Counter c of a set hands out the LOCAL item indices c + K v, v = its value; the first
// first_dynamic (= 3 x grid) local indices are the workgroups' static items.
constexpr int kMaxCounters = 16, kCounterStride = 256;
constexpr size_t kCounterInts = (size_t)(1 + kMaxSlices) * kMaxCounters * kCounterStride;
__device__ inline void init_item_counter(int32_t *counters, int t, int first_dynamic, int K)
{
    const int set = t / K, c = t - set * K;
    counters[((size_t)set * kMaxCounters + c) * kCounterStride] = (first_dynamic - c + K - 1) / K;
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(const TileGeom geo, int cap_direct, int32_t *count, int32_t *start, int4 *work,
                                                         int32_t *n_work, int32_t *counters, int n_counters, int first_dynamic,
                                                         int overwrite, int32_t *shared_flag, const int32_t *needs_scan,
                                                         int32_t *next_count, const SliceCuts slices)
{
    // the other set of counters (ntiles + 2), for the next call
    if (blockIdx.x != 0) {
        const int t = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x;
        if (t < geo.ntiles) next_count[t] = 0;
        if (t < kTileTail) next_count[geo.ntiles + t] = 0;
    }
    // overwrite: the tile kernel initialises the map itself, so tiles without a single pair get an (empty) work item too,
    // and tiles cut into several items -- which add to the map with atomics -- are listed for tile_fill_kernel to clear first
    const int ntiles = geo.ntiles;
    // Fast path (no tile holds more than min(cap_direct, 512) pairs -- the count pass found out: needs_scan): one work item
    // per tile, item = tile, nothing to scan; blocks 1 .. write them in parallel and block 0 has nothing to do.  The
    // single-workgroup scan below (~20 us) is left to dense or crowded catalogs, where it is 1-2 % of the call.
    if (!*needs_scan) {
        if (blockIdx.x == 0) return;
        const int t = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x;
        if (t == 0) { *n_work = ntiles; start[ntiles] = 0; }
        if (t < (1 + slices.n) * n_counters) init_item_counter(counters, t, first_dynamic, n_counters);
        if (t < slices.n) {                                      // item = tile: the cuts are the item ranges
            slices.range[2 * t] = slices.tile[t]; slices.range[2 * t + 1] = slices.tile[t + 1];
        }
        if (t >= ntiles) return;
        const int n = count[t];
        count[t] = 0; start[t] = 0;
        if (shared_flag) shared_flag[t] = 0;
        const int band = geo.tile_band[t];
        work[2 * t] = make_int4(t, t * cap_direct, t * cap_direct + n, 0);
        work[2 * t + 1] = make_int4(band, t - geo.band_tile0[band], geo.band_ns[band], 0);
        return;
    }
    if (blockIdx.x != 0) return;
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry, carry2;
    // block-wide exclusive scan of 4 values per thread (4096 tiles per trip); returns the offset of the thread's first value
    auto scan4 = [&](const int v[4]) -> int {
        const int s = v[0] + v[1] + v[2] + v[3];
        int incl = s;
        for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if ((threadIdx.x & 63) >= d) incl += o; }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int c = carry;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + incl;
        __syncthreads();
        return c + woff + incl - s;
    };
    if (threadIdx.x == 0) { carry = 0; carry2 = 0; }
    __syncthreads();
    int mine = 0;                                       // pairs of this thread's tiles (all of them, for S)
    for (int base = 0; base < ntiles; base += 4096) {
        const int i0 = base + 4 * threadIdx.x;
        int v[4];
        for (int k = 0; k < 4; ++k) {
            const int n = (i0 + k < ntiles) ? count[i0 + k] : 0;
            mine += n;
            v[k] = max(n - cap_direct, 0);
        }
        int o = scan4(v);
        for (int k = 0; k < 4; ++k) if (i0 + k < ntiles) { start[i0 + k] = o; o += v[k]; }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&carry2, mine);
    const int total_ovf = carry;
    __syncthreads();                                   // every thread has read the total before carry is reused
    if (threadIdx.x == 0) { start[ntiles] = total_ovf; carry = 0; }
    __syncthreads();
    const int total = carry2;
    const int S = max(256, (total + kWorkExtra - 1) / kWorkExtra);
    const int64_t ovf_base = (int64_t)ntiles * cap_direct;
    for (int base = 0; base < ntiles; base += 4096) {
        const int i0 = base + 4 * threadIdx.x;
        int nd[4], no[4], v[4], gb[4], gs[4], gn[4];
        for (int k = 0; k < 4; ++k) gb[k] = (i0 + k < ntiles) ? geo.tile_band[i0 + k] : 0;     // all loads of a trip in flight together
        for (int k = 0; k < 4; ++k) { gs[k] = i0 + k - geo.band_tile0[gb[k]]; gn[k] = geo.band_ns[gb[k]]; }
        for (int k = 0; k < 4; ++k) {
            const int n = (i0 + k < ntiles) ? count[i0 + k] : 0;
            nd[k] = min(n, cap_direct); no[k] = n - nd[k];
            v[k] = (nd[k] + S - 1) / S + (no[k] + S - 1) / S;
            if (overwrite && v[k] == 0 && i0 + k < ntiles) v[k] = 1;
        }
        int o = scan4(v);
        for (int k = 0; k < 4; ++k) {
            if (i0 + k >= ntiles) continue;
            for (int q = 0; q < slices.n; ++q)                                               // the first item of a slice's first tile
                if (slices.tile[q] == i0 + k) {
                    slices.range[2 * q] = o;
                    if (q > 0) slices.range[2 * q - 1] = o;
                }
            count[i0 + k] = 0;                                                               // becomes the overflow cursor
            const int shared = v[k] > 1 ? 1 : 0;
            if (shared_flag) shared_flag[i0 + k] = shared;
            if (nd[k] + no[k] == 0 && overwrite) {                                           // a tile no halo touches: zeros
                work[2 * o] = make_int4(i0 + k, 0, 0, 0);
                work[2 * o + 1] = make_int4(gb[k], gs[k], gn[k], 0);
                ++o;
            }
            for (int part = 0; part < 2; ++part) {
                const int n = part ? no[k] : nd[k];
                const int nv = (n + S - 1) / S;
                const int slice = nv ? (n + nv - 1) / nv : 0;                                // equal slices
                const int a0 = part ? (int)(ovf_base + start[i0 + k]) : (i0 + k) * cap_direct;   // positions in pairs[]
                for (int m = 0; m < nv; ++m, ++o) {
                    work[2 * o] = make_int4(i0 + k, a0 + m * slice, a0 + min(n, (m + 1) * slice), shared);
                    work[2 * o + 1] = make_int4(gb[k], gs[k], gn[k], 0);     // what the tile kernel needs about the tile, in one load
                }
            }
        }
    }
    if ((int)threadIdx.x < (1 + slices.n) * n_counters) init_item_counter(counters, (int)threadIdx.x, first_dynamic, n_counters);
    if (threadIdx.x == 0) {
        *n_work = carry;                                         // (the first items of a workgroup are static)
        if (slices.n > 0) slices.range[2 * slices.n - 1] = carry;
    }
}

// What a later call needs to run the tile kernels AGAIN on the records and pair lists of this one (bfg_shell_args.flags &
// BFG_SHELL_REUSE_PLAN: another model tabulated on the same grid, painted over the same catalog): the item counters of the
// persistent tile kernel (consumed by every launch), the planning call's contributions to the statistics, and -- overwrite mode,
// crowded catalogs -- the tiles that several work items add to with atomics, cleared again.  One launch instead of halo_prep_kernel,
// tile_scan_kernel and tile_fill_kernel.
struct ReinitParams {
    TileGeom geo;
    Hpx hpx;
    int32_t *counters;
    int n_counters, first_dynamic, n_slices;
    bfg_stats *stats;
    const int32_t *tail;             // the planning call's counter set behind its tile counts: left_n, needs_scan, plan statistics
    int overwrite, nacc;
    const int32_t *shared_flag;
    double *out;
};
__global__ __launch_bounds__(256) void plan_reinit_kernel(const ReinitParams P)
{
    const int t = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (t < (1 + P.n_slices) * P.n_counters) init_item_counter(P.counters, t, P.first_dynamic, P.n_counters);
    if (t == 0) {
        if (P.tail[kPlanOob]) atomicAdd((unsigned long long *)&P.stats->halos_out_of_table, (unsigned long long)P.tail[kPlanOob]);
        if (P.tail[kPlanWarn]) atomicOr(&P.stats->warn_mask, (uint32_t)P.tail[kPlanWarn]);
        if (P.tail[kPlanFallback]) atomicAdd(&P.stats->halos_scatter_fallback, (uint32_t)P.tail[kPlanFallback]);
    }
    if (P.overwrite && P.tail[1]) clear_shared_tiles(P.geo, P.hpx, P.shared_flag, P.out, P.nacc);     // tail[1] = needs_scan
}

// Per-halo blended radial row: hwin[j][e] = B_i, i = win_lo_j + e, where
// B_i = sum over the 2^(ndim-1) corners of the halo's (z, M, extras) cell of w_c * T[c][i].
// One thread per (halo, node); built once per halo instead of once per (halo, tile) pair.
struct RowParams {
    int64_t n_halo, cap;
    const HaloTile *ht;
    const int32_t *cidx;
    const double *cw;
    DevTable tab;
    int win_nodes;
    double *hwin;
};

__global__ __launch_bounds__(256) void halo_row_kernel(const RowParams P)
{
    // block = (256 / W) halos x W nodes; the halo's corner (row offset, weight) pairs are formed once in LDS
    __shared__ double s_w[256 / 8][kMaxCorner];
    __shared__ int64_t s_off[256 / 8][kMaxCorner];
    __shared__ int s_winlo[256 / 8];
    __shared__ double s_add[256 / 8];
    const DevTable &T = P.tab;
    const int W = P.win_nodes;
    const int hpb = 256 / W;                                   // halos per block (W <= 256, W >= 8)
    const int hl = threadIdx.x / W, e = threadIdx.x - hl * W;
    const int64_t j = (int64_t)blockIdx.x * hpb + hl;
    const bool live = (hl < hpb) && (j < P.n_halo);
    const int ncorner = 1 << T.nouter;
    bool skip = true;
    if (live) {
        const int flags = P.ht[j].flags;
        skip = (flags & (HF_SKIP | HF_OOB)) != 0;
        if (e == 0) { s_winlo[hl] = P.ht[j].win_lo; s_add[hl] = T.log_values ? P.ht[j].spare[0] : 0.0; }
        for (int c = e; c < ncorner; c += W) {
            double w = 1.0;
            int64_t off = j * (int64_t)T.hstride;
            for (int k = 0; k < T.nouter; ++k) {
                const int bit = (c >> (T.nouter - 1 - k)) & 1;
                const double y = halo_cell_weight(P.ht, P.cw, P.cap, T.nouter, j, k);
                const int i = halo_cell_index(P.ht, P.cidx, P.cap, T.nouter, j, k);
                w = w * (bit ? y : 1.0 - y);
                off += (int64_t)(i + bit) * T.ostride[k];
            }
            s_w[hl][c] = w; s_off[hl][c] = off;
        }
    }
    __syncthreads();
    if (!live || skip) return;
    const int ir = s_winlo[hl] + e;
    double b0 = 0.0;
    for (int c = 0; c < ncorner; ++c) b0 = fma(T.values[s_off[hl][c] + ir], s_w[hl][c], b0);
    P.hwin[j * W + e] = b0 + s_add[hl];          // paint: ln(T) + ln(pixarea D^2), see shell_tile_kernel
}

// Four consecutive nodes ir .. ir + 3 of a halo's blended row: b_k = sum_c w[c] * T[off[c] + ir + k], corners in index order.
// Corners go in groups of four with all their loads in flight together (a plain loop over a run-time corner count waits
// for each corner's two loads before it asks for the next: four dependent L2 round trips for the usual 3-D table).
__device__ __forceinline__ void blend_row4(const DevTable &T, const double *w, const int64_t *off, int ncorner, int ir,
                                           double &b0, double &b1, double &b2, double &b3)
{
    b0 = 0.0; b1 = 0.0; b2 = 0.0; b3 = 0.0;
    int c = 0;
    for (; c + 4 <= ncorner; c += 4) {
        const double *r0 = T.values + off[c] + ir, *r1 = T.values + off[c + 1] + ir;
        const double *r2 = T.values + off[c + 2] + ir, *r3 = T.values + off[c + 3] + ir;
        double v[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[0][k] = r0[k]; v[1][k] = r1[k]; v[2][k] = r2[k]; v[3][k] = r3[k]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // same order of additions as the plain loop
            const double wi = w[c + i];
            b0 = fma(v[i][0], wi, b0); b1 = fma(v[i][1], wi, b1); b2 = fma(v[i][2], wi, b2); b3 = fma(v[i][3], wi, b3);
        }
    }
    for (; c < ncorner; ++c) {
        const double *row = T.values + off[c] + ir;
        const double wi = w[c];
        b0 = fma(row[0], wi, b0); b1 = fma(row[1], wi, b1); b2 = fma(row[2], wi, b2); b3 = fma(row[3], wi, b3);
    }
}

// Same for windows whose length is a multiple of 4: one thread per (halo, 4 consecutive nodes), i.e. 4 x 2^(ndim-1)
// independent table loads in flight per thread (the one-node form is latency-bound).
__global__ __launch_bounds__(256) void halo_row4_kernel(const RowParams P)
{
    constexpr int kHpbMax = 64;
    // corner weights / row offsets of the block's halos: dynamic LDS sized for the table's 2^(ndim-1) corners (a static
    // [64][32] pair of arrays took 33 KB and capped the kernel at 4 workgroups per CU)
    extern __shared__ double smem_row4[];
    const int ncorner_ = 1 << P.tab.nouter;
    double (*s_w) = smem_row4;                                            // [kHpbMax][ncorner]
    int64_t (*s_off) = reinterpret_cast<int64_t *>(smem_row4 + kHpbMax * ncorner_);
    __shared__ int s_winlo[kHpbMax];
    __shared__ double s_add[kHpbMax];
    const DevTable &T = P.tab;
    const int W = P.win_nodes;
    const int tph = W >> 2;                                     // threads per halo
    const int hpb = min(256 / tph, kHpbMax);                    // halos per block
    const int hl = threadIdx.x / tph, e4 = (threadIdx.x - hl * tph) << 2;
    const int64_t j = (int64_t)blockIdx.x * hpb + hl;
    const bool live = (hl < hpb) && (j < P.n_halo);
    const int ncorner = 1 << T.nouter;
    bool skip = true;
    if (live) {
        const int flags = P.ht[j].flags;
        skip = (flags & (HF_SKIP | HF_OOB)) != 0;
        const int q = threadIdx.x - hl * tph;
        if (q == 0) { s_winlo[hl] = P.ht[j].win_lo; s_add[hl] = T.log_values ? P.ht[j].spare[0] : 0.0; }
        for (int c = q; c < ncorner; c += tph) {
            double w = 1.0;
            int64_t off = j * (int64_t)T.hstride;
            for (int k = 0; k < T.nouter; ++k) {
                const int bit = (c >> (T.nouter - 1 - k)) & 1;
                const double y = halo_cell_weight(P.ht, P.cw, P.cap, T.nouter, j, k);
                const int i = halo_cell_index(P.ht, P.cidx, P.cap, T.nouter, j, k);
                w = w * (bit ? y : 1.0 - y);
                off += (int64_t)(i + bit) * T.ostride[k];
            }
            s_w[hl * ncorner_ + c] = w; s_off[hl * ncorner_ + c] = off;
        }
    }
    __syncthreads();
    if (!live || skip) return;
    const int ir = s_winlo[hl] + e4;
    double b0, b1, b2, b3;
    blend_row4(T, s_w + hl * ncorner_, s_off + hl * ncorner_, ncorner, ir, b0, b1, b2, b3);
    const double add = s_add[hl];
    double4 out;
    out.x = b0 + add; out.y = b1 + add; out.z = b2 + add; out.w = b3 + add;
    *reinterpret_cast<double4 *>(P.hwin + j * W + e4) = out;   // 32-byte aligned: W and e4 are multiples of 4
}

// One deferred pixel: blend the halo's corner rows directly from the table (same corner order and arithmetic as
// halo_row_kernel / the in-kernel drain) and add exp(.) to the map.  Several halos can leave the same pixel: an atomic.
template <bool ONLY2 = false>
__device__ inline void deferred_add(const TileParams &P, const DeferredOut &e, const double *__restrict__ exptab)
{
    const DevTable &T = P.tab;
    const int64_t j = e.halo;
    const int i = min(max((int)e.t, 0), T.nr - 2);
    const double f = e.t - (double)i;
    const double lnpf = P.ht[j].spare[0];
    double c0v, c1v;
    halo_row_pair<ONLY2>(T, P.ht, P.cidx, P.cw, P.cap, j, i, c0v, c1v);
    const double L = fma(f, c1v - c0v, c0v) + lnpf;
    if (fabs(L) < 709.0) unsafeAtomicAdd(P.out + e.pix, fast_exp(L, exptab));
}

#ifndef BFG_TILE_THREADS
#define BFG_TILE_THREADS 512
#endif
#ifndef BFG_TILE_WAVES_PER_SIMD
#define BFG_TILE_WAVES_PER_SIMD 4
#endif
#ifndef BFG_STAGE_TIMING
#define BFG_STAGE_TIMING 0
#endif
// shape of the paint tiles: rings per band x pixels per ring row (the product sets the LDS accumulator: 2048 pixels)
// 32 x 64 instead of round 1's 64 x 32: the same number of (halo, tile) pairs, but a disc's ring is cut by fewer sector
// borders, i.e. fewer and longer (pair, ring) segments for stage b: tile kernel -7 % at 1e6 halos (profiles/r02_tile_variants.txt)
#ifndef BFG_PAINT_TR
#define BFG_PAINT_TR 32
#endif
#ifndef BFG_PAINT_TW
#define BFG_PAINT_TW 64
#endif
// offsets tiles (3 accumulators per pixel, 1024 pixels): 16 x 64 instead of 32 x 32, tile kernel -5 %
#ifndef BFG_PAINT_PIXMAX
#define BFG_PAINT_PIXMAX 6912      // pixel -> segment table entries per round of stage c: what the LDS left by the 32-ring rows holds
#endif
#ifndef BFG_BARY_TR
#define BFG_BARY_TR 16
#endif
#ifndef BFG_BARY_TW
#define BFG_BARY_TW 64
#endif
#if BFG_STAGE_TIMING
__device__ unsigned long long g_stage_cycles[16];     // profiling build only: barrier-to-barrier cycles per stage ([8..]: inside stage b)
#endif
#if BFG_STAGE_TIMING && BFG_STAGE_TIMING != 4
#define BFG_TICK(slot) do { if (tid == 64) { const long long now_ = clock64(); st_acc[slot] += now_ - st_t; st_t = now_; } } while (0)
#else
#define BFG_TICK(slot) do { } while (0)
#endif
#if BFG_STAGE_TIMING == 2
#define BFG_SUBTICK(slot) do { if (tid == 64) { const long long now_ = clock64(); st_acc[8 + slot] += now_ - st_sub; st_sub = now_; } } while (0)
#else
#define BFG_SUBTICK(slot) do { } while (0)
#endif
// -DBFG_STAGE_TIMING=3: slots 8 .. 15 split a work item instead (thread BFG_TIMING_TID): top of the item (counter, work
// record), accumulator clear, ring rows, first records, the chunk loop, write-back addresses + deferred pixels, stores,
// end-of-item barrier
#ifndef BFG_TIMING_TID
#define BFG_TIMING_TID 64
#endif
// -DBFG_STAGE_TIMING=4: the timeline of a work item as thread BFG_TIMING_TID sees it, accumulated in LDS (two registers instead
// of the 32 of st_acc, whose spills and their vmcnt(0) waits distort modes 1-3): slots 0 .. 9 = top of the item -> work record
// arrived -> accumulator cleared / rows -> stage a + barrier -> stage b -> its barrier -> pixel loop -> end-of-chunk barrier ->
// write-back issued -> end-of-item barrier
#if BFG_STAGE_TIMING == 4
#define BFG_LTICK(slot) do { if (tid == BFG_TIMING_TID) { const long long now_ = clock64(); \
    atomicAdd(reinterpret_cast<unsigned long long *>(smem_raw + lt_off) + (slot), (unsigned long long)(now_ - lt_last)); lt_last = now_; } } while (0)
#else
#define BFG_LTICK(slot) do { } while (0)
#endif
#if BFG_STAGE_TIMING == 3
#define BFG_ITICK(slot) do { if (tid == BFG_TIMING_TID) { const long long now_ = clock64(); st_acc[slot] += now_ - st_sub; st_sub = now_; } } while (0)
#else
#define BFG_ITICK(slot) do { } while (0)
#endif
constexpr int kTileThreads = BFG_TILE_THREADS;
constexpr int kWinLds = 32;          // row windows up to this many nodes are staged in LDS
constexpr int kPrOff = 64;           // slot offsets of the chunk's pairs (one per lane of a wavefront), padded with INT_MAX
// LDS room for second pieces of ring windows that wrap around inside a sector (only where a sector spans most of a ring,
// next to the poles; a chunk that runs out paints the piece through the direct read-out)
constexpr int kSegExtra = 32;

// per-mode shape of a tile workgroup: threads, rings per tile, accumulators per pixel, LDS capacities of a chunk.
// LIGHT = 1: the instantiation for sparse catalogs (up to ~a dozen pairs per tile).  There a tile is one chunk and the
// workgroup's time is its chain of latencies (pair list -> halo records -> table rows -> write-back), not its arithmetic:
// 256 threads and smaller chunks need 40 KB (paint) / 52 KB (offsets) of LDS, so FOUR / THREE workgroups share a CU instead of
// two and twice / half again as many tiles are in flight.  The host picks it by halos per tile (run_shell).
template <int MODE, int LIGHT = 0> struct TileCfg;
template <> struct TileCfg<MODE_PAINT, 0> {
    // rings per tile, accumulators per pixel, (pair, ring) slots / pairs per chunk, pixel -> segment table entries per round
    static constexpr int NT = kTileThreads, WPS = BFG_TILE_WAVES_PER_SIMD, LDS_MAX = 81920;
    static constexpr int TR = BFG_PAINT_TR, TW = BFG_PAINT_TW, NACC = 1, SLOTMAX = 512, PAIRMAX = 64, PIXMAX = BFG_PAINT_PIXMAX, QCAP = 40;
    static constexpr int SEGMAX = SLOTMAX + kSegExtra;
    using Pair = PairInfo;
};
template <> struct TileCfg<MODE_BARYONIFY, 0> {
    static constexpr int NT = kTileThreads, WPS = BFG_TILE_WAVES_PER_SIMD, LDS_MAX = 81920;
    static constexpr int TR = BFG_BARY_TR, TW = BFG_BARY_TW, NACC = 3, SLOTMAX = 448, PAIRMAX = 48, PIXMAX = 4096, QCAP = 0;
    static constexpr int SEGMAX = SLOTMAX + kSegExtra;
    using Pair = PairInfoDisp;
};
template <> struct TileCfg<MODE_PAINT, 1> {
    // FOUR workgroups per CU (40 KB of LDS each): the same 16 wavefronts per CU as two 512-thread workgroups, twice as many items
    // in flight.  Measured (profiles/r03_light_ab.txt): 1e4 halos 0.072 -> 0.057 ms (three per CU with the larger chunks: 0.062),
    // 3e4 halos 0.085 -> 0.079, 1e5 halos 0.140 -> 0.160 (three chunks of 160 slots per item instead of one of 512).
    static constexpr int NT = 256, WPS = 4, LDS_MAX = 40960;
    static constexpr int TR = BFG_PAINT_TR, TW = BFG_PAINT_TW, NACC = 1, SLOTMAX = 160, PAIRMAX = 20, PIXMAX = 2048, QCAP = 8;
    static constexpr int SEGMAX = SLOTMAX + kSegExtra / 2;
    using Pair = PairInfo;
};
template <> struct TileCfg<MODE_BARYONIFY, 1> {
    static constexpr int NT = 256, WPS = 3, LDS_MAX = 54608;
    static constexpr int TR = BFG_BARY_TR, TW = BFG_BARY_TW, NACC = 3, SLOTMAX = 192, PAIRMAX = 24, PIXMAX = 2048, QCAP = 0;
    static constexpr int SEGMAX = SLOTMAX + kSegExtra / 2;
    using Pair = PairInfoDisp;
};

template <int MODE, int LIGHT = 0>
__host__ __device__ constexpr size_t tile_lds_bytes()
{
    using Cfg = TileCfg<MODE, LIGHT>;
    return (size_t)Cfg::TR * Cfg::TW * Cfg::NACC * sizeof(double) + kLogTab * sizeof(double2) +
           (kExpTab + kAtanTab) * sizeof(double) + Cfg::TR * sizeof(RingRow) + Cfg::SEGMAX * sizeof(Seg) +
           Cfg::PAIRMAX * sizeof(typename Cfg::Pair) + (size_t)Cfg::PAIRMAX * kWinLds * sizeof(double) +
           Cfg::PIXMAX * sizeof(uint16_t) + kPrOff * sizeof(int32_t) + Cfg::SEGMAX * sizeof(uint8_t) +
           Cfg::QCAP * sizeof(DeferredPixel) + 24 * sizeof(int32_t);
}

// two (LIGHT: three) tile workgroups share a CU's 160 KB of LDS
static_assert(tile_lds_bytes<MODE_PAINT>() <= 81920 && tile_lds_bytes<MODE_BARYONIFY>() <= 81920, "two workgroups per CU");
static_assert(tile_lds_bytes<MODE_PAINT, 1>() <= 40960 && tile_lds_bytes<MODE_BARYONIFY, 1>() <= 54608, "four / three workgroups per CU");

// sin(h) for h^2 <= kSinSmall: odd series to h^7 (rel err < 3e-12)
__device__ inline double sin_small(double h, double h2)
{
    double p = fma(h2, -1.0 / 5040.0, 1.0 / 120.0);
    p = fma(h2, p, -1.0 / 6.0);
    p = fma(h2, p, 1.0);
    return h * p;
}

// cos(h) for h^2 <= kSinSmall: even series to h^10 (next term h^12 / 12! < 1e-25; rel err < 1e-16 of a value >= 0.98)
__device__ inline double cos_small(double h2)
{
    double p = fma(h2, -1.0 / 3628800.0, 1.0 / 40320.0);
    p = fma(h2, p, -1.0 / 720.0);
    p = fma(h2, p, 1.0 / 24.0);
    p = fma(h2, p, -0.5);
    return fma(h2, p, 1.0);
}

// 1 / x for a normal x away from the ends of the exponent range: v_rcp_f64 + two Newton steps (<= 1 ulp); no scaling, no
// special cases, unlike the IEEE division sequence (div_scale x 2, div_fmas, div_fixup: twice the instructions)
__device__ inline double rcp_newton(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// (sin h, cos h) for |h| <= 3.2 without libm: series on h/16, then four angle doublings
__device__ inline void sincos_wide(double h, double &sh, double &ch)
{
    const double q = 0.0625 * h;
    double s = sin_small(q, q * q);
    double c = sqrt(1.0 - s * s);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double s2 = 2.0 * s * c; c = 1.0 - 2.0 * s * s; s = s2; }
    sh = s; ch = c;
}

// sqrt(x) for x in (1e-290, 1e290) (here: 1 - z^2 - x^2 <= 1): v_rsq_f64 + Goldschmidt / Newton steps, <= 1 ulp; no
// range scaling and no special cases, unlike libm's
__device__ inline double sqrt_unit(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return fma(fma(-g, g, x), h, g);
}

// atan2(y, x) for y >= 0 (result in [0, pi]), <= 2 ulp, without libm: the smaller of (|x|, y) over the larger is
// brought to |u| <= 1/8 with atan(t) = atan(c) + atan((t - c) / (1 + c t)), c in {0, 1/4, 1/2, 3/4, 1}; one division
// (v_rcp_f64 + two Newton steps + one correction of the quotient), odd series to u^17.  ocml's atan2 is ~3x longer.
__device__ inline double atan2_upper(double y, double x)
{
    const double ax = fabs(x);
    const double mx = fmax(ax, y), mn = fmin(ax, y);
    double c = 0.0, ac = 0.0;
    if (mn > 0.125 * mx) { c = 0.25; ac = 0.24497866312686415417; }
    if (mn > 0.375 * mx) { c = 0.5; ac = 0.46364760900080611621; }
    if (mn > 0.625 * mx) { c = 0.75; ac = 0.64350110879328438680; }
    if (mn > 0.875 * mx) { c = 1.0; ac = 0.78539816339744830962; }
    const double num = fma(-c, mx, mn), den = fma(c, mn, mx);
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double u = num * r;
    u = fma(fma(-den, u, num), r, u);
    const double u2 = u * u;
    double p = fma(u2, 1.0 / 17.0, -1.0 / 15.0);
    p = fma(u2, p, 1.0 / 13.0);
    p = fma(u2, p, -1.0 / 11.0);
    p = fma(u2, p, 1.0 / 9.0);
    p = fma(u2, p, -1.0 / 7.0);
    p = fma(u2, p, 1.0 / 5.0);
    p = fma(u2, p, -1.0 / 3.0);
    double a = ac + fma(u * u2, p, u);
    a = (ax < y) ? 1.57079632679489661923 - a : a;
    return (x < 0.0) ? 3.14159265358979323846 - a : a;
}

// The same with the reduction constant from a table: c = k / 64 nearest to min / max (picked through an approximate
// quotient: any neighbour of the nearest will do), atan(c) from `tab` (LDS), |u| <= 1/128 + 2^-20 so the odd series stops at
// u^7 (next term < 2e-18 relative).  One division as before; ~25 instructions fewer than the compare-and-select ladder.
__device__ inline double atan2_upper_tab(double y, double x, const double *tab)
{
    const double ax = fabs(x);
    const double mx = fmax(ax, y), mn = fmin(ax, y);
    int k = (int)fma(mn * __builtin_amdgcn_rcp(mx), 64.0, 0.5);        // v_rcp_f64: ~1e-8 relative, good for any normal mx
    k = min(max(k, 0), 64);
    const double c = (double)k * 0.015625;
    const double num = fma(-c, mx, mn), den = fma(c, mn, mx);
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double u = num * r;
    u = fma(fma(-den, u, num), r, u);
    const double u2 = u * u;
    double p = fma(u2, -1.0 / 7.0, 1.0 / 5.0);
    p = fma(u2, p, -1.0 / 3.0);
    double a = tab[k] + fma(u * u2, p, u);
    a = (ax < y) ? 1.57079632679489661923 - a : a;
    return (x < 0.0) ? 3.14159265358979323846 - a : a;
}

__device__ inline int med3_i32(int x, int lo, int hi)      // clamp(x, lo, hi) in one instruction (lo <= hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}

using lds_double = __attribute__((address_space(3))) double;
template <typename T>
__device__ inline __attribute__((address_space(3))) T *lds_ptr(unsigned byte_address)
{
    return (__attribute__((address_space(3))) T *)(uintptr_t)byte_address;
}

// inclusive prefix sum over the 64 lanes of a wavefront with DPP row shifts / row broadcasts (no LDS round trips;
// __shfl_up compiles to ds_bpermute, ~100 cycles per step)
__device__ inline int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also waits for
// vmcnt(0), i.e. for every global load, store and LDS-DMA the wavefront has in flight: prefetches meant to stay in flight
// across the barrier, and -- in a persistent workgroup -- the previous tile's write-back stores (~5 us per tile).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// BLEND: the instantiation whose stage b blends the pairs' row windows from the table itself (TileParams::blend; paint, 3-D
// tables, dense catalogs).  A template parameter, not a run-time branch: the blend code costs the other path registers (20 -> 36 B
// of scratch and + 4 % on the 1e5-halo run when it was a branch).
// Hand-over of the persistent loop's look-ahead through LDS.  On gfx9 vector loads and stores share the one in-order vmcnt
// counter: a wavefront that waits for a vector load (or a returning atomic) after it has issued its write-back stores waits for
// those stores to be acknowledged first -- ~1-2 us per work item, 17-27 % of a sparse catalog's item (profiles/r03_item_timeline.txt).
// So the look-ahead never passes through a wavefront's registers after its stores: at the top of item k thread 0 requests the
// index of item k + 3 (a returning atomic; parked in LDS right BEFORE the write-back, when it has long arrived and no store is in
// flight) and wave 0 sends the record of item k + 2 straight into LDS by LDS-DMA (two lanes x 16 B; landed by the time the
// barrier that ends stage b has drained wave 0's vmcnt; two alternating slots, so the next item's DMA cannot overtake a slow
// reader); every wavefront picks both up from LDS after the end-of-item barrier.
// LIGHT & 2 (with BLEND): the "table" is the halos' own radial rows (run_shell_nd: values + j * hstride, no outer axes) -- stage b copies
// each pair's window straight out of its halo's row (+ ln(pixarea D^2)) instead of blending four corner rows: no row windows in HBM
// and no row phase in the prep kernel for the N-dimensional tables either.
template <int MODE, bool WIN_LDS, int LIGHT = 0, bool BLEND = false>
__global__ __launch_bounds__((TileCfg<MODE, (LIGHT & 1)>::NT), (TileCfg<MODE, (LIGHT & 1)>::WPS)) void shell_tile_kernel(const TileParams P)
{
    static_assert(!BLEND || (MODE == MODE_PAINT && WIN_LDS), "the blend instantiation is paint with LDS windows");
    constexpr bool ROWS = (LIGHT & 2) != 0;
    static_assert(!ROWS || BLEND, "rows are a variant of the blend instantiation");
    using Cfg = TileCfg<MODE, (LIGHT & 1)>;
    using Pair = typename Cfg::Pair;
    constexpr int TR = Cfg::TR, TW = Cfg::TW, NT = Cfg::NT, NACC = Cfg::NACC;
    static_assert(TR <= 64 && TW <= 128, "segment records pack the ring row in 6 bits; segment lengths in 8");
    constexpr int kTileWaves = NT / 64;
    constexpr int kSegMax = Cfg::SEGMAX, kPairMax = Cfg::PAIRMAX, kSlotMax = Cfg::SLOTMAX, kPixMax = Cfg::PIXMAX;
    static_assert(kSegMax * sizeof(Seg) <= 65536, "ptab holds 16-bit byte offsets of segment records");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS layout (byte offsets are compile-time constants: segment records carry LDS byte addresses)
    constexpr int acc_off = 0;                                                // double [TR*TW*NACC]
    constexpr int logtab_off = acc_off + TR * TW * NACC * (int)sizeof(double);    // double2 [128]
    constexpr int exptab_off = logtab_off + kLogTab * (int)sizeof(double2);   // double [64]
    constexpr int atantab_off = exptab_off + kExpTab * (int)sizeof(double);   // double [72]
    constexpr int rows_off = atantab_off + kAtanTab * (int)sizeof(double);    // RingRow [TR]
    constexpr int segs_off = rows_off + TR * (int)sizeof(RingRow);            // Seg [kSegMax]
    constexpr int pinfo_off = segs_off + kSegMax * (int)sizeof(Seg);          // Pair [kPairMax]
    constexpr int pwin_off = pinfo_off + kPairMax * (int)sizeof(Pair);        // double [kPairMax][kWinLds] row values B_i
    constexpr int ptab_off = pwin_off + kPairMax * kWinLds * (int)sizeof(double);   // uint16 [kPixMax] pixel -> segment
    constexpr int proff_off = ptab_off + kPixMax * (int)sizeof(uint16_t);     // int32 [kPrOff] exclusive slot offsets of the pairs
    constexpr int scnt_off = proff_off + kPrOff * (int)sizeof(int32_t);       // uint8 [kSegMax] pixel count of every segment
    constexpr int kQCap = Cfg::QCAP;
    constexpr int rq_off = scnt_off + kSegMax * (int)sizeof(uint8_t);         // DeferredPixel [kQCap]
    constexpr int ctl_off = rq_off + kQCap * (int)sizeof(DeferredPixel);      // n_take, nslots, extra segments, pixel total, -, queue fill
    static_assert(kSegMax % 16 == 0 && kPairMax <= kPrOff && kSlotMax <= NT && kPairMax <= 64, "chunk shape");
    double *acc = reinterpret_cast<double *>(smem_raw + acc_off);
    double2 *logtab = reinterpret_cast<double2 *>(smem_raw + logtab_off);
    double *exptab = reinterpret_cast<double *>(smem_raw + exptab_off);
    double *atantab = reinterpret_cast<double *>(smem_raw + atantab_off);
    RingRow *rows = reinterpret_cast<RingRow *>(smem_raw + rows_off);
    Seg *segs = reinterpret_cast<Seg *>(smem_raw + segs_off);
    Pair *pinfo = reinterpret_cast<Pair *>(smem_raw + pinfo_off);
    double *pwin = reinterpret_cast<double *>(smem_raw + pwin_off);
    uint16_t *ptab = reinterpret_cast<uint16_t *>(smem_raw + ptab_off);
    int32_t *pr_off = reinterpret_cast<int32_t *>(smem_raw + proff_off);
    uint8_t *scnt = reinterpret_cast<uint8_t *>(smem_raw + scnt_off);
    [[maybe_unused]] DeferredPixel *rq = reinterpret_cast<DeferredPixel *>(smem_raw + rq_off);
    int32_t *ctl = reinterpret_cast<int32_t *>(smem_raw + ctl_off);
    static_assert(sizeof(RingRow) % 16 == 0 && sizeof(Pair) % 16 == 0, "16-byte aligned LDS records");
    static_assert(ctl_off + 24 * sizeof(int32_t) == tile_lds_bytes<MODE, (LIGHT & 1)>(), "layout and tile_lds_bytes() must agree");
    static_assert(ctl_off % 16 == 0, "the work records parked in ctl[8 .. 23] are 16-byte accesses");

    // Persistent workgroups: the grid is a few workgroups per CU and every workgroup takes work items from a counter until
    // the list is empty.  (One workgroup per item -- 16 640 launches at NSIDE 1024, 10 368 of them with nothing to do --
    // cost more in dispatch than the items of a sparse catalog cost to process: at 1e4 halos the kernel took 0.10 ms
    // whether two or three workgroups shared a CU.)  The ln / exp tables are loaded once per workgroup.
    // the work-list state is the same in every lane: readfirstlane tells the compiler, which then keeps it (and what is derived
    // from it: pair counts, ring range, sector) in scalar registers instead of ~30 vector registers: -4 % at 1e5 halos
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni4 = [&](int4 v) { return make_int4(uni(v.x), uni(v.y), uni(v.z), uni(v.w)); };
    const int item_first = P.slice ? uni(P.slice[0]) : 0;
    const int n_work_total = P.slice ? uni(P.slice[1]) : uni(*P.n_work);
    // the binning gave up (pair buffer too small): every halo goes to the scatter kernel; an uninitialised map still has to be cleared
    const bool degraded = (long long)P.tile_start[P.geo.ntiles] > P.pair_cap;
    // sliced call with left-over halos: they are in the (cleared) output already
    const bool accum = P.accum_left && (uni(*P.accum_left) > 0 || degraded);
    const int p_overwrite = accum ? 0 : P.overwrite, p_out_zero = accum ? 0 : P.out_zero;
    if constexpr (Cfg::QCAP > 0) { if (P.defer && threadIdx.x == 0) P.defer_count[blockIdx.x] = 0; }   // (a workgroup that returns early left nothing)
    if (degraded && !p_overwrite) return;
    const Hpx &hp = P.hpx;
    const DevTable &T = P.tab;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if BFG_STAGE_TIMING && BFG_STAGE_TIMING != 4
    long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = clock64();
    [[maybe_unused]] long long st_sub = st_t;
#endif
#if BFG_STAGE_TIMING == 4
    constexpr int lt_off = ctl_off + 24 * (int)sizeof(int32_t);      // 16 x 8 B behind the layout (the host asks for 128 B more)
    long long lt_last = 0;
    if (tid < 16) reinterpret_cast<unsigned long long *>(smem_raw + lt_off)[tid] = 0ull;
#endif
    if (tid < kLogTab) logtab[tid] = P.logtab[tid];
    if (tid < kExpTab) exptab[tid] = P.exptab[tid];
    if (tid >= 128 && tid < 128 + kAtanTab) atantab[tid - 128] = P.atantab[tid - 128];
    // segment records carry absolute LDS byte addresses: the dynamic LDS block of this kernel (it has no static
    // __shared__) starts at address 0; refuse to run otherwise
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem_raw != 0u) {
        if (threadIdx.x == 0) atomicOr(&P.stats->warn_mask, 0x80000000u);
        return;
    }
    constexpr unsigned lds_base = 0u;
    const double inv_dr = T.inv_dr;
    const double t_c = (-T.r0) * inv_dr, t_m = 0.5 * inv_dr;     // cell coordinate t = ln(x) * t_m + t_c
    // the pixel loop works with t1 = t + 1 from the exponent-biased logarithm: t1 = fast_log_biased(x) * t_m + t_c1
    const double t_c1 = t_c + 1.0 - kLogBias * t_m;
    const int NRm1 = T.nr - 1;
    const int W = P.win_nodes;
    constexpr bool win_in_lds = WIN_LDS;     // row windows of <= kWinLds nodes live in LDS, longer ones stay in HBM/L2
    unsigned long long px_total = 0;
    unsigned int oob_total = 0;
    int dfill = 0;                                    // entries in this workgroup's slice of the deferred-pixel list (uniform)

    // Work items are handed out by a counter (they differ in size: a static round-robin was 5 % slower) and pipelined three
    // deep, so that no item starts with the chain of dependent global loads counter -> work record -> pair list -> halo
    // records: while item k is processed, the index of item k + 3 is requested, the work record of item k + 2 and the first
    // pair-list windows of item k + 1 are loaded, and -- after the pixel stage of item k's last chunk -- wave 0 requests the
    // halo records of item k + 1's first candidates.  An item then starts with everything but its row windows at hand.
    // (Stripped of all its work the kernel took 0.083 ms for the 6 272 items of an NSIDE-1024 map.)  The first three items of
    // a workgroup are blockIdx, blockIdx + gridDim, blockIdx + 2 gridDim; the counter starts at 3 gridDim.
    int pjA = -1, pjB = -1, pjA1 = -1, pjB1 = -1;
    int nx_j = -1, nx_first = 0, nx_last = -1, nx_wl = 0;
    [[maybe_unused]] double nx_lnpf = 0.0;
    bool primed = false;                              // wave 0 holds the first chunk's candidates of the item about to start
    constexpr int kNoItem = 0x7fffffff;
    // The items after a workgroup's first three are handed out by counters -- several of them: every returning atomic on ONE
    // address costs the memory side ~12.6 ns, i.e. 0.079 ms for the 6 272 items of an NSIDE-1024 map whatever the items hold (the
    // whole tile kernel takes 0.15 ms at 1e5 halos), and under that load its latency sits in front of every later load of wave 0
    // (one in-order vmcnt).  Counter c, 1 KB from the next, hands out the items c, c + K, c + 2 K, ... (interleaved: neighbouring
    // tiles cost about the same) to the workgroups with blockIdx = c mod K.
    const int n_counters = uni(P.n_counters), my_c = (int)blockIdx.x % n_counters;
    int32_t *my_counter = P.work_counter + my_c * kCounterStride;
    int item = item_first + (int)blockIdx.x;
    int item1 = P.work_counter ? item_first + (int)(blockIdx.x + gridDim.x) : kNoItem;
    int item2 = P.work_counter ? item_first + (int)(blockIdx.x + 2 * gridDim.x) : kNoItem;
    const int4 wzero = make_int4(0, 0, 0, 0);
    int4 wk = wzero, wg = wzero, wk1 = wzero, wg1 = wzero;
    if (item < n_work_total) { wk = uni4(P.work[2 * item]); wg = uni4(P.work[2 * item + 1]); }
    if (item1 < n_work_total) { wk1 = uni4(P.work[2 * item1]); wg1 = uni4(P.work[2 * item1 + 1]); }
    int par = 0;                                      // which of the two LDS slots this iteration's record DMA uses
    while (item < n_work_total) {
#if BFG_STAGE_TIMING == 3
    if (tid == BFG_TIMING_TID) st_sub = clock64();
#endif
#if BFG_STAGE_TIMING == 4
    if (tid == BFG_TIMING_TID) lt_last = clock64();
#endif
    int item3 = kNoItem;
    if (tid == 0 && P.work_counter) item3 = atomicAdd(my_counter, 1);      // (the counter's value; made an item index where it is parked)
    const bool have_next = item1 < n_work_total;
    if (wave == 0 && lane < 2 && item2 < n_work_total)    // the record of item k + 2 -> ctl[8 + 8 par ..] (see the hand-over above)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P.work + 2 * item2 + lane),
                                         (__attribute__((address_space(3))) void *)(ctl + 8 + 8 * par), 16, 0, 0);
    if (wave == 0 && have_next) {                     // first pair-list windows of the next item
        const int32_t *plist1 = P.pairs + wk1.y;
        // (the binning gave up: the work records point past the end of the pair buffer -- found by the soak as a memory fault
        // once the buffer happened to be the last allocation of its region)
        const int n1 = degraded ? 0 : wk1.z - wk1.y;
        pjA1 = (lane < n1) ? plist1[lane] : -1;
        pjB1 = (64 + lane < n1) ? plist1[64 + lane] : -1;
    }
    const int n_pairs = degraded ? 0 : wk.z - wk.y;
    bool handed = false;
    const int band = wg.x, sector = wg.y, NS = wg.z;
    const int ring_lo = 1 + band * TR;
    const int ring_hi = min((int)(4 * hp.nside - 1), ring_lo + TR - 1);

    BFG_ITICK(8);
    BFG_LTICK(0);
    if (n_pairs != 0) {
    for (int i = tid; i < TR * TW * NACC; i += NT) acc[i] = 0.0;
    if (tid == 0) ctl[5] = 0;
    BFG_ITICK(9);
    // the ring rows are wave 1's (stage b is their first reader): wave 0 goes straight to stage a of the first chunk, and the one
    // barrier after stage a covers the cleared accumulator, the rows and the pair records alike
    constexpr int kRowWave = (NT >= 128) ? 64 : 0;
    // ... and the sector starts of the rows are wave 2's: two f64 divisions and four 64-bit products per ring, as long again as the
    // ring's geometry (a square root and a division), and every wavefront waits for the rows at the barrier after stage a
    // (paint: - 2 % at 1e6 and 1e5 halos; the offsets kernel, 16 rings per tile, lost 0.5 % and keeps both on wave 1: profiles/r04_prep_ab.txt)
    constexpr int kSecWave = (MODE == MODE_PAINT && NT >= 192) ? 128 : kRowWave;
    if (tid >= kRowWave && tid < kRowWave + TR) {
        const int rtid = tid - kRowWave;
        const int ring = ring_lo + rtid;
        double rz = 0, rsth = 0, rstep = 0, roff = 0;
        int64_t rstart = 0;
        if (ring <= ring_hi) {
            const RingGeom g = ring_geom(hp, ring);
            rz = g.z;                  // identical formula to ring2z, which query_disc uses
            rsth = g.sth; rstep = g.phistep; roff = g.phioff; rstart = g.start;
        }
        RingRow &rr = rows[rtid];
        rr.z = rz; rr.sth = rsth; rr.phistep = rstep; rr.phioff = roff; rr.start = rstart; rr.pad = 0;
    }
    if (tid >= kSecWave && tid < kSecWave + TR) {
        const int rtid = tid - kSecWave;
        const int ring = ring_lo + rtid;
        int rnr = 1, rk0 = 0, rk1 = 0, rrowoff = 0;
        if (ring <= ring_hi) {
            const int nside = (int)hp.nside;
            const int gnr = 4 * min(min(ring, nside), 4 * nside - ring);       // ring_geom's nr
            // floor(sector nr / NS) without a 64-bit integer division (~80 instructions each, on a wavefront every item
            // waits for): the double quotient of two integers below 2^53 lies at least 1 / NS from the next integer, far more
            // than its rounding error, and one multiply-compare makes it exact regardless
            // (and no f64 division either: the reciprocal of NS by Newton, once; the quotient only has to land within 1)
            const double inv_ns = rcp_newton((double)NS);
            auto sector_start = [&](int sct) -> int {
                const uint64_t prod = (uint64_t)sct * (uint64_t)gnr;
                uint64_t q = (uint64_t)((double)prod * inv_ns);
                if ((q + 1) * (uint64_t)NS <= prod) ++q;
                if (q * (uint64_t)NS > prod) --q;
                return (int)q;
            };
            rnr = gnr;
            rk0 = sector_start(sector);
            rk1 = sector_start(sector + 1);
            rrowoff = rtid * TW - rk0;
        }
        RingRow &rr = rows[rtid];
        rr.nr = rnr; rr.k0 = rk0; rr.k1 = rk1; rr.rowoff = rrowoff;
    }
    BFG_ITICK(10);

    unsigned long long my_pixels = 0;
    const int32_t *plist = P.pairs + wk.y;
    const int tile = wk.x;
    (void)tile;

    // Wave 0 runs the pair pipeline two chunks deep, so that no stage waits for a dependent pair -> halo record load:
    //   pjA / pjB  the pair list entries [base, base + 128) of the current chunk's first pair (issued one chunk ago);
    //   nx_*       halo-record fields of the NEXT chunk's candidate pairs.
    // Both sets of loads are issued at the start of stage c (not before the stage-b barrier, whose vmcnt(0) -- needed
    // for the LDS-DMA -- would expose their latency) and are consumed in the next chunk's stage a.
    auto load_list_windows = [&](int wb) {
        pjA = (wb + lane < n_pairs) ? plist[wb + lane] : -1;
        pjB = (wb + 64 + lane < n_pairs) ? plist[wb + 64 + lane] : -1;
    };
    auto load_records = [&]() {
        nx_first = 0; nx_last = -1;
        if (nx_j >= 0) {
            const HaloTile &h = P.ht[nx_j];
            nx_first = h.rfirst; nx_last = h.rlast; nx_wl = h.win_lo;
            if constexpr (MODE == MODE_PAINT) nx_lnpf = h.spare[0];
        }
    };
    if (wave == 0 && !primed) {
        load_list_windows(0);
        nx_j = (lane < kPairMax) ? pjA : -1;
        load_records();
    }
    // (no barrier here: stage a of the first chunk reads nothing the prologue wrote; the barrier after it orders the rest)

    unsigned int n_oob32 = 0;

    // rare: a pixel whose table cell lies outside the pair's staged row window -> blend the corners directly
    auto direct_row_halo = [&](int64_t j, double t) -> double {
        const int i = min(max((int)t, 0), NRm1 - 1);
        const double f = t - (double)i;
        double c0v, c1v;
        halo_row_pair<(BLEND && !ROWS)>(T, P.ht, P.cidx, P.cw, P.cap, j, i, c0v, c1v);
        return fma(f, c1v - c0v, c0v);
    };
    auto direct_row = [&](int pidx, double t) -> double {
        double L = direct_row_halo(pinfo[pidx].halo, t);
        if constexpr (MODE == MODE_PAINT) L += pinfo[pidx].lnpf;
        return L;
    };
    // paint: those pixels (the innermost few of large discs, ~0.2 %) cost two dependent rounds of global loads.
    // Done inline they stall their wavefront and, through the end-of-chunk barrier, the whole workgroup; so they
    // are queued in LDS and drained by all threads at once when the queue is half full and at the end of the tile.
    const int qcap = (P.debug & 32) ? min(4, kQCap) : kQCap;      // debug bit 32: tiny queue (tests the inline fallback)
    [[maybe_unused]] auto drain = [&]() {
        if constexpr (kQCap > 0) {
            const int n = min(ctl[5], qcap);
            for (int i = tid; i < n; i += NT) {
                const DeferredPixel e = rq[i];
                const double L = direct_row_halo(e.halo, e.t) + P.ht[e.halo].spare[0];
                if (fabs(L) < 709.0)
                    __hip_atomic_fetch_add(lds_ptr<double>(lds_base + e.abyte), fast_exp(L, exptab), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            lds_barrier();
            if (tid == 0) ctl[5] = 0;
            lds_barrier();
        }
    };

    // interpolant of the pair's blended row.  t1 = cell coordinate + 1 (so that truncation is a floor for everything
    // that can be in range: t1 in (0, 1) -- below the table -- truncates to 0, never a window cell); wl1 = first cell
    // of the staged window + 1; the segment's wbyte is biased by -8 to match.  `in` = the cell lies in the window.
    auto window_row = [&](const Seg &sg, int wl1, int pidx, double t1, bool &in) -> double {
        const int i1 = (int)t1;                                           // saturating conversion
        const int ic = med3_i32(i1, wl1, wl1 + W - 2);
        in = (ic == i1);
        const double f = t1 - (double)ic;
        double B0, B1;
        if constexpr (win_in_lds) {
            const lds_double *wp = lds_ptr<double>(lds_base + sg.wbyte + 8 * ic);
            B0 = wp[0]; B1 = wp[1];
        } else if (P.win_table) {
            // Finely sampled radial axes (the reference's examples use 2000 nodes: 174 per e-fold of radius) need ~600 nodes
            // per halo, more than the halo has pixels, so pre-blended windows cost more than they save (4.8 KB per halo
            // written and read back through HBM).  Blend the corner rows of the halo's (z, M[, params]) cell per pixel instead:
            // the table stays in L2.  Same corner order and the same + ln(pixarea D^2) as halo_row_kernel: same bits.
            const int ncorner = 1 << T.nouter;                             // <= kWinLds / 2 (host-checked)
            const double *cwn = pwin + pidx * kWinLds;                     // [0 .. nc) weights, [nc .. 2 nc) row offsets (int64)
            const int64_t *con = reinterpret_cast<const int64_t *>(cwn + ncorner);
            B0 = 0.0; B1 = 0.0;
            if (in) {                                  // a cell outside the axis is the caller's business; never read there
                // corners four at a time with all their loads in flight together (the plain loop waited for two corners' loads
                // before it asked for the next two: two dependent L2 round trips per pixel, and this read-out is latency-bound);
                // same order of additions
                int c = 0;
                for (; c + 4 <= ncorner; c += 4) {
                    const double *r0 = T.values + con[c] + (ic - 1), *r1 = T.values + con[c + 1] + (ic - 1);
                    const double *r2 = T.values + con[c + 2] + (ic - 1), *r3 = T.values + con[c + 3] + (ic - 1);
                    const double a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1], a2 = r2[0], b2 = r2[1], a3 = r3[0], b3 = r3[1];
                    B0 = fma(a0, cwn[c], B0); B1 = fma(b0, cwn[c], B1);
                    B0 = fma(a1, cwn[c + 1], B0); B1 = fma(b1, cwn[c + 1], B1);
                    B0 = fma(a2, cwn[c + 2], B0); B1 = fma(b2, cwn[c + 2], B1);
                    B0 = fma(a3, cwn[c + 3], B0); B1 = fma(b3, cwn[c + 3], B1);
                }
                for (; c < ncorner; ++c) {
                    const double *row = T.values + con[c] + (ic - 1);
                    const double w = cwn[c];
                    B0 = fma(row[0], w, B0); B1 = fma(row[1], w, B1);
                }
                if constexpr (MODE == MODE_PAINT) { const double add = pinfo[pidx].lnpf; B0 += add; B1 += add; }
            }
        } else {
            const double *wp = P.hwin + pinfo[pidx].hoff + (ic - wl1);
            B0 = wp[0]; B1 = wp[1];
        }
        return fma(f, B1 - B0, B0);
    };

    // ptab[q0 .. q1) = off, two entries per store where aligned
    auto fill_ptab = [&](int q0, int q1, uint32_t off) {
        int q = q0;
        if ((q & 1) && q < q1) { ptab[q] = (uint16_t)off; ++q; }
        const uint32_t two = off | (off << 16);
#pragma unroll 1
        for (; q + 1 < q1; q += 2) *reinterpret_cast<uint32_t *>(ptab + q) = two;
        if (q < q1) ptab[q] = (uint16_t)off;
    };

    // one pixel of the flattened chunk: segment record -> chord^2 -> ln -> row read-out -> accumulate in LDS
    auto do_pixel = [&](int q, const Seg &sg) {
        const int k = q - sg.excl;                                         // pixel index inside the segment
        const double h = fma((double)k, sg.hstep, sg.c0);
        const double h2 = h * h;
        if constexpr (MODE == MODE_PAINT) {
            if constexpr (win_in_lds) {
                // the common case behind ONE branch (small angle, cell inside the staged window, value inside exp's range); a
                // pixel that fails any of the three falls through to the general code below, which starts over
                const double xf = fma(sg.Bq, sin_squared_small(h2), sg.Aq);
                const double t1f = fma(fast_log_biased(xf, logtab), t_m, t_c1);
                const int i1f = (int)t1f;
                const int icf = med3_i32(i1f, sg.pk, sg.pk + W - 2);
                const lds_double *wpf = lds_ptr<double>(lds_base + sg.wbyte + 8 * icf);
                const double B0f = wpf[0], B1f = wpf[1];
                const double Lf = fma(t1f - (double)icf, B1f - B0f, B0f);
                if ((h2 <= kSinSmall) && (icf == i1f) && (fabs(Lf) < 709.0)) {
                    __hip_atomic_fetch_add(lds_ptr<double>(lds_base + sg.abyte + 8 * k), fast_exp(Lf, exptab), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    return;
                }
            }
            if constexpr (!win_in_lds) {
                // the same for the table-direct read-out of a 3-D table (finely sampled radial axes): the pair's four corner weights,
                // the offset of its first corner row and ln(pixarea D^2) come from ONE 72-byte LDS record (stage b), the other three
                // rows are strides away; the four 16-byte loads go out together and the cell is clamped, so they need no guard
                if (P.win_table && T.nouter == 2) {
                    const double xf = fma(sg.Bq, sin_squared_small(h2), sg.Aq);
                    const double t1f = fma(fast_log_biased(xf, logtab), t_m, t_c1);
                    const int i1f = (int)t1f;
                    const int icf = med3_i32(i1f, sg.pk, sg.pk + W - 2);
                    const double *rec = pwin + sg.wbyte * kWinLds;
                    const double w0 = rec[0], w1 = rec[1], w2 = rec[2], w3 = rec[3], addf = rec[8];
                    const double *r0 = T.values + reinterpret_cast<const int64_t *>(rec)[4] + (icf - 1);
                    const double *r1 = r0 + T.ostride[1], *r2 = r0 + T.ostride[0], *r3 = r2 + T.ostride[1];
                    const double a0 = r0[0], b0 = r0[1], a1 = r1[0], b1 = r1[1], a2 = r2[0], b2 = r2[1], a3 = r3[0], b3 = r3[1];
                    const double B0f = fma(a3, w3, fma(a2, w2, fma(a1, w1, fma(a0, w0, 0.0)))) + addf;
                    const double B1f = fma(b3, w3, fma(b2, w2, fma(b1, w1, fma(b0, w0, 0.0)))) + addf;
                    const double Lf = fma(t1f - (double)icf, B1f - B0f, B0f);
                    if ((h2 <= kSinSmall) && (icf == i1f) && (fabs(Lf) < 709.0)) {
                        __hip_atomic_fetch_add(lds_ptr<double>(lds_base + sg.abyte + 8 * k), fast_exp(Lf, exptab), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                        return;
                    }
                }
            }
            double s2 = sin_squared_small(h2);
            if (__any(h2 > kSinSmall)) {                                   // wave-uniform branch: only near the poles
                if (h2 > kSinSmall) s2 = sin_squared_wide(h);
            }
            const double x = fma(sg.Bq, s2, sg.Aq);                        // r_com^2
            // x = 0 or NaN never lands in the window (ln of the bit pattern is hugely negative / NaN)
            const double t1 = fma(fast_log_biased(x, logtab), t_m, t_c1);
            const int wl1 = sg.pk;
            bool in;
            double L = window_row(sg, wl1, win_in_lds ? 0 : sg.wbyte, t1, in);
            if (!in) {                                                     // divergent and rare
                const double t = t1 - 1.0;
                if ((t >= 0.0) && (t <= (double)NRm1)) {
                    const int pidx = win_in_lds ? (sg.wbyte + 8 * wl1 - pwin_off) / (8 * kWinLds) : sg.wbyte;
                    const int qi = (kQCap > 0) ? atomicAdd(&ctl[5], 1) : kQCap;
                    if (qi < qcap) {
                        DeferredPixel e;
                        e.halo = pinfo[pidx].halo; e.abyte = sg.abyte + 8 * k; e.t = t;
                        rq[qi] = e;
                    } else { L = direct_row(pidx, t); in = true; }         // queue full: inline
                } else n_oob32 += 1;
            }
            // NaN / +-inf L paint nothing (HealpixRunner.py:473).  L already holds + ln(pixarea D^2) (:478, folded
            // into the row window by halo_row_kernel); tables with finite |ln T| > 650 never take this path, so a
            // finite L is always inside exp's range
            const bool go = in && (fabs(L) < 709.0);                       // false for NaN too
            const double v = fast_exp(L, exptab);                          // garbage when !go, never added
            if (go) __hip_atomic_fetch_add(lds_ptr<double>(lds_base + sg.abyte + 8 * k), v, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // HealpixRunner.py:336-355 for one pixel
            // (the read-out first, the pixel's geometry only where the displacement is non-zero: what is live across the read-out --
            // and across its rare direct path, a function of its own -- is the chord, not six vector components)
            const int pidx = sg.pk & 63, wl1 = sg.pk >> 12;
            const Pair &pi = pinfo[pidx];
            double sh = sin_small(h, h2), ch = cos_small(h2);              // sin, cos of dphi/2: series (h^2 <= 0.04), no square root
            if (__any(h2 > kSinSmall)) { if (h2 > kSinSmall) sincos_wide(h, sh, ch); }
            const double s2 = sh * sh;
            const double x = fma(sg.Bq, s2, sg.Aq);                        // r_com^2
            const double t1 = fma(fast_log_biased(x, logtab), t_m, t_c1 + pi.tshift);
            bool in;
            double d = window_row(sg, wl1, pidx, t1, in);                  // comoving displacement; table holds d
            if (!in) {
                const double t = t1 - 1.0;
                if ((t >= 0.0) && (t <= (double)NRm1)) { d = direct_row(pidx, t); in = true; }
                else n_oob32 += 1;
            }
            // zero outside the hull (NaN fill), beyond the model's epsilon_max R (BaryonCorrection.py:410-411),
            // for non-finite table values and at r = 0 (HealpixRunner.py:347)
            const bool use = in && (x < pi.xcut) && (x > 0.0) && (fabs(d) < 1.0e300);
            const double dcom = use ? d : 0.0;                             // comoving; physical d = dcom a (HealpixRunner.py:345)
            if (dcom != 0.0) {
                const RingRow &rr = rows[(sg.pk >> 6) & 63];
                const double sd = 2.0 * sh * ch, cd = 1.0 - 2.0 * s2;      // sin, cos of dphi
                const double cphi = pi.cp0 * cd - pi.sp0 * sd, sphi = pi.sp0 * cd + pi.cp0 * sd;
                const double vx = rr.sth * cphi, vy = rr.sth * sphi, vz = rr.z;                 // pixel unit vector
                const double dx = vx - pi.st * pi.cp0, dy = vy - pi.st * pi.sp0, dz = vz - pi.ct;   // vec - vec_j
                // new direction = (D vec + d u) / |D vec + d u|, u = (vec - vec_j) / chord (unit-sphere chord = r_com a / D); the offset
                // is vec g + (vec - vec_j) kk with g = D / |.| - 1, kk = d / (chord |.|).  With delta = d / D (~1e-3 at most) and
                // e = |.|^2 / D^2 - 1 = delta (chord + delta): g = (1 + e)^(-1/2) - 1 by its series (no cancellation, no second square
                // root, no division) and kk = dcom (1 + g) / r_com; 1 / r_com and r_com come out of ONE v_rsq_f64 + a coupled Newton
                // step.  The offsets are continuous in all of this: rounding-level differences.  Anything outside the series' range
                // (|e| >= 2^-10: d / D of a few per cent) or the lean square root's takes the plain formulas.
                double g, kk;
                const double y0 = __builtin_amdgcn_rsq(x);
                double sq = x * y0, hq = 0.5 * y0;                         // -> sqrt(x), 0.5 / sqrt(x)
                const double rq = fma(-hq, sq, 0.5);
                sq = fma(sq, rq, sq); hq = fma(hq, rq, hq);
                hq = fma(hq, fma(-hq, sq, 0.5), hq);
                const double delta = dcom * pi.a_over_D, chord = sq * pi.a_over_D;
                const double e = delta * (chord + delta);
                if ((x > 1e-280) && (x < 1e280) && (fabs(e) < 0.0009765625)) {
                    double ps = fma(e, -63.0 / 256.0, 35.0 / 128.0);
                    ps = fma(e, ps, -5.0 / 16.0);
                    ps = fma(e, ps, 3.0 / 8.0);
                    ps = fma(e, ps, -0.5);
                    g = e * ps;                                            // (1 + e)^(-1/2) - 1; next term 231/1024 e^6 < 2e-19
                    const double tq = dcom * hq;
                    kk = 2.0 * fma(tq, g, tq);
                } else {
                    const double dp = dcom * pi.a;
                    const double ch2 = sqrt(x) * pi.a_over_D;
                    const double qq = pi.D * dp * ch2 + dp * dp;           // |pos + off|^2 - D^2
                    const double nwn = sqrt(fma(pi.D, pi.D, qq));
                    g = -qq / (nwn * (nwn + pi.D));                        // D / |nw| - 1 without cancellation
                    kk = dp / (ch2 * nwn);
                }
                lds_double *ap = lds_ptr<double>(lds_base + sg.abyte + 24 * k);
                __hip_atomic_fetch_add(ap + 0, fma(vx, g, dx * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ap + 1, fma(vy, g, dy * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(ap + 2, fma(vz, g, dz * kk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

#if BFG_STAGE_TIMING
    BFG_TICK(6);                                       // prologue: LDS clear, tables, ring rows, first pair records
#endif
    BFG_ITICK(11);
    BFG_LTICK(1);
    for (int base = 0; base < n_pairs;) {
        // ---- stage a: one lane per pair of the chunk (wave 0) ---------------------------------------
        if (wave == 0) {
            const int j = nx_j;
            const bool valid = j >= 0;
            const int ra = max(nx_first, ring_lo);
            const int nrings = valid ? max(0, min(nx_last, ring_hi) - ra + 1) : 0;
            const int cum = wave_scan_incl(nrings);
            // take the longest prefix of pairs whose (pair, ring) slots fit; always at least one pair
            const bool fits = valid && (cum <= kSlotMax || lane == 0);
            const unsigned long long fm = __ballot(fits);
            const int first_bad = __ffsll((long long)~fm);          // 1-based; 0 if all 64 fit
            const int n_take = first_bad ? first_bad - 1 : 64;
            if (lane < n_take) {
                Pair &pi = pinfo[lane];
                pi.hoff = (int64_t)j * W; pi.win_lo = nx_wl; pi.halo = j; pi.ra = ra; pi.pad = 0;
                if constexpr (MODE == MODE_PAINT) pi.lnpf = nx_lnpf;   // baryonify: the rest is filled in stage b
            }
            pr_off[lane] = (lane < n_take) ? cum - nrings : 0x7fffffff;
            if (lane == n_take - 1) { ctl[0] = n_take; ctl[1] = cum; ctl[2] = 0; ctl[3] = 0; }
            // candidates of the next chunk: list entries base + n_take + lane, out of the two list windows
            const int idx = n_take + lane;
            const int fa = __shfl(pjA, idx & 63), fb = __shfl(pjB, idx & 63);
            nx_j = (lane < kPairMax) ? ((idx < 64) ? fa : fb) : -1;
        }
        lds_barrier();
        const int n_take = ctl[0], nslots = ctl[1];
        BFG_TICK(0);
        BFG_LTICK(2);
#if BFG_STAGE_TIMING == 2
        if (tid == 64) st_sub = clock64();
#endif

        // ---- stage b: one thread per (pair, ring) slot; row windows -> LDS -----------------------------
        // With full-width windows the copy is an LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous;
        // drained by the barrier that ends stage b).  dest = wave-uniform base + lane * 16 B == pwin[2 * tid].
        if constexpr (win_in_lds) {
            if (P.debug & 64) {                                       // profiling: no window copy at all (wrong results)
            } else if constexpr (BLEND) {
                // No windows in HBM: one thread per (pair, 4 nodes) blends them from the four corner rows of the halo's (z, M)
                // cell -- the table is L2-resident -- with the arithmetic of halo_row4_kernel (corner order, fma chain from
                // 0, + ln(pixarea D^2)): the same bits.  All eight 16-byte loads of a thread are in flight together.
                typedef double double2u __attribute__((ext_vector_type(2), aligned(8)));
                // from the LAST thread down: the (pair, ring) slots below are dealt from thread 0 up, so in a chunk that does not
                // fill the workgroup (sparse catalogs) the blend -- two dependent L2 round trips -- runs in wavefronts that have
                // no slot, beside the slot work instead of in front of it
                const int i = NT - 1 - tid;
                if constexpr (ROWS) {
                    if (i < n_take * (kWinLds / 4)) {
                        const int p = i >> 3, q = i & 7;
                        const Pair &pi = pinfo[p];
                        const double *r0 = T.values + (int64_t)pi.halo * T.hstride + (pi.win_lo + 4 * q);
                        const double2u a0 = *reinterpret_cast<const double2u *>(r0), b0 = *reinterpret_cast<const double2u *>(r0 + 2);
                        const double add = pi.lnpf;                    // (what halo_row4_kernel adds to a one-corner "blend": fma(v, 1, 0) + add)
                        double2 o0, o1;
                        o0.x = a0.x + add; o0.y = a0.y + add; o1.x = b0.x + add; o1.y = b0.y + add;
                        double2 *dst = reinterpret_cast<double2 *>(pwin + p * kWinLds + 4 * q);
                        dst[0] = o0; dst[1] = o1;
                    }
                } else
                if (i < n_take * (kWinLds / 4)) {
                    const int p = i >> 3, q = i & 7;
                    const Pair &pi = pinfo[p];
                    const HaloTile &hb = P.ht[pi.halo];                // the line the slot threads of this pair fetch as well
                    const int c0 = hb.ci0, c1 = hb.ci1;
                    const double y0 = hb.spare[1], y1 = hb.spare[2];
                    const double *r0 = T.values + (int64_t)c0 * T.ostride[0] + (int64_t)c1 * T.ostride[1] + (pi.win_lo + 4 * q);
                    const double *r1 = r0 + T.ostride[1], *r2 = r0 + T.ostride[0], *r3 = r2 + T.ostride[1];
                    const double2u a0 = *reinterpret_cast<const double2u *>(r0), b0 = *reinterpret_cast<const double2u *>(r0 + 2);
                    const double2u a1 = *reinterpret_cast<const double2u *>(r1), b1 = *reinterpret_cast<const double2u *>(r1 + 2);
                    const double2u a2 = *reinterpret_cast<const double2u *>(r2), b2 = *reinterpret_cast<const double2u *>(r2 + 2);
                    const double2u a3 = *reinterpret_cast<const double2u *>(r3), b3 = *reinterpret_cast<const double2u *>(r3 + 2);
                    const double w0 = (1.0 * (1.0 - y0)) * (1.0 - y1), w1 = (1.0 * (1.0 - y0)) * y1;
                    const double w2 = (1.0 * y0) * (1.0 - y1), w3 = (1.0 * y0) * y1;
                    double add = 0.0;
                    if constexpr (MODE == MODE_PAINT) add = hb.spare[0];
                    double2 o0, o1;
                    o0.x = fma(a3.x, w3, fma(a2.x, w2, fma(a1.x, w1, fma(a0.x, w0, 0.0)))) + add;
                    o0.y = fma(a3.y, w3, fma(a2.y, w2, fma(a1.y, w1, fma(a0.y, w0, 0.0)))) + add;
                    o1.x = fma(b3.x, w3, fma(b2.x, w2, fma(b1.x, w1, fma(b0.x, w0, 0.0)))) + add;
                    o1.y = fma(b3.y, w3, fma(b2.y, w2, fma(b1.y, w1, fma(b0.y, w0, 0.0)))) + add;
                    double2 *dst = reinterpret_cast<double2 *>(pwin + p * kWinLds + 4 * q);
                    dst[0] = o0; dst[1] = o1;
                }
            } else if (W == kWinLds) {
                for (int i = tid; i - lane < n_take * (kWinLds / 2); i += NT) {          // whole wavefronts step together
                    if (i < n_take * (kWinLds / 2)) {
                        const double *src = P.hwin + pinfo[i / (kWinLds / 2)].hoff + 2 * (i % (kWinLds / 2));
                        double *dst = pwin + 2 * (i - lane);                              // wave-uniform; lane * 16 B added by HW
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                    }
                }
            } else {
                for (int idx = tid; idx < n_take * W; idx += NT) {
                    const int p = idx / W, e = idx - p * W;
                    pwin[p * kWinLds + e] = P.hwin[pinfo[p].hoff + e];
                }
            }
        }
        if constexpr (!win_in_lds) {
            if (P.win_table && wave == kTileWaves - 1 && lane < n_take) {   // corner rows of the pair's halo
                const int64_t j = pinfo[lane].halo;
                const int ncorner = 1 << T.nouter;
                double *cwn = pwin + lane * kWinLds;
                int64_t *con = reinterpret_cast<int64_t *>(cwn + ncorner);
                for (int c = 0; c < ncorner; ++c) {                        // corner order and products of halo_row_kernel
                    double w = 1.0;
                    int64_t off = j * (int64_t)T.hstride;
                    for (int k = 0; k < T.nouter; ++k) {
                        const int bit = (c >> (T.nouter - 1 - k)) & 1;
                        const double y = halo_cell_weight(P.ht, P.cw, P.cap, T.nouter, j, k);
                        w = w * (bit ? y : 1.0 - y);
                        off += (int64_t)(halo_cell_index(P.ht, P.cidx, P.cap, T.nouter, j, k) + bit) * T.ostride[k];
                    }
                    cwn[c] = w; con[c] = off;
                }
                if constexpr (MODE == MODE_PAINT) { if (ncorner == 4) cwn[8] = pinfo[lane].lnpf; }   // the fast path's record
            }
        }
        if constexpr (MODE == MODE_BARYONIFY) {
            static_assert(MODE != MODE_BARYONIFY || kSlotMax <= NT - 64, "the last wavefront has no slots");
            if (wave == kTileWaves - 1 && lane < n_take) {              // per-pair constants of the pixel stage
                Pair &pi = pinfo[lane];
                const int j = pi.halo;
                const HaloTile &h = P.ht[j];
                const HaloDisp &hd = P.hd[j];
                pi.cp0 = hd.cp0; pi.sp0 = hd.sp0; pi.st = h.st; pi.ct = h.ct;
                pi.a = hd.a; pi.D = hd.D; pi.xcut = hd.xcut; pi.tshift = hd.tshift; pi.a_over_D = hd.a / hd.D;
            }
        }
        if (wave * 64 < nslots && !(P.debug & 8)) {                 // whole wavefronts: kSlotMax <= NT, one pass
            const int slot = tid;
            const bool live = slot < nslots;
            // pair p with pr_off[p] <= slot < pr_off[p+1]: one LDS read, then a wave-uniform loop over the pairs
            // that start inside this wavefront's 64 slots
            int p;
            {
                const int sb = wave * 64;
                const int myoff = pr_off[lane];
                p = __popcll(__ballot(myoff <= sb)) - 1;
                unsigned long long inside = __ballot(myoff > sb && myoff <= sb + 63);
                while (inside) {
                    const int kk = __ffsll((long long)inside) - 1;
                    inside &= inside - 1;
                    const int o = __builtin_amdgcn_readlane(myoff, kk);
                    p += (slot >= o) ? 1 : 0;
                }
            }
            BFG_SUBTICK(0);                                          // window DMA issued, pair of the slot found
            int cnt1 = 0, cnt2 = 0, aa1 = 0, aa2 = 0, ab1 = 0, ab2 = 0;
            Seg sg;
            sg.excl = 0; sg.abyte = 0; sg.wbyte = 0; sg.pk = 0; sg.hstep = 0; sg.c0 = 0; sg.Aq = 0; sg.Bq = 0;
            if (live) {
                const int j = pinfo[p].halo;
                const int ring = pinfo[p].ra + (slot - pr_off[p]);
                const int row = ring - ring_lo;
                const RingRow rr = rows[row];
                const HaloTile &h = P.ht[j];
                const double st = h.st, ct = h.ct, pphi = h.pphi, S = h.S;
                const int nr = rr.nr;
                // query_disc ring window.  Evaluated unconditionally so that every field of the halo record is
                // fetched in one burst (a branch on irmin / irmax first serialises three L2 round trips); rings that
                // lie entirely inside the disc (ring outside [irmin, irmax]) override the result.
                const int irmin = h.irmin, irmax = h.irmax;
#if BFG_STAGE_TIMING == 2
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BFG_SUBTICK(1);                                      // halo record arrived
#endif
                int lo = 0, cnt = 0;
                {
                    const double x = (h.cosr - rr.z * h.z0) * h.xa;
                    const double ysq = 1.0 - rr.z * rr.z - x * x;
                    const double dphi = (ysq > 0.0) ? atan2_upper_tab(sqrt_unit(ysq), x, atantab) : 0.0;
                    if (dphi > 0.0) {
                        const double shift = (rr.phioff != 0.0) ? 0.5 : 0.0;
                        // |nr (pphi -+ dphi) / 2 pi| < 2 nr <= 8 nside: 32-bit is enough (the tile variant needs nside <= 2^24)
                        const double fn = (double)nr * kInvTwoPi;
                        const int l32 = (int)floor(fn * (pphi - dphi) - shift) + 1;
                        const int h32 = (int)floor(fn * (pphi + dphi) - shift);
                        const int c = min(h32 - l32 + 1, nr);
                        if (c > 0) { cnt = c; lo = l32; }                // unwrapped: lo in (-nr, 1.5 nr)
                    }
                }
                if (ring < irmin || ring > irmax) { cnt = nr; lo = 0; }  // ring entirely inside the disc
                const int wl = pinfo[p].win_lo;
                // window start + 1 and the LDS address of node (cell - 1): see window_row
                sg.wbyte = win_in_lds ? pwin_off + p * (kWinLds * 8) - 8 * (wl + 1) : p;
                if constexpr (MODE == MODE_PAINT) sg.pk = wl + 1;
                else sg.pk = p | (row << 6) | ((wl + 1) << 12);
                sg.hstep = 0.5 * rr.phistep;
                sg.c0 = 0.5 * (rr.phioff * rr.phistep - pphi);          // + first pixel * hstep, below
                const double ds = rr.sth - st, dz = rr.z - ct;
                sg.Aq = (ds * ds + dz * dz) * S;
                sg.Bq = 4.0 * rr.sth * st * S;
                // clip to the sector; a window that wraps around the ring can meet it twice
                if (cnt > 0) {
                    const bool wraps = (lo < 0) || (lo + cnt > nr);
                    if (!__any(wraps)) {                                // the usual case for a whole wavefront
                        const int aa = max(lo, rr.k0), bb = min(lo + cnt, rr.k1);
                        if (bb > aa) { cnt1 = bb - aa; aa1 = aa; ab1 = acc_off + 8 * NACC * (rr.rowoff + aa); }
                    } else {
#pragma unroll
                        for (int mi = 0; mi < 3; ++mi) {
                            const int m = (mi == 0) ? 0 : (mi == 1 ? -1 : 1);
                            const int aa = max(lo, rr.k0 + m * nr), bb = min(lo + cnt, rr.k1 + m * nr);
                            if (bb > aa) {
                                const int ab = acc_off + 8 * NACC * (rr.rowoff - m * nr + aa);
                                if (cnt1 == 0) { cnt1 = bb - aa; aa1 = aa; ab1 = ab; }
                                else if (cnt2 == 0) { cnt2 = bb - aa; aa2 = aa; ab2 = ab; }
                            }
                        }
                    }
                }
            }
            BFG_SUBTICK(2);                                          // ring window, clipping, segment constants
            // a second piece needs its own segment record; should the chunk run out of them (only possible where a
            // sector spans a whole ring, at the poles) the piece is painted right here, through the direct read-out
            int idx2 = 0;
            bool spill2 = false;
            if (cnt2 > 0) { idx2 = nslots + atomicAdd(&ctl[2], 1); spill2 = idx2 >= kSegMax || (P.debug & 16); }   // debug bit 16: force the spill path (tests)
            // offsets in the chunk's flattened pixel list: wave scan + one LDS atomic per wavefront (any order will do)
            const int tot = cnt1 + (spill2 ? 0 : cnt2);
            const int incl = wave_scan_incl(tot);
            const int wtot = __builtin_amdgcn_readlane(incl, 63);
            int wbase = 0;
            if (lane == 0 && wtot > 0) wbase = atomicAdd(&ctl[3], wtot);
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            const int e1 = wbase + incl - tot;
            BFG_SUBTICK(3);                                          // pixel-list offsets (scan + LDS atomics)
            if (live) {
                scnt[slot] = (uint8_t)cnt1;
                if (cnt1 > 0) {
                    const double c0b = sg.c0;
                    sg.excl = e1; sg.abyte = ab1; sg.c0 = fma((double)aa1, sg.hstep, c0b);
                    segs[slot] = sg;
                    fill_ptab(e1, min(e1 + cnt1, kPixMax), (uint32_t)(slot * (int)sizeof(Seg)));
                    if (cnt2 > 0 && !spill2) {
                        scnt[idx2] = (uint8_t)cnt2;
                        const int e2 = e1 + cnt1;
                        sg.excl = e2; sg.abyte = ab2; sg.c0 = fma((double)aa2, sg.hstep, c0b);
                        segs[idx2] = sg;
                        fill_ptab(e2, min(e2 + cnt2, kPixMax), (uint32_t)(idx2 * (int)sizeof(Seg)));
                    } else if (cnt2 > 0) {
                        constexpr int wlf = 1 << 18;                   // a window no cell falls into -> direct_row()
                        sg.excl = 0; sg.abyte = ab2; sg.c0 = fma((double)aa2, sg.hstep, c0b);
                        sg.wbyte = win_in_lds ? pwin_off + p * (kWinLds * 8) - 8 * wlf : p;
                        if constexpr (MODE == MODE_PAINT) sg.pk = wlf;
                        else sg.pk = (sg.pk & 0xFFF) | (wlf << 12);
                        for (int k = 0; k < cnt2; ++k) do_pixel(k, sg);
                        my_pixels += (unsigned long long)cnt2;
                    }
                }
            }
        }
        BFG_SUBTICK(4);                                              // segment records + pixel -> segment table written
        BFG_LTICK(3);
        __syncthreads();
        BFG_LTICK(4);
        BFG_SUBTICK(5);                                              // wait for the other wavefronts and the window DMA
        BFG_TICK(1);
        const int nseg = min(nslots + ctl[2], kSegMax);
        const int ptotal = (P.debug & 2) ? 0 : ctl[3];
        my_pixels += (tid == 0) ? (unsigned long long)ctl[3] : 0ull;
        // in flight during the pixel stage: the records of the next chunk's candidates and the list windows after them -- or,
        // in the item's last chunk, the first list windows of the NEXT item (its records follow after the pixel stage)
        const bool last_chunk = base + n_take >= n_pairs;
        if (wave == 0 && !last_chunk) { load_records(); load_list_windows(base + n_take); }
        BFG_TICK(2);

        // ---- stage c: one thread per pixel of the flattened list (rounds of kPixMax pixels; the first round's
        //      pixel -> segment table was filled by stage b) -------------------------------------------------
        for (int pbase = 0; pbase < ptotal; pbase += kPixMax) {
            if (pbase > 0) {                                          // rare: refill the table for the next round
                lds_barrier();
                for (int sidx = tid; sidx < nseg; sidx += NT) {
                    const int cnt = scnt[sidx];
                    if (cnt > 0) {
                        const int ex = segs[sidx].excl;
                        const int a0 = max(ex, pbase), a1 = min(ex + cnt, pbase + kPixMax);
                        for (int q = a0; q < a1; ++q) ptab[q - pbase] = (uint16_t)(sidx * (int)sizeof(Seg));
                    }
                }
                lds_barrier();
            }
            BFG_TICK(3);
            const int pend = min(ptotal, pbase + kPixMax);
            const uint16_t *pp = ptab + tid;
            // by value: one burst of LDS reads; an absolute LDS address (the block starts at 0, checked above) -- through
            // smem_raw the compiler adds the block's relocatable base to the loaded offset, one more VALU instruction per pixel
            auto load_seg = [&](const uint16_t *pq) -> Seg {
                typedef int v4i_t __attribute__((ext_vector_type(4)));
                typedef double v2d_t __attribute__((ext_vector_type(2)));
                const unsigned sa = (unsigned)segs_off + (unsigned)*pq;
                const v4i_t s0 = *lds_ptr<const v4i_t>(sa);
                const v2d_t s1 = *lds_ptr<const v2d_t>(sa + 16), s2 = *lds_ptr<const v2d_t>(sa + 32);
                Seg sg;
                sg.excl = s0.x; sg.abyte = s0.y; sg.wbyte = s0.z; sg.pk = s0.w; sg.hstep = s1.x; sg.c0 = s1.y; sg.Aq = s2.x; sg.Bq = s2.y;
                return sg;
            };
            for (int q = pbase + tid; q < pend; q += NT, pp += NT) {
                const Seg sg = load_seg(pp);
                do_pixel(q, sg);
            }
        }
        if (wave == 0 && last_chunk && have_next) {     // the next item's first candidates: their records arrive during the write-back
            pjA = pjA1; pjB = pjB1;
            nx_j = (lane < kPairMax) ? pjA : -1;
            load_records();
        }
        handed = handed || (last_chunk && have_next);
        BFG_TICK(4);
        BFG_LTICK(5);
        lds_barrier();
        BFG_TICK(5);
        BFG_LTICK(6);
        base += n_take;
        if constexpr (kQCap > 0) { if (ctl[5] >= qcap / 2) drain(); }     // uniform: ctl[5] is stable between the barriers
    }
    // write the tile back: every pixel belongs to exactly one tile -> plain read-modify-write (unless the tile's pair list
    // was cut into several work items), or plain stores when the caller vouches for a cleared map (out_zero).  Each thread
    // owns TR * TW / NT pixels; their map values are fetched in one burst.  Deferred pixels still in the queue are not
    // drained here (two dependent rounds of global loads, 33 % of the kernel at 1e5 halos): they go to this work item's
    // slice of a global list and tile_deferred_kernel adds them after this kernel.
    BFG_ITICK(12);
    // the look-ahead of the persistent loop goes to LDS BEFORE the write-back stores (see the hand-over above)
    if (tid == 0) ctl[6] = P.work_counter ? item_first + item3 * n_counters + my_c : kNoItem;
    constexpr int kPerThread = (TR * TW + NT - 1) / NT;
    const bool shared = wk.w && !degraded;
    const bool rmw = !shared && !p_out_zero && !p_overwrite;
    // Overwrite mode, tile not shared (the runners' and the bench's case): plain stores of every pixel and NOT ONE vector load in
    // this stretch of code -- a load here (the read-modify-write path's) leaves the compiler a pending destination register to
    // protect with s_waitcnt vmcnt(0) after the stores, i.e. a wait for their acknowledgement at the end of every item.
    const bool plain = p_overwrite && !shared;
    if (plain && !(P.debug & 128)) {
        if constexpr (kQCap > 0) {
            const int n = min(ctl[5], qcap);
            if (P.defer && dfill + n <= P.defer_cap_wg) {
                if (tid < n) {
                    const DeferredPixel e = rq[tid];
                    const int i = (e.abyte - acc_off) >> 3;                // accumulator index = row * TW + column
                    const int row = i / TW, col = i % TW;
                    DeferredOut o;
                    o.pix = rows[row].start + rows[row].k0 + col; o.t = e.t; o.halo = e.halo; o.pad[0] = o.pad[1] = o.pad[2] = 0;
                    P.defer[(size_t)blockIdx.x * P.defer_cap_wg + dfill + tid] = o;
                }
                dfill += n;
            } else if (n > 0) drain();                                     // no list, or this workgroup's slice is full
        }
        // (the thread index through an opaque asm: otherwise the compiler hoists the per-u row addresses out of the persistent
        // loop, spills some of them, and reloads them here with s_waitcnt vmcnt(0) -- between the stores, i.e. waiting for their
        // acknowledgement after all)
        BFG_LTICK(9);
        unsigned wt = (unsigned)tid;
        asm volatile("" : "+v"(wt));
        // all LDS reads of the thread's pixels first (ring rows and accumulator values in flight together, one wait), then the
        // stores: read -> wait -> read -> wait per pixel cost ~3 LDS round trips x 4 pixels on the critical path of every item.
        // (Tried: wave 0, whose loads pace the next item, issuing no stores at all -- no change: profiles/r03_writeback_ab.txt.)
        if constexpr (NACC == 1 && TW % 2 == 0 && (TR * TW / 2) % NT == 0) {
        // paint: two neighbouring pixels of a ring per thread -- one 16-byte LDS read and one 16-byte store (8-byte aligned: the
        // row's first pixel may be odd) instead of two of each: half the store and LDS-read instructions of the write-back
        typedef double wb_v2d __attribute__((ext_vector_type(2), aligned(8)));
        constexpr int kPP = TR * TW / 2 / NT;
        int pk0[kPP], pk1[kPP];
        int64_t pst[kPP];
        double2 pv[kPP];
#pragma unroll
        for (int u = 0; u < kPP; ++u) {
            const unsigned i = 2u * (wt + (unsigned)(u * NT));
            const RingRow &rr = rows[i / TW];
            pk0[u] = rr.k0; pk1[u] = rr.k1; pst[u] = rr.start;
            pv[u] = *reinterpret_cast<const double2 *>(acc + i);
        }
        BFG_LTICK(10);
#pragma unroll
        for (int u = 0; u < kPP; ++u) {
            const unsigned i = 2u * (wt + (unsigned)(u * NT));
            const int row = (int)(i / TW), col = (int)(i % TW);
            if (ring_lo + row <= ring_hi && pk0[u] + col < pk1[u]) {
                double *dst = P.out + (pst[u] + pk0[u] + col);
                if (pk0[u] + col + 1 < pk1[u]) { wb_v2d v; v.x = pv[u].x; v.y = pv[u].y; *reinterpret_cast<wb_v2d *>(dst) = v; }
                else dst[0] = pv[u].x;
            }
        }
        } else {
        // (four pixels per thread at a time: the 256-thread instantiations own eight, and eight sets of these spill)
        constexpr int kGrp = kPerThread < 4 ? kPerThread : 4;
        static_assert(kPerThread % kGrp == 0, "write-back groups");
#pragma unroll 1
        for (int u0 = 0; u0 < kPerThread; u0 += kGrp) {
        int wk0[kGrp], wk1[kGrp];
        int64_t wst[kGrp];
        double wv[kGrp][NACC];
#pragma unroll
        for (int u = 0; u < kGrp; ++u) {
            const unsigned i = min(wt + (unsigned)((u0 + u) * NT), (unsigned)(TR * TW - 1));
            const RingRow &rr = rows[i / TW];
            wk0[u] = rr.k0; wk1[u] = rr.k1; wst[u] = rr.start;
#pragma unroll
            for (int c = 0; c < NACC; ++c) wv[u][c] = acc[NACC * i + c];
        }
        BFG_LTICK(10);
#pragma unroll
        for (int u = 0; u < kGrp; ++u) {
            const unsigned i = wt + (unsigned)((u0 + u) * NT);
            const int row = (int)(i / TW), col = (int)(i % TW);
            if (i < (unsigned)(TR * TW) && ring_lo + row <= ring_hi && wk0[u] + col < wk1[u]) {
                double *dst = P.out + NACC * (wst[u] + wk0[u] + col);
#pragma unroll
                for (int c = 0; c < NACC; ++c) dst[c] = wv[u][c];
            }
        }
        }
        }
    } else if (!(P.debug & 128)) {                     // profiling: bit 128 skips the write-back (wrong results)
    if constexpr (kQCap > 0) {
        const int n = min(ctl[5], qcap);
        if (P.defer && dfill + n <= P.defer_cap_wg) {
            if (tid < n) {
                const DeferredPixel e = rq[tid];
                const int i = (e.abyte - acc_off) >> 3;                    // accumulator index = row * TW + column
                const int row = i / TW, col = i % TW;
                DeferredOut o;
                o.pix = rows[row].start + rows[row].k0 + col; o.t = e.t; o.halo = e.halo; o.pad[0] = o.pad[1] = o.pad[2] = 0;
                P.defer[(size_t)blockIdx.x * P.defer_cap_wg + dfill + tid] = o;
            }
            dfill += n;
        } else if (n > 0) drain();                                         // no list, or this workgroup's slice is full
    }
    BFG_ITICK(13);
    // four pixels per thread at a time (their old map values are fetched in one burst); the 256-thread instantiations own eight
    constexpr int kGrp = kPerThread < 4 ? kPerThread : 4;
    static_assert(kPerThread % kGrp == 0, "write-back groups");
#pragma unroll 1
    for (int u0 = 0; u0 < kPerThread; u0 += kGrp) {
        int64_t wpix[kGrp];
        double wold[kGrp][NACC];
#pragma unroll
        for (int u = 0; u < kGrp; ++u) {
            const int i = tid + (u0 + u) * NT;
            wpix[u] = -1;
            if (i < TR * TW) {
                const int row = i / TW, col = i % TW;
                const int ring = ring_lo + row;
                if (ring <= ring_hi && rows[row].k0 + col < rows[row].k1) wpix[u] = rows[row].start + rows[row].k0 + col;
            }
#pragma unroll
            for (int c = 0; c < NACC; ++c) wold[u][c] = (wpix[u] >= 0 && rmw) ? P.out[NACC * wpix[u] + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kGrp; ++u) {
            if (wpix[u] < 0) continue;
            const int i = tid + (u0 + u) * NT;
#pragma unroll
            for (int c = 0; c < NACC; ++c) {
                const double v = acc[NACC * i + c];
                if (shared) { if (v != 0.0) unsafeAtomicAdd(P.out + NACC * wpix[u] + c, v); }   // the tile is shared with other workgroups
                else if (p_overwrite) P.out[NACC * wpix[u] + c] = v;             // every pixel, zeros included: the map was not cleared
                else if (v != 0.0) P.out[NACC * wpix[u] + c] = wold[u][c] + v;
            }
        }
    }
    }
    px_total += my_pixels;
    oob_total += n_oob32;
    BFG_ITICK(14);
    BFG_LTICK(7);
#if BFG_STAGE_TIMING && BFG_STAGE_TIMING != 4
#if BFG_STAGE_TIMING != 3
    __syncthreads();
#endif
    BFG_TICK(7);                                       // epilogue: final queue drain, write-back
#endif
    } else {
        // an item without pairs (a tile no halo touches; every item when the binning gave up): nothing to paint.  An
        // uninitialised map still gets its zeros (tiles shared between items were cleared by tile_fill_kernel).
        if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the record DMA has landed (no stage-b barrier here)
        if (tid == 0) ctl[6] = P.work_counter ? item_first + item3 * n_counters + my_c : kNoItem;
        if (p_overwrite && !(wk.w && !degraded)) {
            for (int i = tid; i < TR * TW; i += NT) {
                const int row = i / TW, col = i % TW;
                const int ring = ring_lo + row;
                if (ring > ring_hi) continue;
                int64_t start, nr64; bool shifted;
                ring_info_small(hp, ring, start, nr64, shifted);
                const int k0 = (int)(((int64_t)sector * nr64) / NS), k1 = (int)(((int64_t)(sector + 1) * nr64) / NS);
                if (k0 + col < k1) {
#pragma unroll
                    for (int c = 0; c < NACC; ++c) P.out[NACC * (start + k0 + col) + c] = 0.0;
                }
            }
        }
    }
    // on to the next work item: every thread is done with the accumulator, the ring rows and the queue
    lds_barrier();
    BFG_ITICK(15);
    BFG_LTICK(8);
    primed = handed;                                  // (an item without pairs has no last chunk to hand over in)
    item = item1; item1 = item2; item2 = uni(ctl[6]);
    wk = wk1; wg = wg1;
    wk1 = uni4(*reinterpret_cast<const int4 *>(ctl + 8 + 8 * par)); wg1 = uni4(*reinterpret_cast<const int4 *>(ctl + 12 + 8 * par));
    par ^= 1;
    }   // work items
    // The deferred pixels of this workgroup's items.  defer_tail: added here, by the workgroup that queued them -- its write-back
    // stores of the same pixels have been acknowledged (the barrier waits for vmcnt(0)), other workgroups touch these pixels
    // with atomics only (shared tiles); the entries are read back past the L1 (volatile: other wavefronts wrote them).  Else
    // tile_deferred_kernel adds them after this kernel (BFG_FINAL_DRAIN=kernel: the A/B, 0.015 ms of launch and tail).
    if constexpr (kQCap > 0) {
        if (P.defer) {
            if (P.defer_tail) {
                // Every wavefront's write-back stores and list entries have been acknowledged by the L2 -- an explicit
                // s_waitcnt vmcnt(0), not an assumption about what __syncthreads() happens to wait for -- before any wavefront of
                // this workgroup adds to those pixels (atomics, resolved in that same L2: one CU, one XCD) or reads those entries
                // (volatile: past the L1).  (An agent-scope release / acquire fence pair is NOT what is wanted here: on this part
                // it writes back and invalidates the whole L2 -- the 1e5-halo tile kernel went from 0.151 to 0.226 ms with it.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const volatile DeferredOut *slice = P.defer + (size_t)blockIdx.x * P.defer_cap_wg;
                for (int i = tid; i < dfill; i += NT) {
                    DeferredOut e;
                    e.pix = slice[i].pix; e.t = slice[i].t; e.halo = slice[i].halo;
                    deferred_add<(BLEND && !ROWS)>(P, e, exptab);
                }
            } else if (tid == 0) P.defer_count[blockIdx.x] = dfill;
        }
    }
    // counters last, once per workgroup: nothing waits for these atomics
    if (px_total) atomicAdd((unsigned long long *)&P.stats->pixel_updates, px_total);
    if (oob_total) {
        atomicAdd((unsigned long long *)&P.stats->pixels_out_of_table, (unsigned long long)oob_total);
        atomicOr(&P.stats->warn_mask, BFG_WARN_R_RANGE);
    }
#if BFG_STAGE_TIMING && BFG_STAGE_TIMING != 4
    if (tid == 64) for (int i = 0; i < 16; ++i) atomicAdd(&g_stage_cycles[i], (unsigned long long)st_acc[i]);
#endif
#if BFG_STAGE_TIMING == 4
    __syncthreads();
    if (tid < 16) atomicAdd(&g_stage_cycles[tid], reinterpret_cast<unsigned long long *>(smem_raw + lt_off)[tid]);
#endif
}

// The deferred pixels the paint tile workgroups left in their slices, when they do not add them themselves (defer_tail = 0):
// one wavefront per slice, striding over the slices of the tile kernel's grid.
__global__ __launch_bounds__(256) void tile_deferred_kernel(const TileParams P, int n_slices)
{
    if ((long long)P.tile_start[P.geo.ntiles] > P.pair_cap) return;
    const int lane = threadIdx.x & 63;
    for (int w = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); w < n_slices; w += (int)gridDim.x * 4) {
        const int n = P.defer_count[w];
        for (int i = lane; i < n; i += 64) deferred_add(P, P.defer[(size_t)w * P.defer_cap_wg + i], P.exptab);
    }
}

}  // namespace bfg
