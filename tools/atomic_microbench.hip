// atomic_microbench.hip -- measured ceilings for the scatter-add design (SURVEY.md 8d asks for the
// f64-atomic rate next to the roofline number):
//   global f64 atomic add throughput for the access shapes the paint kernel produces
//   (runs of SEG consecutive doubles at pseudo-random map offsets), LDS f64 atomic add throughput,
//   and plain f64 store throughput of the same shape for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_microbench.hip -o tools/atomic_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// every group of SEG lanes adds to SEG consecutive doubles at a pseudo-random offset; ITER rounds
template <int SEG, int MODE>   // MODE 0: global atomic, 1: plain store
__global__ __launch_bounds__(256) void global_kernel(double *map, uint32_t npix, int iters, uint32_t spread)
{
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const uint32_t l = threadIdx.x % SEG;
    // groups of one workgroup stay within `spread` pixels of the workgroup's base (spatially sorted halos)
    const uint32_t base = hash32(blockIdx.x * 2654435761u) % (npix - spread - 64 * 1024);
    for (int it = 0; it < iters; ++it) {
        uint32_t off = base + hash32(gid * 9781u + it * 6271u) % spread;
        if (MODE == 0) unsafeAtomicAdd(map + off + l, 1.0);
        else map[off + l] = 1.0;
    }
}

template <int SEG>
__global__ __launch_bounds__(256) void lds_kernel(double *out, int iters, int tile)
{
    extern __shared__ double t[];
    for (int i = threadIdx.x; i < tile; i += 256) t[i] = 0.0;
    __syncthreads();
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / SEG;
    const uint32_t l = threadIdx.x % SEG;
    for (int it = 0; it < iters; ++it) {
        uint32_t off = hash32(gid * 9781u + it * 6271u) % (tile - SEG);
        unsafeAtomicAdd(t + off + l, 1.0);
    }
    __syncthreads();
    double s = 0;
    for (int i = threadIdx.x; i < tile; i += 256) s += t[i];
    if (s == -1.0) out[0] = s;
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const uint32_t npix = 12582912;   // NSIDE 1024
    double *map; CHECK(hipMalloc(&map, (size_t)npix * 8)); CHECK(hipMemset(map, 0, (size_t)npix * 8));
    double *out; CHECK(hipMalloc(&out, 64));
    const int blocks = 256 * 16, iters = 256;
    const double n_ops = (double)blocks * 256 * iters;
    printf("global f64 atomics / stores: %d blocks x 256 thr x %d iters = %.3g ops per launch\n", blocks, iters, n_ops);
    for (uint32_t spread : {4096u * 64u, 4096u * 1024u, npix - 128 * 1024}) {
        printf(" spread %u pixels per workgroup\n", spread);
#define RUN(SEG) { \
        double ms = time_ms([&] { hipLaunchKernelGGL((global_kernel<SEG, 0>), dim3(blocks), dim3(256), 0, 0, map, npix, iters, spread); }, 5); \
        double ms2 = time_ms([&] { hipLaunchKernelGGL((global_kernel<SEG, 1>), dim3(blocks), dim3(256), 0, 0, map, npix, iters, spread); }, 5); \
        printf("  SEG %2d: atomic %.3f ms  %.3g adds/s  (%.2f TB/s of added bytes) | store %.3f ms %.3g /s\n", SEG, ms, n_ops / ms * 1e3, n_ops * 8 / ms * 1e3 / 1e12, ms2, n_ops / ms2 * 1e3); }
        RUN(64) RUN(32) RUN(16) RUN(8) RUN(4)
    }
    printf("LDS f64 atomics (ds_add_f64), 32 KB tile, 1024 blocks (4/CU):\n");
#define RUNL(SEG) { \
        double ms = time_ms([&] { hipLaunchKernelGGL((lds_kernel<SEG>), dim3(1024), dim3(256), 32768, 0, out, 4096, 4096); }, 5); \
        printf("  SEG %2d: %.3f ms  %.3g adds/s\n", SEG, ms, 1024.0 * 256 * 4096 / ms * 1e3); }
    RUNL(64) RUNL(16) RUNL(4)
    return 0;
}
