// Scattered 32-bit atomic increments on a small counter array (the tile-binning pattern: 4.4e6 increments on 6272
// counters from threads in random order): agent scope (what atomicAdd gives; coherent across the 8 XCDs) vs
// workgroup scope (resolved in the issuing XCD's L2; only valid if every counter copy is private to one XCD).
// hipcc --offload-arch=gfx950 -O3 -o tools/atomic_scope_microbench tools/atomic_scope_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int SCOPE, bool RET>
__global__ void bump(int *cnt, const uint32_t *idx, int per_thread, int n_threads, int copies, int ncnt, int *sink, int stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    // XCC id of this workgroup (HW_REG_XCC_ID = 20, bits 3:0) selects the private copy when copies == 8
    const int xcc = (copies > 1) ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7) : 0;
    int acc = 0;
    for (int k = 0; k < per_thread; ++k) {
        int *p = cnt + ((size_t)xcc * ncnt + idx[(size_t)t * per_thread + k]) * stride;
        if (RET) acc += __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
        else (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == -12345) *sink = acc;
}

int main()
{
    const int ncnt = 6272, n_threads = 1000000, per_thread = 4;
    std::vector<uint32_t> h((size_t)n_threads * per_thread);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % ncnt); }
    uint32_t *d_idx; int *d_cnt, *d_sink;
    hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_cnt, (size_t)8 * ncnt * 4 * 32); hipMalloc(&d_sink, 4);
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, auto kern, int copies, int stride = 1) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(d_cnt, 0, (size_t)8 * ncnt * 4 * 32);
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3((n_threads + 255) / 256), dim3(256), 0, 0, d_cnt, d_idx, per_thread, n_threads, copies, ncnt, d_sink, stride);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        std::vector<int> c((size_t)8 * ncnt * 32);
        hipMemcpy(c.data(), d_cnt, c.size() * 4, hipMemcpyDeviceToHost);
        long long tot = 0; for (int v : c) tot += v;
        printf("%-44s %7.3f ms  %.3e atomics/s  sum %lld (expect %lld)\n", name, best, (double)n_threads * per_thread / (best * 1e-3), tot,
               (long long)n_threads * per_thread);
    };
    run("agent scope, no return, 1 copy", bump<__HIP_MEMORY_SCOPE_AGENT, false>, 1);
    run("agent scope, return,    1 copy", bump<__HIP_MEMORY_SCOPE_AGENT, true>, 1);
    run("agent scope, no return, per-XCC copies", bump<__HIP_MEMORY_SCOPE_AGENT, false>, 8);
    run("workgroup scope, no return, per-XCC copies", bump<__HIP_MEMORY_SCOPE_WORKGROUP, false>, 8);
    run("workgroup scope, return,    per-XCC copies", bump<__HIP_MEMORY_SCOPE_WORKGROUP, true>, 8);
    run("agent scope, no return, 1 copy, 1 counter per 32 B", bump<__HIP_MEMORY_SCOPE_AGENT, false>, 1, 8);
    run("agent scope, no return, 1 copy, 1 counter per 128 B", bump<__HIP_MEMORY_SCOPE_AGENT, false>, 1, 32);
    run("agent scope, return,    1 copy, 1 counter per 128 B", bump<__HIP_MEMORY_SCOPE_AGENT, true>, 1, 32);
    return 0;
}
