// tools/membench.hip -- what the chip sustains for the pileup pass's memory shape (diagnostic, not product code).
//
// Round 4 rewrite (the old tool issued its reads as a dependent chain with a 64-bit modulo per load in front of
// its stores, and its 10 % swing with the grid shape was never explained).  Now:
//   * every wave streams its own contiguous piece: per step it STORES `rows` x 1 KiB (int4 per lane, what a pileup wave writes
//     per row) and LOADS `reads` x 256 B (one dword per lane: the interval columns), the loads of step i+1 issued before the
//     stores of step i and consumed one step later (prefetched: no load is waited for right after its issue);
//   * addresses advance by constants (no division, no modulo);
//   * workgroups of 64 .. 256 threads, 1 .. 8 waves per SIMD, three mixes: store-only, load-only, the int32 pass's mix
//     (12 B read per 0.147 windows + 4 B per window written: 0.44 B read per B written), the byte pass's mix (1 B per window).
// usage: membench [GiB of stores] [allocation mode] [chunk MiB] [n buffers]; prints GB/s per configuration (best of 4 launches).
// Allocation modes -- what a stream gets depends on where its buffers lie (profiles/r04_membench_placement.txt): 0 hipMalloc,
// 1 hipDeviceMallocContiguous, 2 / 3 hipMemCreate chunks taken one after the other and mapped in shuffled / creation order,
// 4 + k: (k + 2) times the chunks created, every (k + 2)-th kept (10: every eighth -- engine.hip DevBuf).  With [n buffers] only
// the store-only stream runs, into n buffers allocated one after the other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int RD_PER_ROW_X16>      // loads per 16 stored rows (0: store-only)
__global__ __launch_bounds__(256) void stream_kernel(int4 *out, const int *in, long long rows_per_wave, long long in_per_wave, int *sink, int do_store)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int4 *o = out + wave * rows_per_wave * 64 + lane;
    const int *p = in + wave * in_per_wave + lane;
    int acc = 0;
    int nxt[RD_PER_ROW_X16 > 0 ? RD_PER_ROW_X16 : 1];
#pragma unroll
    for (int r = 0; r < RD_PER_ROW_X16; ++r) nxt[r] = p[r * 64];
    p += RD_PER_ROW_X16 * 64;
    for (long long row = 0; row + 16 <= rows_per_wave; row += 16) {
        int cur[RD_PER_ROW_X16 > 0 ? RD_PER_ROW_X16 : 1];
#pragma unroll
        for (int r = 0; r < RD_PER_ROW_X16; ++r) { cur[r] = nxt[r]; nxt[r] = p[r * 64]; }
        p += RD_PER_ROW_X16 * 64;
#pragma unroll
        for (int r = 0; r < RD_PER_ROW_X16; ++r) acc += cur[r];
        if (do_store) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(row + r) * 64] = make_int4(acc, r, lane, (int)row);
        }
    }
    if (acc == 0x7fffffff) *sink = acc;
}

template <int RD>
static void run(const char *name, int4 *out, int *in, double gib, int *sink, int do_store, double bytes_per_row_read)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256})
        for (int wps : {1, 2, 4, 6, 8}) {
            const long long waves = 256LL * 4 * wps;
            const long long total_rows = (long long)(gib * (1LL << 30)) / 1024;
            const long long rows_per_wave = (total_rows / waves) / 16 * 16;
            const long long in_per_wave = (rows_per_wave / 16 + 1) * (RD > 0 ? RD : 0) * 64;
            const unsigned grid = (unsigned)(waves / (threads / 64));
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((stream_kernel<RD>), dim3(grid), dim3(threads), 0, 0, out, in, rows_per_wave, in_per_wave, sink, do_store);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double wr = do_store ? (double)rows_per_wave * waves * 1024 : 0.0;
            const double rd = (double)(rows_per_wave / 16) * waves * RD * 256;
            printf("%-28s wg %3d thr  waves/SIMD %d  %.3f ms  written %.2f GB read %.2f GB  %.0f GB/s\n", name, threads, wps, best, wr / 1e9, rd / 1e9,
                   (wr + rd) / best / 1e6);
        }
    (void)bytes_per_row_read;
}

// Allocation modes (argv[2]): 0 hipMalloc; 1 physically contiguous (hipDeviceMallocContiguous); 2 / 3: virtual range backed by
// 2 MiB (or the device's granularity) physical chunks mapped in shuffled / in creation order (hipMemCreate + hipMemMap) --
// the pileup kernel's time differs by 10 % between processes with where its buffers lie; does a plain stream's?
#include <vector>
#include <chrono>
#include <algorithm>
#include <random>
static size_t g_chunk = 2u << 20;
static void *alloc_mode(size_t bytes, int mode)
{
    void *p = nullptr;
    if (mode == 0) { hipMalloc(&p, bytes); return p; }
    if (mode == 1) { if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous allocation failed\n"); hipMalloc(&p, bytes); } return p; }
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (gran < g_chunk) gran = g_chunk;
    const size_t n = (bytes + gran - 1) / gran;
    hipDeviceptr_t va;
    if (hipMemAddressReserve(&va, n * gran, 0, 0, 0) != hipSuccess) { printf("reserve failed\n"); hipMalloc(&p, bytes); return p; }
    if (mode >= 4) {      // mode 4 + k: create (k + 2) * n chunks, keep every (k + 2)-th, release the rest: a buffer spread over a wide physical span
        const size_t k = (size_t)mode - 2, big_n = n * k;
        std::vector<hipMemGenericAllocationHandle_t> all(big_n);
        for (size_t i = 0; i < big_n; ++i) if (hipMemCreate(&all[i], gran, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", i); exit(1); }
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = i;
        std::mt19937_64 rng(12345); std::shuffle(order.begin(), order.end(), rng);
        for (size_t i = 0; i < n; ++i) hipMemMap((hipDeviceptr_t)((char *)va + i * gran), gran, 0, all[order[i] * k], 0);
        for (size_t i = 0; i < big_n; ++i) if (i % k) hipMemRelease(all[i]);
        hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        hipMemSetAccess(va, n * gran, &acc, 1);
        return (void *)va;
    }
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    hipEvent_t t0, t1; (void)t0; (void)t1;
    auto w0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i) if (hipMemCreate(&h[i], gran, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", i); exit(1); }
    printf("  %zu x hipMemCreate: %.1f ms\n", n, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    if (mode == 2) { std::mt19937_64 rng(12345); std::shuffle(order.begin(), order.end(), rng); }
    for (size_t i = 0; i < n; ++i) hipMemMap((hipDeviceptr_t)((char *)va + i * gran), gran, 0, h[order[i]], 0);
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemSetAccess(va, n * gran, &acc, 1);
    printf("mode %d: %zu chunks of %zu KiB\n", mode, n, gran >> 10);
    return (void *)va;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 7.4;     // (the int32 pass of the bench set writes 7.9 GB)
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    if (argc > 3) g_chunk = (size_t)atoi(argv[3]) << 20;
    int4 *out; int *in; int *sink;
    out = (int4 *)alloc_mode((size_t)(gib * (1LL << 30)) + (1 << 20), mode);
    const size_t in_bytes = (size_t)(gib * (1LL << 30)) / 2 + (64 << 20);
    in = (int *)alloc_mode(in_bytes, mode); hipMemset(in, 1, in_bytes);
    hipMalloc(&sink, 4);
    if (argc > 4) {     // several output buffers, one after the other: is a buffer's rate a property of where it lies?
        const int nbuf = atoi(argv[4]);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int b = 0; b < nbuf; ++b) {
            int4 *o2 = b == 0 ? out : (int4 *)alloc_mode((size_t)(gib * (1LL << 30)) + (1 << 20), mode);
            const long long waves = 4096, total_rows = (long long)(gib * (1LL << 30)) / 1024, rows_per_wave = (total_rows / waves) / 16 * 16;
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((stream_kernel<0>), dim3(4096), dim3(64), 0, 0, o2, in, rows_per_wave, 0LL, sink, 1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("buffer %d at %p: store-only %.3f ms = %.0f GB/s\n", b, (void *)o2, best, (double)rows_per_wave * waves * 1024 / best / 1e6);
        }
        return 0;
    }
    run<0>("store only (1 KiB rows)", out, in, gib, sink, 1, 0);
    run<28>("int32 pass mix (0.44 rd/wr)", out, in, gib, sink, 1, 0);      // 28 x 256 B read per 16 KiB written
    run<16>("load only (256 B rows)", out, in, gib, sink, 0, 0);
    return 0;
}
