// tools/membench.hip -- what the chip sustains for the pileup kernel's memory shape (diagnostic, not product code):
//   fill:   persistent 256-thread workgroups, each wave stores rows of 1 KiB (int4 per lane), tile after tile
//   mixed:  same stores + per tile a coalesced read of 30 % as many bytes (the interval columns)
// usage: membench <GiB of stores> ; prints GB/s for several workgroups-per-CU counts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void fill_kernel(int4 *out, long long n_tiles, int rows_per_wave, int mode,
                                                   const int4 *in, long long in_vec, int *sink)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int acc = 0;
    for (long long k = blockIdx.x; k < n_tiles; k += gridDim.x) {
        const long long base = (k * 4 + wid) * rows_per_wave * 64;
        if (mode == 1) { // read 0.3x the bytes first (3 x int4 per 10 stored), like the interval columns
            const long long ib = (k * 4 + wid) * (long long)((rows_per_wave * 3 + 9) / 10) * 64;
            for (int r = 0; r < (rows_per_wave * 3 + 9) / 10; ++r) {
                const int4 v = in[(ib + r * 64 + lane) % in_vec];
                acc += v.x + v.y + v.z + v.w;
            }
        }
        for (int r = 0; r < rows_per_wave; ++r) out[base + r * 64 + lane] = make_int4(acc, r, lane, wid);
    }
    if (acc == 0x7fffffff) *sink = acc;
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const int rows_per_wave = 5;
    const long long tile_bytes = 4LL * rows_per_wave * 1024;
    const long long n_tiles = (long long)(gib * (1LL << 30)) / tile_bytes;
    int4 *out, *in; int *sink;
    hipMalloc(&out, n_tiles * tile_bytes);
    const long long in_vec = (3LL << 30) / 16;
    hipMalloc(&in, in_vec * 16); hipMemset(in, 1, in_vec * 16);
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int bpc : {1, 2, 4, 5, 8}) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(fill_kernel, dim3(256 * bpc), dim3(256), 0, 0, out, n_tiles, rows_per_wave, mode, in, in_vec, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)n_tiles * tile_bytes * (mode ? 1.3 : 1.0);
            printf("mode %s  wg/CU %d  %.3f ms  %.0f GB/s\n", mode ? "store+0.3read" : "store", bpc, best, bytes / best / 1e6);
        }
    return 0;
}
