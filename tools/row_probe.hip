// row_probe.hip -- what does one row of pileup_fast_kernel's pass B cost, piece by piece?  One workgroup of four waves per
// CU (or more: argv[1] workgroups per CU), every wave walks rows of 256 LDS slots exactly as pass B does; the pieces are
// switched off one at a time.  Prints cycles (s_memtime) per row.   hipcc -O3 --offload-arch=gfx950 tools/row_probe.hip -o tools/row_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../raft_amd/csrc/wave.hpp"
using namespace raft;

template <int MODE>   // bit 0: global store, 1: zero the row, 2: ballots + branch, 3: DPP scan + carry, 4: LDS read of the next row
__global__ __launch_bounds__(256, 4) void row_kernel(int32_t *cov, int rows_per_wave, int iters, int high, unsigned long long *out, int *sink)
{
    __shared__ __attribute__((aligned(16))) int32_t diff[8192 + 256];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192 + 256; i += 256) diff[i] = (i * 7 + 3) & 1;
    __syncthreads();
    int carry = 0, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int32_t *const cov0 = cov + ((long long)blockIdx.x * iters + it) * 8192;
        const int row_b = wid * rows_per_wave, row_e = row_b + rows_per_wave;
        int4 dn = *reinterpret_cast<const int4 *>(&diff[row_b * 256 + lane * 4]);
        for (int row = row_b; row < row_e; ++row) {
            const int p0 = row * 256 + lane * 4;
            const int4 d = dn;
            if (MODE & 16) dn = *reinterpret_cast<const int4 *>(&diff[p0 + 256]);
            if (MODE & 2) *reinterpret_cast<int4 *>(&diff[p0]) = make_int4(0, 1, 0, 1);
            const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
            int incl = w;
            if (MODE & 8) { incl = wave_incl_scan_add(w); carry += __builtin_amdgcn_readlane(incl, 63); }
            const int excl = incl - w + carry;
            const int c0 = excl + x, c1 = excl + y, c2 = excl + z, c3 = excl + w;
            if (MODE & 1) *reinterpret_cast<int4 *>(reinterpret_cast<char *>(cov0) + (unsigned)p0 * 4u) = make_int4(c0, c1, c2, c3);
            else acc += c0 ^ c1 ^ c2 ^ c3;
            if (MODE & 4) {
                const unsigned long long M0 = __ballot(c0 >= high), M1 = __ballot(c1 >= high), M2 = __ballot(c2 >= high), M3 = __ballot(c3 >= high);
                if ((M0 | M1 | M2 | M3) != 0ull) acc += (int)__popcll(M0 ^ M3);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x7fffffff) *sink = carry;
}

// the same rows two at a time: the two prefix scans are independent chains that the wave can interleave
template <int MODE>
__global__ __launch_bounds__(256, 4) void row2_kernel(int32_t *cov, int rows_per_wave, int iters, int high, unsigned long long *out, int *sink)
{
    __shared__ __attribute__((aligned(16))) int32_t diff[8192 + 512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192 + 512; i += 256) diff[i] = (i * 7 + 3) & 1;
    __syncthreads();
    int carry = 0, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        int32_t *const cov0 = cov + ((long long)blockIdx.x * iters + it) * 8192;
        const int row_b = wid * rows_per_wave, row_e = row_b + rows_per_wave;
        int4 dn = *reinterpret_cast<const int4 *>(&diff[row_b * 256 + lane * 4]);
        int4 dm = *reinterpret_cast<const int4 *>(&diff[row_b * 256 + 256 + lane * 4]);
        for (int row = row_b; row < row_e; row += 2) {
            const int p0 = row * 256 + lane * 4;
            const int4 d = dn, e = dm;
            dn = *reinterpret_cast<const int4 *>(&diff[p0 + 512]);
            dm = *reinterpret_cast<const int4 *>(&diff[p0 + 768]);
            if (MODE & 2) { *reinterpret_cast<int4 *>(&diff[p0]) = make_int4(0, 1, 0, 1); *reinterpret_cast<int4 *>(&diff[p0 + 256]) = make_int4(0, 1, 0, 1); }
            const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
            const int x2 = e.x, y2 = x2 + e.y, z2 = y2 + e.z, w2 = z2 + e.w;
            // two scans, interleaved by hand (the compiler keeps each builtin chain together otherwise)
            int a1 = w, a2 = w2;
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x111, 0xf, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x111, 0xf, 0xf, false);
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x112, 0xf, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x112, 0xf, 0xf, false);
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x114, 0xf, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x114, 0xf, 0xf, false);
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x118, 0xf, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x118, 0xf, 0xf, false);
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x142, 0xa, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x142, 0xa, 0xf, false);
            a1 += __builtin_amdgcn_update_dpp(0, a1, 0x143, 0xc, 0xf, false); a2 += __builtin_amdgcn_update_dpp(0, a2, 0x143, 0xc, 0xf, false);
            const int excl = a1 - w + carry;
            carry += __builtin_amdgcn_readlane(a1, 63);
            const int excl2 = a2 - w2 + carry;
            carry += __builtin_amdgcn_readlane(a2, 63);
            const int c0 = excl + x, c1 = excl + y, c2 = excl + z, c3 = excl + w;
            const int f0 = excl2 + x2, f1 = excl2 + y2, f2 = excl2 + z2, f3 = excl2 + w2;
            if (MODE & 1) {
                *reinterpret_cast<int4 *>(reinterpret_cast<char *>(cov0) + (unsigned)p0 * 4u) = make_int4(c0, c1, c2, c3);
                *reinterpret_cast<int4 *>(reinterpret_cast<char *>(cov0) + (unsigned)(p0 + 256) * 4u) = make_int4(f0, f1, f2, f3);
            } else acc += c0 ^ c1 ^ c2 ^ c3 ^ f0 ^ f1 ^ f2 ^ f3;
            if (MODE & 4) {
                const unsigned long long M0 = __ballot(c0 >= high), M1 = __ballot(c1 >= high), M2 = __ballot(c2 >= high), M3 = __ballot(c3 >= high);
                const unsigned long long N0 = __ballot(f0 >= high), N1 = __ballot(f1 >= high), N2 = __ballot(f2 >= high), N3 = __ballot(f3 >= high);
                if ((M0 | M1 | M2 | M3 | N0 | N1 | N2 | N3) != 0ull) acc += (int)__popcll(M0 ^ N3);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 0x7fffffff) *sink = carry;
}

template <int MODE, bool TWO = false>
void run(const char *name, int wg_per_cu, int32_t *cov, unsigned long long *out, int *sink)
{
    const int grid = 256 * wg_per_cu, rows = 8, iters = 200;
    auto launch = [&]() {
        if (TWO) hipLaunchKernelGGL(row2_kernel<MODE>, dim3(grid), dim3(256), 0, 0, cov, rows, iters, 1 << 30, out, sink);
        else hipLaunchKernelGGL(row_kernel<MODE>, dim3(grid), dim3(256), 0, 0, cov, rows, iters, 1 << 30, out, sink);
    };
    launch();
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per_row = s / grid / (double)(rows * iters);
    printf("%-46s wg/cu %d: %7.1f cycles per row and wave, kernel %.3f ms, %.2f TB/s of stores\n", name, wg_per_cu, per_row, ms,
           (MODE & 1) ? (double)grid * iters * 8192 * 4 / (ms * 1e-3) / 1e12 : 0.0);
}

int main(int argc, char **argv)
{
    const int wpc = argc > 1 ? atoi(argv[1]) : 1;
    int32_t *cov; unsigned long long *out; int *sink;
    hipMalloc(&cov, (size_t)256 * 4 * 200 * 8192 * 4 + (1 << 20)); hipMalloc(&out, 8192 * 8); hipMalloc(&sink, 4);
    run<31>("everything (pass B's common path)", wpc, cov, out, sink);
    run<30>("no global store", wpc, cov, out, sink);
    run<29>("no zeroing of the row", wpc, cov, out, sink);
    run<27>("no ballots / branch", wpc, cov, out, sink);
    run<23>("no DPP scan / carry", wpc, cov, out, sink);
    run<15>("no LDS read of the next row (one row re-used)", wpc, cov, out, sink);
    run<31, true>("everything, two rows at a time", wpc, cov, out, sink);
    run<30, true>("no global store, two rows at a time", wpc, cov, out, sink);
    run<16>("LDS read only", wpc, cov, out, sink);
    run<1>("store only", wpc, cov, out, sink);
    return 0;
}
