// pack.hpp -- transfer encoding of the coverage array, and the small kernels of the chunked host pipeline.
//
// cov[] (int32 per window, repeat.hpp:105-108's "pos,cov" pairs without the implied pos) is the one large output of
// the path: 8 GB at human scale, two thirds of the bytes that cross PCIe in an end-to-end pass.  The reference's
// consumer is a text formatter, and real coverage rarely needs more than a byte, so the host side takes
//     cov8[i] = min(cov[i], 255)      plus a list of (window index, value) for the windows with cov >= 255
// -- a quarter of the bytes, decoded exactly (raft_host_unpack_coverage / raft_host_write_coverage_packed).
// The engine's own output stays the int32 array; this is a copy made on request (raft_hip_fetch_packed).
#pragma once
#include "wave.hpp"

namespace raft {

struct PackOut {
    uint8_t *cov8;
    unsigned long long *n_exc;   // windows with cov >= 255 (counted even when they no longer fit the list)
    long long exc_cap;
    long long *exc_idx;
    int32_t *exc_val;
};

__device__ __forceinline__ unsigned pack4(const int4 v, long long i, const PackOut &o)
{
    // coverage is never negative: every read's +1/-1 balance out
    const unsigned a = min((unsigned)v.x, 255u), b = min((unsigned)v.y, 255u), c = min((unsigned)v.z, 255u), d = min((unsigned)v.w, 255u);
    if (((unsigned)v.x | (unsigned)v.y | (unsigned)v.z | (unsigned)v.w) >= 255u) {   // cheap pre-test: any value >= 255 sets a bit >= 2^8 or is 255
        const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (vv[k] >= 255) {
                const unsigned long long slot = atomicAdd(o.n_exc, 1ull);
                if ((long long)slot < o.exc_cap) { o.exc_idx[slot] = i + k; o.exc_val[slot] = vv[k]; }
            }
    }
    return a | (b << 8) | (c << 16) | (d << 24);
}

// 16 B in, 4 B out per lane and step: 1 KiB wave loads, 256 B wave stores; the pass is bound by the 4 B/window it reads.
__global__ __launch_bounds__(256) void pack_cov_kernel(const int32_t *__restrict__ cov, long long n_bins, PackOut o)
{
    const long long n4 = n_bins >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int4 *src = reinterpret_cast<const int4 *>(cov);
    unsigned *dst = reinterpret_cast<unsigned *>(o.cov8);
    for (; g + 3 * stride < n4; g += 4 * stride) {       // four independent loads in flight per lane
        const int4 v0 = src[g], v1 = src[g + stride], v2 = src[g + 2 * stride], v3 = src[g + 3 * stride];
        dst[g] = pack4(v0, g << 2, o); dst[g + stride] = pack4(v1, (g + stride) << 2, o);
        dst[g + 2 * stride] = pack4(v2, (g + 2 * stride) << 2, o); dst[g + 3 * stride] = pack4(v3, (g + 3 * stride) << 2, o);
    }
    for (; g < n4; g += stride) dst[g] = pack4(src[g], g << 2, o);
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n_bins & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const int v = cov[i];
        o.cov8[i] = (uint8_t)min((unsigned)v, 255u);
        if (v >= 255) {
            const unsigned long long slot = atomicAdd(o.n_exc, 1ull);
            if ((long long)slot < o.exc_cap) { o.exc_idx[slot] = i; o.exc_val[slot] = v; }
        }
    }
}

// chunked pipeline: read ids of a chunk's records are rebased to the chunk's first read
__global__ __launch_bounds__(256) void rebase_ids_kernel(int32_t *ids, long long n, int32_t base)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ids[i] -= base;
}

// chunked pipeline: CSR offsets of a chunk become offsets into the whole job's arrays
__global__ __launch_bounds__(256) void add_base_kernel(long long *a, long long n, long long base)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] += base;
}

} // namespace raft
