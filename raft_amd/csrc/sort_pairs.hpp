// sort_pairs.hpp -- a small LSD radix sort of (64-bit key, 32-bit value) pairs, hand-written for the one list the engine has to
// hand out in order: the windows a packed coverage encoding lists beside its codes (pack.hpp: "exceptions"; 3.7e6 pairs at human
// scale with four-bit steps, none with a byte per window on a 32x set).  Off the hot path: three small launches per 8-bit digit
// of the key bits in use, stable, any n.  (Replaces nothing in the reference: repeat.hpp:102-108 walks cov[] in order.)
//   hist:    every workgroup (one wave) counts the digit values of its tile of kSortTile pairs -> hist[digit][tile]
//   scan:    one workgroup, exclusive scan over the 256 x tiles counts in digit-major order -> where every (digit, tile) goes
//   scatter: the wave walks its tile 64 pairs at a time; lanes with the same digit find each other with eight ballots (one per
//            bit), the first of them draws the group's place from the tile's running offsets in LDS: stable.
#pragma once
#include "wave.hpp"

namespace raft {

constexpr int kSortTile = 2048;

__global__ __launch_bounds__(64) void sort_hist_kernel(const unsigned long long *__restrict__ key, long long n, int shift, int n_tiles, int32_t *__restrict__ hist)
{
    __shared__ int32_t h[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) h[i] = 0;
    __syncthreads();
    const long long t0 = (long long)blockIdx.x * kSortTile;
    for (int i = lane; i < kSortTile; i += 64)
        if (t0 + i < n) atomicAdd(&h[(int)((key[t0 + i] >> shift) & 255ull)], 1);
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(long long)d * n_tiles + blockIdx.x] = h[d];
}

__global__ __launch_bounds__(1024) void sort_scan_kernel(long long m, int32_t *__restrict__ hist)      // exclusive, in place; totals fit 31 bits (n < 2^31)
{
    __shared__ long long part[1024];
    const long long per = (m + 1023) / 1024;
    const long long lo = std::min<long long>((long long)threadIdx.x * per, m), hi = std::min<long long>(lo + per, m);
    long long s = 0;
    for (long long i = lo; i < hi; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) { const int32_t v = hist[i]; hist[i] = (int32_t)run; run += v; }
}

__global__ __launch_bounds__(64) void sort_scatter_kernel(const unsigned long long *__restrict__ key_in, const int32_t *__restrict__ val_in, long long n, int shift,
                                                          int n_tiles, const int32_t *__restrict__ hist, unsigned long long *__restrict__ key_out,
                                                          int32_t *__restrict__ val_out)
{
    __shared__ int32_t next[256];             // where the tile's next pair with this digit goes
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) next[d] = hist[(long long)d * n_tiles + blockIdx.x];
    __syncthreads();
    const long long t0 = (long long)blockIdx.x * kSortTile;
    for (int b = 0; b < kSortTile; b += 64) {
        const long long i = t0 + b + lane;
        const bool live = i < n;
        if (__ballot(live) == 0ull) break;
        const unsigned long long k = live ? key_in[i] : 0ull;
        const int v = live ? val_in[i] : 0;
        const int d = (int)((k >> shift) & 255ull);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((d >> bit) & 1);
            peers &= ((d >> bit) & 1) ? m : ~m;
        }
        // (peers: the live lanes with this lane's digit; the lowest of them draws for all)
        const int rank = (int)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = (int)__builtin_ctzll(peers | (1ull << 63));
        int first = 0;
        if (live && lane == leader) { first = next[d]; next[d] = first + (int)__popcll(peers); }
        first = __shfl(first, leader, kWave);
        if (live) { key_out[first + rank] = k; val_out[first + rank] = v; }
        __syncthreads();                      // (one wave: orders the LDS updates of this batch before the next batch's reads)
    }
}

// n pairs, ascending by the low `bits` bits of the key.  The result is in (key_a, val_a) when the number of digits is even,
// else in (key_b, val_b): *in_b says which.  hist: 256 * ceil(n / kSortTile) int32.
inline hipError_t sort_pairs(hipStream_t st, unsigned long long *key_a, int32_t *val_a, unsigned long long *key_b, int32_t *val_b, long long n, int bits,
                             int32_t *hist, bool *in_b)
{
    *in_b = false;
    if (n < 2) return hipSuccess;
    const int n_tiles = (int)((n + kSortTile - 1) / kSortTile);
    unsigned long long *ki = key_a, *ko = key_b;
    int32_t *vi = val_a, *vo = val_b;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, ki, n, shift, n_tiles, hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, st, (long long)256 * n_tiles, hist);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)n_tiles), dim3(64), 0, st, ki, vi, n, shift, n_tiles, hist, ko, vo);
        std::swap(ki, ko); std::swap(vi, vo);
        *in_b = !*in_b;
    }
    return hipGetLastError();
}
inline size_t sort_pairs_hist_bytes(long long n) { return (size_t)256 * (size_t)((n + kSortTile - 1) / kSortTile + 1) * 4; }

} // namespace raft
