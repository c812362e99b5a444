// device_scan.hpp -- device-wide exclusive prefix sums over per-read quantities.
//
// These run over N (reads), not over bins or records, so they are a few MB of
// traffic against the GBs of the pileup kernel; a plain three-launch
// reduce / scan-partials / apply scheme is enough.  K independent int64 sums are
// carried at once so that one pass produces several CSR offset arrays.
// (Measured and dropped, round 3: ONE launch with a decoupled look-back -- rocPRIM's device scan over a loader iterator and
// an iterator that writes the K arrays.  7 us faster per pass on 50 k reads, 0.3 ms SLOWER on 3.3 M: with 1600 workgroups
// resident at once the look-backs walk over each other, as in the fused finalize chain of round 2.)
#pragma once
#include "wave.hpp"

namespace raft {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

template <int K> struct ScanOut { long long *p[K]; };

inline int scan_blocks(long long n) { return (int)((n + kScanTile - 1) / kScanTile); }

// block-wide exclusive scan of one int64 per thread; returns exclusive prefix, total in *total
template <int THREADS>
__device__ __forceinline__ long long block_excl_scan64(long long v, long long *total, long long *lds /*[THREADS/64 + 1]*/)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    long long incl = wave_incl_scan_add64(v);
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    long long base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        long long s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// Hooks (round 6: fewer launches at the head of a pass).  `Beside`: workgroups behind the scan's own run something else in the
// same launch (the run guess, bucket.hpp GuessBeside).  `Post`: the apply kernel hands every element's exclusive offsets and values
// to a functor as it writes them (the per-read work of tile_first_kernel, bucket.hpp PrepPost), and the grand totals to its
// closing() on the last workgroup.
struct NoBeside { __device__ void operator()(int) const {} };
struct NoPost {
    template <int K> __device__ void operator()(long long, const long long (&)[K], const long long (&)[K]) const {}
    template <int K> __device__ void closing(const long long (&)[K]) const {}
};

template <class Loader, int K, class Beside = NoBeside>
__global__ __launch_bounds__(kScanThreads) void scan_partials_kernel(Loader ld, long long n, long long *partials, int n_scan_blocks = 0x7fffffff,
                                                                     Beside beside = Beside())
{
    if ((int)blockIdx.x >= n_scan_blocks) { beside((int)blockIdx.x - n_scan_blocks); return; }
    __shared__ long long lds[kScanThreads / 64 + 1];
    long long acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0;
    // only the block total matters here, so the elements are taken striped: coalesced loads
    const long long block0 = (long long)blockIdx.x * kScanTile;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const long long i = block0 + j * kScanThreads + threadIdx.x;
        if (i < n) {
            long long v[K];
            ld(i, v);
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long long tot;
        (void)block_excl_scan64<kScanThreads>(acc[k], &tot, lds);
        if (threadIdx.x == 0) partials[(long long)blockIdx.x * K + k] = tot;
    }
}

// FUSED: `partials` holds the workgroups' raw totals (scan_partials_kernel) and every workgroup adds up the ones before it by
// itself -- a few KB from L2 per workgroup -- instead of waiting for a one-workgroup kernel to scan them (12 us per scan at
// human scale, twice per pass); the last workgroup writes the grand totals.
template <class Loader, int K, bool FUSED = true, class Post = NoPost>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(Loader ld, long long n, const long long *partials,
                                                                  long long *totals, ScanOut<K> out, Post post = Post())
{
    __shared__ long long lds[kScanThreads / 64 + 1];
    __shared__ long long stage[kScanTile];      // blocked -> striped, so that the stores are coalesced
    long long base[K];
    // (the workgroup's own elements are asked for first: they arrive while the totals of the workgroups before are added up)
    long long v[kScanItems][K];
    const long long block0 = (long long)blockIdx.x * kScanTile;
    const long long i0 = block0 + (long long)threadIdx.x * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        const long long i = i0 + j;
        if (i < n) ld(i, v[j]);
        else {
#pragma unroll
            for (int k = 0; k < K; ++k) v[j][k] = 0;
        }
    }
    if (FUSED) {
        long long mine[K];
#pragma unroll
        for (int k = 0; k < K; ++k) mine[k] = 0;
        for (int b = (int)threadIdx.x; b < (int)blockIdx.x; b += kScanThreads) {
#pragma unroll
            for (int k = 0; k < K; ++k) mine[k] += partials[(long long)b * K + k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            long long tot;
            (void)block_excl_scan64<kScanThreads>(mine[k], &tot, lds);
            base[k] = tot;
        }
    }
    long long acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += v[j][k];
    }
    long long first[K], grand[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long long tot;
        long long run = (FUSED ? base[k] : partials[(long long)blockIdx.x * K + k]) + block_excl_scan64<kScanThreads>(acc[k], &tot, lds);
        first[k] = run; grand[k] = FUSED ? base[k] + tot : 0;
        if (FUSED && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { totals[k] = base[k] + tot; if (out.p[k]) out.p[k][n] = base[k] + tot; }
        if (!out.p[k]) continue;                     // (a sum whose total is all anybody wants: no array written)
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            stage[threadIdx.x * kScanItems + j] = run;
            run += v[j][k];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kScanItems; ++j) {
            const long long i = block0 + j * kScanThreads + threadIdx.x;
            if (i < n) out.p[k][i] = stage[j * kScanThreads + threadIdx.x];
        }
        __syncthreads();
    }
    if (!FUSED && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) if (out.p[k]) out.p[k][n] = totals[k];
    }
    // the elements once more, for the hook: where each begins (all K sums) and what it holds
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (i0 + j < n) post(i0 + j, first, v[j]);
#pragma unroll
        for (int k = 0; k < K; ++k) first[k] += v[j][k];
    }
    if (FUSED && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) post.closing(grand);
}

// Host driver.  partials must hold (scan_blocks(n) * K) + K int64; totals = partials + nblocks*K.
template <class Loader, int K>
inline void exclusive_scan(hipStream_t st, Loader ld, long long n, long long *partials, ScanOut<K> out,
                           long long **totals_dev = nullptr)
{
    int nb = scan_blocks(n);
    if (nb < 1) nb = 1;
    long long *totals = partials + (long long)nb * K;
    hipLaunchKernelGGL((scan_partials_kernel<Loader, K>), dim3(nb), dim3(kScanThreads), 0, st, ld, n, partials);
    hipLaunchKernelGGL((scan_apply_kernel<Loader, K, true>), dim3(nb), dim3(kScanThreads), 0, st, ld, n, partials, totals, out);
    if (totals_dev) *totals_dev = totals;
}

} // namespace raft
