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

// ---- the big one: sides of a record stream by read id (bucket.hpp: general bucketing; round 5, replaces rocprim::radix_sort_pairs) -------
// LSD radix sort of (32-bit key, 32- or 64-bit value) pairs, 8 bits per pass, every global store part of a RUN: a workgroup takes a
// tile, ranks its pairs by digit (stable: waves own consecutive quarters of the tile, a wave walks its quarter 64 pairs at a time
// and finds the lanes with its digit by eight ballots), places them in LDS in digit order and writes the tile out from there --
// consecutive lanes, consecutive addresses, ~32 pairs per digit and tile.  (What bucket.hpp's dropped two-step partition did not do:
// a wave's 64 lanes storing to 64 different lines retire at ~34 ps per lane on this part.)
//   rs_hist                          digit counts of every tile -> hist[tile][digit]                  (4 B read per pair)
//   rs_segsum / segscan / tilescan   the table's exclusive scan, digit-major: where every (tile, digit) goes (coalesced, < 1 % of the bytes)
//   rs_scatter                       rank, stage in LDS, write runs                                     (8-12 B read, 8-12 B written per pair)
#ifndef RAFT_RS_WAVES
#define RAFT_RS_WAVES 4            // waves per workgroup of the sort's tile kernels
#endif
constexpr int kRsWaves = RAFT_RS_WAVES, kRsThreads = 64 * kRsWaves;
// Workgroups go to the eight XCDs round-robin; tile = (b % 8) * ceil(n / 8) + b / 8 hands every XCD a CONTIGUOUS eighth of the tiles, so
// that the runs two neighbouring tiles write for one digit -- which are neighbours in memory -- meet in one L2 and leave it as whole lines.
__device__ __forceinline__ int rs_tile_of_block(int b, int n_tiles) { const int per = (n_tiles + 7) >> 3; return (b & 7) * per + (b >> 3); }
inline unsigned rs_grid(int n_tiles) { return (unsigned)(((n_tiles + 7) >> 3) * 8); }
template <int IPT> struct RsGeom { static constexpr int kTile = kRsThreads * IPT, kQuarter = 64 * IPT; };

template <int IPT>
__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t *__restrict__ key, long long n, int shift, int n_tiles, int32_t *__restrict__ hist)
{
    __shared__ int32_t h[256];
    const int tile = rs_tile_of_block((int)blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const long long t0 = (long long)tile * RsGeom<IPT>::kTile;
#pragma unroll 8
    for (int b = 0; b < IPT; ++b) {
        const long long i = t0 + (long long)b * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(long long)tile * 256 + threadIdx.x] = h[threadIdx.x];      // [tile][digit]: every access below is 256 consecutive words
}

// The table's exclusive scan in digit-major order (all tiles' digit 0, then all tiles' digit 1, ...) in three coalesced steps over
// kRsSegs segments of consecutive tiles; thread d owns digit d everywhere.
constexpr int kRsSegs = 256;
__global__ __launch_bounds__(256) void rs_segsum_kernel(int n_tiles, const int32_t *__restrict__ hist, int32_t *__restrict__ segtot)
{
    const int L = (n_tiles + kRsSegs - 1) / kRsSegs;
    const int t0 = blockIdx.x * L, t1 = min(t0 + L, n_tiles);
    int s = 0;
#pragma unroll 8
    for (int t = t0; t < t1; ++t) s += hist[(long long)t * 256 + threadIdx.x];
    segtot[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void rs_segscan_kernel(int32_t *__restrict__ segtot)      // in place: where every (segment, digit) begins
{
    __shared__ int32_t wsum[4];
    int run = 0;
    for (int s = 0; s < kRsSegs; ++s) { const int v = segtot[s * 256 + threadIdx.x]; segtot[s * 256 + threadIdx.x] = run; run += v; }
    const int incl = wave_incl_scan_add(run);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = incl - run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
    for (int s = 0; s < kRsSegs; ++s) segtot[s * 256 + threadIdx.x] += base;
}
__global__ __launch_bounds__(256) void rs_tilescan_kernel(int n_tiles, int32_t *__restrict__ hist, const int32_t *__restrict__ segbase)
{
    const int L = (n_tiles + kRsSegs - 1) / kRsSegs;
    const int t0 = blockIdx.x * L, t1 = min(t0 + L, n_tiles);
    int run = segbase[blockIdx.x * 256 + threadIdx.x];
    for (int t = t0; t < t1; t += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = t + k < t1 ? hist[(long long)(t + k) * 256 + threadIdx.x] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (t + k < t1) { hist[(long long)(t + k) * 256 + threadIdx.x] = run; run += v[k]; }
    }
}

template <class V, int IPT>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint32_t *__restrict__ key_in, const V *__restrict__ val_in, long long n, int shift, int n_tiles,
                                                         const int32_t *__restrict__ tile_off, uint32_t *__restrict__ key_out, V *__restrict__ val_out)
{
    constexpr int TILE = RsGeom<IPT>::kTile, QUARTER = RsGeom<IPT>::kQuarter;
    __shared__ int32_t cnt[kRsWaves][256];     // per wave: pairs of its share with the digit; then: where its next pair with the digit goes (tile-local)
    __shared__ int32_t goff[256];              // global place of the tile's first pair with the digit, minus its tile-local place
    __shared__ int32_t wsum[4];
    __shared__ uint32_t s_key[TILE];
    __shared__ V s_val[TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = rs_tile_of_block((int)blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const long long t0 = (long long)tile * TILE;
    const int n_valid = (int)min((long long)TILE, n - t0);
    for (int i = tid; i < kRsWaves * 256; i += kRsThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();
    // 1. the keys of this thread's pairs (wave w: quarter w, 64 consecutive pairs per step) and the waves' digit counts
    uint32_t k[IPT];
    V v[IPT];                                  // (the values too: a load per step inside the ranking loop below is a latency per step)
#pragma unroll
    for (int b = 0; b < IPT; ++b) {
        const int j = w * QUARTER + b * 64 + lane;
        k[b] = j < n_valid ? key_in[t0 + j] : 0xffffffffu;
        v[b] = j < n_valid ? val_in[t0 + j] : V(0);
    }
#pragma unroll
    for (int b = 0; b < IPT; ++b)
        if (w * QUARTER + b * 64 + lane < n_valid) atomicAdd(&cnt[w][(k[b] >> shift) & 255u], 1);
    __syncthreads();
    // 2. thread d: the tile's pairs with digit d begin at lstart (exclusive scan over the digits); wave w's at lstart + the waves before
    {
        int c[kRsWaves], tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < kRsWaves; ++q) { c[q] = cnt[q][tid]; tot += c[q]; }
        }
        const int incl = wave_incl_scan_add(tot);                 // (threads 256 and up carry zeros: every wave takes part in the scan)
        if (tid < 256 && lane == 63) wsum[w] = incl;
        __syncthreads();
        if (tid < 256) {
            int lstart = incl - tot;
            for (int q = 0; q < w; ++q) lstart += wsum[q];
            int run = lstart;
#pragma unroll
            for (int q = 0; q < kRsWaves; ++q) { cnt[q][tid] = run; run += c[q]; }
            goff[tid] = tile_off[(long long)tile * 256 + tid] - lstart;
        }
    }
    __syncthreads();
    // 3. every wave walks its quarter in order: lanes with the same digit find each other, the lowest draws the group's place
#pragma unroll
    for (int b = 0; b < IPT; ++b) {
        const int j = w * QUARTER + b * 64 + lane;
        const bool live = j < n_valid;
        if (__ballot(live) == 0ull) break;
        const int d = (int)((k[b] >> shift) & 255u);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((d >> bit) & 1);
            peers &= ((d >> bit) & 1) ? m : ~m;
        }
        const int rank = (int)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = (int)__builtin_ctzll(peers | (1ull << 63));
        int first = 0;
        if (live && lane == leader) { first = cnt[w][d]; cnt[w][d] = first + (int)__popcll(peers); }
        first = __shfl(first, leader, kWave);
        if (live) { s_key[first + rank] = k[b]; s_val[first + rank] = v[b]; }
    }
    __syncthreads();
    // 4. the tile in digit order, written as runs
    for (int j = tid; j < n_valid; j += kRsThreads) {
        const uint32_t kk = s_key[j];
        const int dst = goff[(kk >> shift) & 255u] + j;
        key_out[dst] = kk; val_out[dst] = s_val[j];
    }
}

// n pairs by the low `bits` bits of the key.  Buffers a and b ping-pong; *in_b says where the result is.  tmp: rs_tmp_bytes(n) bytes.
template <class V> struct RsIpt { static constexpr int v = sizeof(V) == 8 ? 16 : 32; };      // 4096 pairs of 12 bytes / 8192 of 8: 48 / 64 KB of LDS per tile
template <class V>
inline size_t rs_tmp_bytes(long long n) { return ((size_t)256 * (size_t)((n + RsGeom<RsIpt<V>::v>::kTile - 1) / RsGeom<RsIpt<V>::v>::kTile + 1) + (size_t)256 * kRsSegs) * 4; }
template <class V>
inline hipError_t radix_sort_by_key(hipStream_t st, uint32_t *key_a, V *val_a, uint32_t *key_b, V *val_b, long long n, int bits, void *tmp, bool *in_b)
{
    constexpr int IPT = RsIpt<V>::v;
    *in_b = false;
    if (n < 2) return hipSuccess;
    const int n_tiles = (int)((n + RsGeom<IPT>::kTile - 1) / RsGeom<IPT>::kTile);
    int32_t *hist = static_cast<int32_t *>(tmp), *seg = hist + (size_t)256 * n_tiles;
    uint32_t *ki = key_a, *ko = key_b;
    V *vi = val_a, *vo = val_b;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL((rs_hist_kernel<IPT>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, ki, n, shift, n_tiles, hist);
        hipLaunchKernelGGL(rs_segsum_kernel, dim3(kRsSegs), dim3(256), 0, st, n_tiles, hist, seg);
        hipLaunchKernelGGL(rs_segscan_kernel, dim3(1), dim3(256), 0, st, seg);
        hipLaunchKernelGGL(rs_tilescan_kernel, dim3(kRsSegs), dim3(256), 0, st, n_tiles, hist, seg);
        hipLaunchKernelGGL((rs_scatter_kernel<V, IPT>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, ki, vi, n, shift, n_tiles, hist, ko, vo);
        std::swap(ki, ko); std::swap(vi, vo);
        *in_b = !*in_b;
    }
    return hipGetLastError();
}

// ---- the same sort over 64-bit ITEMS whose low 32 bits are the key (the general bucketing's window-record route: an interval is
// (read id, first window | one past the last << 16) -- 8 bytes where the coordinate pair takes 12, one load and one store per pair and
// pass instead of two, runs of twice the bytes)
// Where a pass of the item sort takes its items from: the array the pass before wrote -- or, for the FIRST pass, a functor that makes
// them on the fly (bucket.hpp SideSource: the sides of the record columns; round 6 -- until then a kernel of its own wrote the items
// out and the first pass read them back, 16 bytes of traffic per side for nothing).  key(i): the low 32 bits alone (the histogram
// looks at nothing else); load(i) + make<REPORT>(raw, i): the whole item, REPORT: errors of the input are raised here, once.
// (load / make: the scatter asks for a batch of raw values first and makes the items afterwards -- made one at a time, a tile's 32
// rounds of loads were 32 round trips: 4.0 ms for the pass where the items' own array takes 1.3)
struct ItemArray {
    const unsigned long long *p;
    typedef unsigned long long Raw;
    __device__ __forceinline__ uint32_t key(long long i) const { return (uint32_t)p[i]; }
    __device__ __forceinline__ Raw load(long long i) const { return p[i]; }
    template <bool REPORT> __device__ __forceinline__ unsigned long long make(const Raw &r, long long) const { return r; }
};

template <int IPT, class Src>
__global__ __launch_bounds__(kRsThreads) void rs_hist_items_kernel(Src src, long long n, int shift, int n_tiles, int32_t *__restrict__ hist)
{
    __shared__ int32_t h[256];
    const int tile = rs_tile_of_block((int)blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const long long t0 = (long long)tile * RsGeom<IPT>::kTile;
#pragma unroll 8
    for (int b = 0; b < IPT; ++b) {
        const long long i = t0 + (long long)b * kRsThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(src.key(i) >> shift) & 255u], 1);
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(long long)tile * 256 + threadIdx.x] = h[threadIdx.x];
}

template <int IPT, class Src, bool REPORT>
__global__ __launch_bounds__(kRsThreads) void rs_scatter_items_kernel(Src src, long long n, int shift, int n_tiles,
                                                               const int32_t *__restrict__ tile_off, unsigned long long *__restrict__ out)
{
    constexpr int TILE = RsGeom<IPT>::kTile, SHARE = RsGeom<IPT>::kQuarter;
    __shared__ int32_t cnt[kRsWaves][256];
    __shared__ int32_t goff[256];
    __shared__ int32_t wsum[4];
    __shared__ unsigned long long s_item[TILE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = rs_tile_of_block((int)blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const long long t0 = (long long)tile * TILE;
    const int n_valid = (int)min((long long)TILE, n - t0);
    for (int i = tid; i < kRsWaves * 256; i += kRsThreads) (&cnt[0][0])[i] = 0;
    __syncthreads();
    unsigned long long it[IPT];
    constexpr int G = 8;                          // raw values asked for at once
    static_assert(IPT % G == 0, "batches of G");
#pragma unroll
    for (int b0 = 0; b0 < IPT; b0 += G) {
        typename Src::Raw raw[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = w * SHARE + (b0 + g) * 64 + lane;
            raw[g] = src.load(t0 + min(j, max(n_valid - 1, 0)));
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int j = w * SHARE + (b0 + g) * 64 + lane;
            it[b0 + g] = j < n_valid ? src.template make<REPORT>(raw[g], t0 + j) : ~0ull;
        }
    }
#pragma unroll
    for (int b = 0; b < IPT; ++b)
        if (w * SHARE + b * 64 + lane < n_valid) atomicAdd(&cnt[w][((uint32_t)it[b] >> shift) & 255u], 1);
    __syncthreads();
    {
        int c[kRsWaves], tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < kRsWaves; ++q) { c[q] = cnt[q][tid]; tot += c[q]; }
        }
        const int incl = wave_incl_scan_add(tot);
        if (tid < 256 && lane == 63) wsum[w] = incl;
        __syncthreads();
        if (tid < 256) {
            int lstart = incl - tot;
            for (int q = 0; q < w; ++q) lstart += wsum[q];
            int run = lstart;
#pragma unroll
            for (int q = 0; q < kRsWaves; ++q) { cnt[q][tid] = run; run += c[q]; }
            goff[tid] = tile_off[(long long)tile * 256 + tid] - lstart;
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < IPT; ++b) {
        const int j = w * SHARE + b * 64 + lane;
        const bool live = j < n_valid;
        if (__ballot(live) == 0ull) break;
        const int d = (int)(((uint32_t)it[b] >> shift) & 255u);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((d >> bit) & 1);
            peers &= ((d >> bit) & 1) ? m : ~m;
        }
        const int rank = (int)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = (int)__builtin_ctzll(peers | (1ull << 63));
        int first = 0;
        if (live && lane == leader) { first = cnt[w][d]; cnt[w][d] = first + (int)__popcll(peers); }
        first = __shfl(first, leader, kWave);
        if (live) s_item[first + rank] = it[b];
    }
    __syncthreads();
    for (int j = tid; j < n_valid; j += kRsThreads) {
        const unsigned long long x = s_item[j];
        out[goff[((uint32_t)x >> shift) & 255u] + j] = x;
    }
}

constexpr int kRsItemsIpt = 32;      // 8192 items of 8 bytes: 64 KB of LDS per tile, runs of 256 bytes
inline size_t rs_items_tmp_bytes(long long n) { return ((size_t)256 * (size_t)((n + RsGeom<kRsItemsIpt>::kTile - 1) / RsGeom<kRsItemsIpt>::kTile + 1) + (size_t)256 * kRsSegs) * 4; }
// the first pass reads `first` (n items: any Src), every pass writes into a / b in turn beginning with a; *in_b: the sorted items are in b
template <class First>
inline hipError_t radix_sort_items(hipStream_t st, First first, unsigned long long *a, unsigned long long *b, long long n, int bits, void *tmp, bool *in_b)
{
    constexpr int IPT = kRsItemsIpt;
    *in_b = true;                        // (nothing written yet: the roles swap before the first pass)
    const int n_tiles = (int)std::max<long long>(1, (n + RsGeom<IPT>::kTile - 1) / RsGeom<IPT>::kTile);
    int32_t *hist = static_cast<int32_t *>(tmp), *seg = hist + (size_t)256 * n_tiles;
    unsigned long long *xi = b, *xo = a;
    bool first_pass = true;
    for (int shift = 0; shift < bits || first_pass; shift += 8) {
        if (first_pass) hipLaunchKernelGGL((rs_hist_items_kernel<IPT, First>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, first, n, shift, n_tiles, hist);
        else hipLaunchKernelGGL((rs_hist_items_kernel<IPT, ItemArray>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, ItemArray{xi}, n, shift, n_tiles, hist);
        hipLaunchKernelGGL(rs_segsum_kernel, dim3(kRsSegs), dim3(256), 0, st, n_tiles, hist, seg);
        hipLaunchKernelGGL(rs_segscan_kernel, dim3(1), dim3(256), 0, st, seg);
        hipLaunchKernelGGL(rs_tilescan_kernel, dim3(kRsSegs), dim3(256), 0, st, n_tiles, hist, seg);
        if (first_pass) hipLaunchKernelGGL((rs_scatter_items_kernel<IPT, First, true>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, first, n, shift, n_tiles, hist, xo);
        else hipLaunchKernelGGL((rs_scatter_items_kernel<IPT, ItemArray, false>), dim3(rs_grid(n_tiles)), dim3(kRsThreads), 0, st, ItemArray{xi}, n, shift, n_tiles, hist, xo);
        std::swap(xi, xo);
        *in_b = !*in_b;
        first_pass = false;
    }
    return hipGetLastError();
}

} // namespace raft
