// pack.hpp -- transfer encoding of the coverage array, and the small kernels of the chunked host pipeline.
//
// cov[] (int32 per window, repeat.hpp:105-108's "pos,cov" pairs without the implied pos) is the one large output of
// the path: 8 GB at human scale, two thirds of the bytes that cross PCIe in an end-to-end pass.  The reference's
// consumer is a text formatter, and real coverage rarely needs more than a byte, so the host side takes
//     code[i] = min(cov[i], LIMIT)    plus a list of (window index, value) for the windows with cov >= LIMIT
// with LIMIT = 255 in one byte per window -- a quarter of the bytes -- or, for deep sets whose repeats pile up beyond
// that (60x with six-copy tandem arrays: half of the windows), LIMIT = 65535 in two; decoded exactly
// (raft_host_unpack_coverage_w / raft_host_write_coverage_packed_w).
// pack_cov_kernel makes it from an int32 cov[] on request (raft_hip_fetch_packed_w); a context whose output width is set
// to 1 or 2 (raft_hip_set_output_width) has the pileup kernel write it directly and never holds the int32 array at all
// unless asked for it (unpack_cov_kernel).
#pragma once
#include "raft_types.hpp"
#include "wave.hpp"

namespace raft {

template <class T>                // T = uint8_t (limit 255) or uint16_t (limit 65535): the encoding's width
struct PackOut {
    T *covp;
    unsigned long long *n_exc;   // windows with cov >= the limit (counted even when they no longer fit the list)
    long long exc_cap;
    long long *exc_idx;
    int32_t *exc_val;
};

template <class T> struct PackLimit;
template <> struct PackLimit<uint8_t> { static constexpr unsigned value = 255u; };
template <> struct PackLimit<uint16_t> { static constexpr unsigned value = 65535u; };

template <class T>
__device__ __forceinline__ void pack_note(int v, long long i, const PackOut<T> &o)
{
    if ((unsigned)v >= PackLimit<T>::value) {
        const unsigned long long slot = atomicAdd(o.n_exc, 1ull);
        if ((long long)slot < o.exc_cap) { o.exc_idx[slot] = i; o.exc_val[slot] = v; }
    }
}

// four windows -> four codes (coverage is never negative: every read's +1/-1 balance out)
template <class T>
__device__ __forceinline__ void pack4(const int4 v, long long i, const PackOut<T> &o, T *dst)
{
    constexpr unsigned L = PackLimit<T>::value;
    const unsigned a = min((unsigned)v.x, L), b = min((unsigned)v.y, L), c = min((unsigned)v.z, L), d = min((unsigned)v.w, L);
    if (((unsigned)v.x | (unsigned)v.y | (unsigned)v.z | (unsigned)v.w) >= L) {   // cheap pre-test: any value >= L makes the OR >= L
        pack_note(v.x, i, o); pack_note(v.y, i + 1, o); pack_note(v.z, i + 2, o); pack_note(v.w, i + 3, o);
    }
    if (sizeof(T) == 1) *reinterpret_cast<unsigned *>(dst) = a | (b << 8) | (c << 16) | (d << 24);
    else *reinterpret_cast<uint2 *>(dst) = make_uint2(a | (b << 16), c | (d << 16));
}

// 16 B in, 4 (or 8) B out per lane and step: 1 KiB wave loads; the pass is bound by the 4 B/window it reads.
template <class T>
__global__ __launch_bounds__(256) void pack_cov_kernel(const int32_t *__restrict__ cov, long long n_bins, PackOut<T> o)
{
    const long long n4 = n_bins >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int4 *src = reinterpret_cast<const int4 *>(cov);
    for (; g + 3 * stride < n4; g += 4 * stride) {       // four independent loads in flight per lane
        const int4 v0 = src[g], v1 = src[g + stride], v2 = src[g + 2 * stride], v3 = src[g + 3 * stride];
        pack4(v0, g << 2, o, o.covp + (g << 2)); pack4(v1, (g + stride) << 2, o, o.covp + ((g + stride) << 2));
        pack4(v2, (g + 2 * stride) << 2, o, o.covp + ((g + 2 * stride) << 2)); pack4(v3, (g + 3 * stride) << 2, o, o.covp + ((g + 3 * stride) << 2));
    }
    for (; g < n4; g += stride) pack4(src[g], g << 2, o, o.covp + (g << 2));
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n_bins & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const int v = cov[i];
        o.covp[i] = (T)min((unsigned)v, PackLimit<T>::value);
        pack_note(v, i, o);
    }
}

// The other direction, for a caller that asks for cov[] as int32 after a pass that wrote the encoding directly
// (pileup_wave.hpp OW = 1 / 2): codes widened, then the listed windows overwritten with their values.
template <class T>
__global__ __launch_bounds__(256) void unpack_cov_kernel(const T *__restrict__ codes, long long n_bins, int32_t *__restrict__ cov)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n_bins >> 2;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
        int4 v;
        if (sizeof(T) == 1) {
            const unsigned w = reinterpret_cast<const unsigned *>(codes)[g];
            v = make_int4((int)(w & 255u), (int)((w >> 8) & 255u), (int)((w >> 16) & 255u), (int)(w >> 24));
        } else {
            const uint2 w = reinterpret_cast<const uint2 *>(codes)[g];
            v = make_int4((int)(w.x & 65535u), (int)(w.x >> 16), (int)(w.y & 65535u), (int)(w.y >> 16));
        }
        reinterpret_cast<int4 *>(cov)[g] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n_bins & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        cov[i] = (int32_t)codes[i];
    }
}

__global__ __launch_bounds__(256) void scatter_exceptions_kernel(const long long *__restrict__ idx, const int32_t *__restrict__ val,
                                                                 long long n, int32_t *__restrict__ cov)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) cov[idx[i]] = val[i];
}

// ---- "delta4": four bits per window -----------------------------------------------------------------------------------
// Once the records cross PCIe as one word each, the coverage array is two thirds of a job's bytes -- and it barely moves
// from one window to the next (the difference array the pileup builds IS the step; 99.8 % of the steps of a 32x set lie
// within +-7).  Encoding, over the concatenated array cov[0 .. B):
//     nib[w]  (byte w >> 1, low nibble = even w)   step + 8 for a step cov[w] - cov[w-1] in [-7, 7] (cov[-1] = 0), 0 = escape
//     exceptions (w, cov[w])                        the ABSOLUTE value of every escaped window, ascending when handed out
//     anchor[k] = cov[1024 k - 1]  (anchor[0] = 0)  where a decoder may start: blocks of 1024 windows decode independently
// The pileup kernel writes it directly (pileup_wave.hpp OW = kCovDelta4): a step is the LDS difference array's own value,
// except at a tile's first window, whose predecessor another workgroup holds -- that window is always escaped.
// Decoding (raft_host_unpack_coverage_d4 / unpack_delta4_kernel): v = anchor[k]; per window v = escape ? listed value : v + step.
// (kCovDelta4, kD4Block: raft_types.hpp)

struct Delta4Out {
    uint8_t *nib;                 // ceil(B / 2) bytes (+ padding to a dword)
    int32_t *anchor;              // ceil(B / 1024) entries
    unsigned long long *n_exc;
    long long exc_cap;
    long long *exc_idx;
    int32_t *exc_val;
};

__device__ __forceinline__ unsigned delta4_code(int step, int value, long long w, bool force, const Delta4Out &o)
{
    if (!force && (unsigned)(step + 7) <= 14u) return (unsigned)(step + 8);
    const unsigned long long slot = atomicAdd(o.n_exc, 1ull);
    if ((long long)slot < o.exc_cap) { o.exc_idx[slot] = w; o.exc_val[slot] = value; }
    return 0u;
}

// The windows the pileup listed tile by tile (PileupArgs::exc_pidx: kExcPerTile slots per tile, exc_tile_n of them used) appended
// to the shared list.  One workgroup per 1024 tiles: the tiles' counts are scanned in the workgroup, ONE atomic reserves the
// room of all of them, and the slots are copied with a thread per slot -- a tile's entries lie together, consecutive tiles'
// entries land behind one another.  (First version: a thread per tile copying its own slots one by one, an atomic per wave:
// 0.21 ms at human scale, its 8-byte stores scattered at a tile's stride.)
constexpr int kCompactTiles = 1024;
__global__ __launch_bounds__(256) void compact_exceptions_kernel(long long n_tiles, int per_tile, const int32_t *__restrict__ tile_n,
                                                                 const long long *__restrict__ pidx, const int32_t *__restrict__ pval,
                                                                 unsigned long long *n_exc, long long exc_cap, long long *__restrict__ exc_idx,
                                                                 int32_t *__restrict__ exc_val)
{
    __shared__ int pre[kCompactTiles + 1];
    __shared__ int wsum[4];
    __shared__ long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long t0 = (long long)blockIdx.x * kCompactTiles;
    int c[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long t = t0 + tid * 4 + j;
        c[j] = t < n_tiles ? tile_n[t] : 0;
        sum += c[j];
    }
    const int incl = wave_incl_scan_add(sum);
    if (lane == 63) wsum[wid] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int x = wsum[w]; if (w < wid) before += x; total += x; }
    if (total == 0) return;                              // (uniform: most workgroups of the extra tiles' range)
    int run = before + incl - sum;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pre[tid * 4 + j] = run; run += c[j]; }
    if (tid == 255) pre[kCompactTiles] = run;
    if (tid == 0) base_s = (long long)atomicAdd(n_exc, (unsigned long long)total);
    __syncthreads();
    const long long base = base_s;
    const int slot = tid % per_tile, tile_in_group = tid / per_tile, tiles_per_group = 256 / per_tile;   // (per_tile divides 256)
    for (int g = 0; g < kCompactTiles; g += tiles_per_group) {
        if (pre[g + tiles_per_group] == pre[g]) continue;   // (uniform: none of these tiles listed a window)
        const int tl = g + tile_in_group;
        const int first = pre[tl], cnt = pre[tl + 1] - first;
        if (slot < cnt) {
            const long long at = base + first + slot, from = (t0 + tl) * per_tile + slot;
            if (at < exc_cap) { exc_idx[at] = pidx[from]; exc_val[at] = pval[from]; }
        }
    }
}

// int32 cov[] -> delta4 (after a pass that wrote int32: the caller asked late)
__global__ __launch_bounds__(256) void pack_delta4_kernel(const int32_t *__restrict__ cov, long long n_bins, Delta4Out o, int shift)
{
    const long long n4 = (n_bins + 3) >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += stride) {
        const long long w0 = g << 2;
        int prev = w0 > 0 ? cov[w0 - 1] : 0;
        if (((w0 + shift) & (kD4Block - 1)) == 0) o.anchor[(w0 + shift) >> 10] = prev;   // (shift: see PileupArgs::d4_shift)
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (w0 + k < n_bins) {
                const int v = cov[w0 + k];
                code |= delta4_code(v - prev, v, w0 + k, false, o) << (4 * k);
                prev = v;
            }
        }
        reinterpret_cast<uint16_t *>(o.nib)[g] = (uint16_t)code;
    }
}

// delta4 -> int32 cov[] on the device (tests, raft_hip_outputs_device after a pass that wrote the encoding): steps
// widened, the listed windows overwritten with their values and flagged, then one thread per block of 1024 windows walks it.
// Not a fast path: nothing in the product decodes on the device.
__global__ __launch_bounds__(256) void delta4_expand_kernel(const uint8_t *__restrict__ nib, long long n_bins, int32_t *__restrict__ cov,
                                                            unsigned *__restrict__ is_abs)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n32 = (n_bins + 31) >> 5;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n32; g += stride) {
        unsigned esc = 0;
        for (int k = 0; k < 32; ++k) {
            const long long w = (g << 5) + k;
            if (w < n_bins) {
                const unsigned c = (nib[w >> 1] >> (4 * (w & 1))) & 15u;
                cov[w] = c ? (int)c - 8 : 0;
                if (!c) esc |= 1u << k;
            }
        }
        is_abs[g] = esc;
    }
}

__global__ __launch_bounds__(256) void delta4_walk_kernel(long long n_bins, const int32_t *__restrict__ anchor, const unsigned *__restrict__ is_abs,
                                                          int32_t *__restrict__ cov)
{
    const long long n_blocks = (n_bins + kD4Block - 1) / kD4Block;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += stride) {
        int v = anchor[b];
        const long long w0 = b * kD4Block, w1 = min(n_bins, w0 + kD4Block);
        for (long long w = w0; w < w1; ++w) {
            const bool abs_v = (is_abs[w >> 5] >> (w & 31)) & 1u;    // (the listed value was scattered into cov[w] before this kernel)
            v = abs_v ? cov[w] : v + cov[w];
            cov[w] = v;
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
