// pileup.hpp -- what the pileup kernels share: error flags, the kernels' argument block, the tile boundaries and the kernel that
// finds them.
//
// Reference semantics the kernels reproduce (closed forms of SURVEY.md §3.2, checked against oracle/raft_oracle.c and the compiled
// reference):
//   profileCoverage  repeat.hpp:28-79   interval (s,e) adds 1 to windows s/reso .. (e-1)/reso
//   run scan         repeat.hpp:111-168 maximal runs of windows with cov >= high_cov, kept when (#windows*reso) >= repeat_length,
//                                       widened by flanking_length and clamped to [0,len]
//
// The kernels: pileup_wave.hpp (one wave per tile, 16-bit difference array: the pass's dominant kernel) and pileup_deep.hpp (the
// rare tile with 2^15 intervals or more, 32-bit, a workgroup per tile).  The workgroup-tile kernels of rounds 1-3 (pileup_kernel,
// pileup_fast_kernel: int32 LDS windows, descriptors per tile, re-cut tiles) were the fallback for deep tiles until round 5 and are
// gone with it (docs/history.md describes them).
//
// Algorithmic bytes per launch: 12*I + 4*B + 4*N + 8*R.
#pragma once
#include "wave.hpp"
#include "raft_types.hpp"

namespace raft {

// window index of base n (0 <= n < 2^31): n / reso without a hardware divide
__device__ __forceinline__ unsigned win_of(const PileupArgs &a, unsigned n)
{
    return a.div_shift < 0 ? n : (__umulhi(n, a.div_magic) >> a.div_shift);
}

// s_waitcnt vmcnt(0) that the compiler's wait-count pass can see (it then knows every earlier load has landed)
__device__ __forceinline__ void wait_all_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ void raise_error(const PileupArgs &a, int flag, long long idx)
{
    atomicOr(a.err_flags, flag);
    atomicMin((unsigned long long *)a.err_index, (unsigned long long)idx);
}

// A window at or above the limit of the one- or two-byte encoding (rare by the choice of the width), or a window the four-bit
// encoding lists with its value.  (Inlined: a call from the row loop costs the kernel 46 registers and a stack.)
__device__ __forceinline__ void note_exception(const PileupArgs &a, long long window, int v)
{
    const unsigned long long slot = atomicAdd(a.n_exc, 1ull);
    if ((long long)slot < a.exc_cap) { a.exc_idx[slot] = window; a.exc_val[slot] = v; }
}

// lower bound of read id `r` in iv_rid[lo, hi)
__device__ __forceinline__ long long lower_bound_rid(const int32_t *iv_rid, long long lo, long long hi, int r)
{
    while (lo < hi) {
        long long mid = (lo + hi) >> 1;
        if (iv_rid[mid] < r) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per range boundary (the boundaries a worker of pileup_wave_kernel draws: TileCut): the boundary's first read, that
// read's first window, and its first record in every sorted run -- looked up in the caller's offsets (grouped input) or in the
// counting sort's, or searched: the samples guess_runs_kernel kept first (a coarse index, cache-resident), then the ~1 k records
// between two samples.  A range's records end where the next range's begin, so each lane searches once per run and takes the end
// from its neighbour lane (the last lane of a wave searches twice).
__global__ __launch_bounds__(256) void tile_desc_kernel(long long n_tiles, SegStarts sb, const long long *seg_end_dev,
                                                        const int32_t *iv_rid, const int32_t *tile_first,
                                                        const long long *cov_off, TileCut *cuts, const int32_t *samples, long long n_rec,
                                                        const long long *bucket_off, int32_t *err_flags, MirrorArgs mir,
                                                        GroupedOff grp, const GuessOut *verify_guess)
{
    if (*(volatile int32_t *)err_flags & kErrStop) return;   // (sizes or offsets the device found wrong: nothing here is safe)
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (verify_guess && k == 0) {
        // a pass built on the run ends the context's LAST pass over a stream of this shape sampled (engine.hip run_pass `speculate`):
        // this pass's own samples must say the same, or the pass is not what it was built for
        const int nd = verify_guess->n_desc;
        bool same = nd == sb.n_seg - 1;
        for (int i = 0; same && i < nd; ++i) {
            bool found = false;
            for (int j = 1; j < sb.n_seg; ++j) found |= verify_guess->desc_pos[i] == sb.start[j];
            same = found;
        }
        if (!same) atomicOr(err_flags, kErrHint);
    }
    const bool live = k < n_tiles;
    const bool edge = k <= n_tiles;                 // boundary n_tiles closes the last tile
    const bool mirror = mir.tid && k == n_tiles + 1;   // (see MirrorArgs)
    struct { int32_t r_lo, r_hi; long long g_lo, g_hi; long long iv_lo[kMaxSeg]; int32_t n_iv[kMaxSeg]; } d{};
    if (live) {
        d.r_lo = tile_first[k]; d.r_hi = tile_first[k + 1];
        d.g_lo = cov_off[d.r_lo]; d.g_hi = cov_off[d.r_hi];
    } else if (edge) {
        d.r_lo = d.r_hi = tile_first[k];
        d.g_lo = d.g_hi = cov_off[d.r_lo];
    } else if (mirror) {
        d.r_lo = mir.tid[0]; d.r_hi = d.r_lo < INT32_MAX ? d.r_lo + 1 : d.r_lo;
    }
    // A tile's interval range ends where the next tile's begins: every lane takes that from its neighbour (the last
    // lane of a wave searches for its own end as well).  The kernel's time is the chain of dependent probes (~1.3 us
    // each into a GB-sized array), so all bisections of a thread advance together: each round issues the probes of
    // every segment -- and of the own-end searches -- back to back and only then looks at them.
    const bool own_end = (live && lane == 63) || mirror;
    long long blo[2 * kMaxSeg], bhi[2 * kMaxSeg];
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
        blo[s] = bhi[s] = blo[kMaxSeg + s] = bhi[kMaxSeg + s] = 0;
        if (s < sb.n_seg && !grp.off) {             // uniform
            const long long seg_e = seg_end_dev ? *seg_end_dev : sb.start[s + 1];
            if (edge || mirror) { blo[s] = sb.start[s]; bhi[s] = seg_e; }
            if (own_end) { blo[kMaxSeg + s] = sb.start[s]; bhi[kMaxSeg + s] = seg_e; }
        }
    }
    if (grp.off) {                                  // grouped input: the caller's offsets answer directly
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) {
            if (s < sb.n_seg) {
                if (edge) blo[s] = bhi[s] = grp.at(s, d.r_lo);
                if (own_end) blo[kMaxSeg + s] = bhi[kMaxSeg + s] = grp.at(s, d.r_hi);
            }
        }
    }
    // Before bisecting the record stream itself: the counting sort's own offsets answer directly; on the sorted-segment
    // path the 256 k samples guess_runs_kernel kept (a coarse index, cache-resident) are bisected first, which leaves the
    // ~1 k records between two samples -- 10 probes into a handful of lines instead of 28 all over a GB-sized column.
    if (bucket_off) {
#pragma unroll
        for (int q = 0; q < 2 * kMaxSeg; ++q)
            if (blo[q] < bhi[q]) blo[q] = bhi[q] = bucket_off[q < kMaxSeg ? d.r_lo : d.r_hi];
    } else if (samples) {
        // (all searches of a thread advance together here too: one after the other, the wave waited for its last lane, which has
        // twice the searches.  Measured and dropped, round 5: EIGHT ways a round, seven probes at the eighths of what is left, here
        // and in the record stream below -- a third of the rounds and 109 us instead of 74: the kernel's time is the number of
        // lines its probes pull in, not the length of a thread's chain.  What helps is a finer index: kSamples.)
        const int sh = sample_shift(n_rec);
        int sx[2 * kMaxSeg], sy[2 * kMaxSeg], sjl[2 * kMaxSeg], sjh[2 * kMaxSeg];
#pragma unroll
        for (int q = 0; q < 2 * kMaxSeg; ++q) {
            sx[q] = sy[q] = sjl[q] = 0; sjh[q] = -1;
            if (blo[q] < bhi[q]) {
                // samples inside this run: jl = first at or after its start, jh = last before its end (the closing sample
                // n_samples - 1 repeats the last record and is not needed: the run's own end bounds the search)
                sjl[q] = (int)((blo[q] + (1LL << sh) - 1) >> sh); sjh[q] = (int)((bhi[q] - 1) >> sh);
                if (sjl[q] <= sjh[q]) { sx[q] = sjl[q]; sy[q] = sjh[q] + 1; }   // first sample in [jl, jh] with an id >= key, or jh + 1
            }
        }
        for (;;) {
            int v[2 * kMaxSeg];
            bool any = false;
#pragma unroll
            for (int q = 0; q < 2 * kMaxSeg; ++q) {
                v[q] = 0;
                if (sx[q] < sy[q]) { v[q] = samples[(sx[q] + sy[q]) >> 1]; any = true; }
            }
            if (!any) break;
#pragma unroll
            for (int q = 0; q < 2 * kMaxSeg; ++q) {
                if (sx[q] < sy[q]) {
                    const int m = (sx[q] + sy[q]) >> 1;
                    if (v[q] < (q < kMaxSeg ? d.r_lo : d.r_hi)) sx[q] = m + 1; else sy[q] = m;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2 * kMaxSeg; ++q) {
            if (blo[q] < bhi[q] && sjl[q] <= sjh[q]) {
                const long long x = sx[q];
                if (x > sjl[q]) blo[q] = ((x - 1) << sh) + 1;
                if (x <= sjh[q]) bhi[q] = x << sh;
            }
        }
    }
    for (;;) {
        int v[2 * kMaxSeg];
        bool any = false;
#pragma unroll
        for (int q = 0; q < 2 * kMaxSeg; ++q) {
            v[q] = 0;
            if (blo[q] < bhi[q]) { v[q] = iv_rid[(blo[q] + bhi[q]) >> 1]; any = true; }
        }
        if (!any) break;
#pragma unroll
        for (int q = 0; q < 2 * kMaxSeg; ++q) {
            if (blo[q] < bhi[q]) {
                const long long mid = (blo[q] + bhi[q]) >> 1;
                if (v[q] < (q < kMaxSeg ? d.r_lo : d.r_hi)) blo[q] = mid + 1; else bhi[q] = mid;
            }
        }
    }
    if (mir.tid && __ballot(mirror) != 0ull) {      // the mirror thread's wave: look at the records of record 0's target
        const int src = __ffsll((long long)__ballot(mirror)) - 1;
        const int32_t q0 = iv_rid[0], t0 = mir.tid[0], qs0 = mir.qs[0], qe0 = mir.qe[0], ts0 = mir.ts[0], te0 = mir.te[0];
        bool hit = false;
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) {
            const long long lo = __shfl(blo[s], src, kWave), hi = __shfl(blo[kMaxSeg + s], src, kWave);
            if (s < sb.n_seg)
                for (long long i = lo + lane; i < hi; i += kWave)
                    hit |= i > 0 && iv_rid[i] == t0 && mir.tid[i] == q0 && mir.ts[i] == qs0 && mir.te[i] == qe0 &&
                           mir.qs[i] == ts0 && mir.qe[i] == te0;
        }
        if (__ballot(hit) != 0ull && lane == 0) *mir.found = 1;
    }
    // The cuts of a run tile it: the first tile begins where the run begins, the closing boundary is its end, and they
    // never step back.  On sorted runs the searches give exactly that; a pass that only trusts a sampled guess of the
    // runs (engine.hip run_pass) relies on it being enforced -- whatever the records are, every one of them then lies in
    // exactly one tile's range, and the pileup kernels flag those that belong to another tile's reads (kErrOrder).
    bool back = false;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
        if (s < sb.n_seg && !grp.off) {             // (grouped offsets begin and end where the runs do: checked by the scan's loader)
            const long long seg_e = seg_end_dev ? *seg_end_dev : sb.start[s + 1];
            if (k == 0) blo[s] = sb.start[s];
            if (k == n_tiles) blo[s] = seg_e;
            if (own_end && k + 1 == n_tiles) blo[kMaxSeg + s] = seg_e;
        }
        const long long lo = blo[s];
        long long hi = __shfl_down(lo, 1, kWave);
        if (own_end) hi = blo[kMaxSeg + s];
        if (live && hi < lo) { back = true; hi = lo; }
        // A tile in which no read begins is handed to no kernel, so it must not own records.  On sorted runs its range is
        // empty -- except behind the last read, where ids >= n_reads end up (the closing boundary is forced to the run's
        // end): such records refute the guess like any other record outside its tile's reads.
        if (live && d.r_hi == d.r_lo && hi > lo) back = true;
        d.iv_lo[s] = lo;
        d.n_iv[s] = (int)(hi - lo);
    }
    if (__ballot(back) != 0ull && lane == 0) atomicOr(err_flags, kErrOrder);
    if (edge) {
        TileCut c;
        c.r_lo = d.r_lo; c.flags = 0; c.g_lo = d.g_lo;
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) c.iv_lo[s] = (int32_t)d.iv_lo[s];
        cuts[k] = c;
    }
}

} // namespace raft
