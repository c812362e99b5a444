// pileup_deep.hpp -- the tiles pileup_wave_kernel cannot take: 2^15 or more intervals on one tile.
//
// The reference has no depth limit: profileCoverage counts into std::vector<int> (repeat.hpp:39-44, 62-77), and an rDNA or
// satellite pile of a real human set is tens of thousands of overlaps deep on a handful of reads.  The wave kernel's difference
// array holds 16 bits per window (pileup_wave.hpp), so a tile that deep cannot go through it.  Until round 5 such a tile sent the
// WHOLE pass to the int32 kernels of rounds 1-3 (kErrDeep: a second pass at their speed, and two kernel generations kept alive
// for it).  Now the wave kernel cuts the tile as any other, checks its records, and -- instead of piling them up -- leaves the tile's
// description in a list; this kernel, launched behind it in every pass (a launch that finds the list empty costs its 4 us),
// takes the listed tiles one workgroup each, in plain 32-bit arithmetic:
//     records -> +1 / -1 on an int32 difference array in LDS (the tile's windows, at most the wave kernel's 4092)
//     -> prefix sum in place -> coverage out in the pass's encoding (int32, one / two bytes + listed windows, four-bit steps)
//     -> maximal runs of windows at or above high_cov, per read (repeat.hpp:111-168), flanked and clamped (:129-140) -- or, for a
//        piece of a read longer than a tile, unflanked with the piece-edge rule, to be joined by finalize_count_kernel.
// Nothing here is tuned: the tiles are rare and a workgroup's tile is ~4 k windows and ~4e4 records.  What matters is that every
// output is the one the wave kernel would have written had its counters been wide enough -- tests/test_gpu_deep.py runs the parity
// suites with RAFT_DEEP_MIN lowered so that ordinary tiles come this way.
#pragma once
#include "pileup_wave.hpp"
#include "wave_launch.hpp"

namespace raft {

constexpr int kDeepThreads = 256;
constexpr int kDeepSlots = kWaveSlots;       // the wave kernel's tile: 3 alignment slots + its windows + 1 sentinel

template <int SLOTS>
struct DeepSmem {
    int32_t cov[SLOTS + 8];          // difference array, then coverage, slot order (slot 0 = window a0 of cov[])
    int32_t roff[kWaveMaxReads + 2]; // first slot of read r_a + j (j <= nr)
    int32_t so[kMaxSeg][kWaveMaxReads + 2];   // window records: where read r_a + j's records begin in the tile's share of run s
    int32_t wsum[kDeepThreads / 64];
    long long rep_bp;
};

__global__ __launch_bounds__(kDeepThreads) void pileup_deep_kernel(PileupArgs a, const DeepTile *list, const int32_t *n_list, int32_t cap, int ow)
{
    __shared__ DeepSmem<kDeepSlots> sm;
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = min(*(volatile const int32_t *)n_list, cap);
    if (n <= 0 || (*(volatile int32_t *)a.err_flags & (kErrExtra | kErrStop))) return;
    const unsigned limit = ow == 1 ? 255u : 65535u;
    for (int b = (int)blockIdx.x; b < n; b += (int)gridDim.x) {
        const DeepTile t = list[b];
        const long long a0 = t.g_lo & ~3LL;
        const int off0 = (int)(t.g_lo - a0), t_end = off0 + t.nwin;
        __syncthreads();                                 // (the tile before is done with the arrays)
        for (int i = tid; i < kDeepSlots + 8; i += kDeepThreads) sm.cov[i] = 0;
        if (tid <= t.nr) {
            sm.roff[tid] = (int)(a.cov_off[t.r_a + tid] - a0);
            if (a.iv_w) {
                for (int s = 0; s < a.n_seg; ++s) sm.so[s][tid] = (int)(a.grp.at(s, t.r_a + tid) - a.grp.at(s, t.r_a));
            }
        }
        if (tid == 0) sm.rep_bp = 0;
        __syncthreads();
        // ---- records -> +1 / -1 (the wave kernel has checked them: ids, coordinates; what it flagged is not piled up here either)
        for (int s = 0; s < a.n_seg; ++s) {
            for (int i = tid; i < t.cnt[s]; i += kDeepThreads) {
                const long long at = (long long)t.lo[s] + i;
                int j, first, last1;
                bool ok = true;
                if (a.iv_w) {
                    const unsigned w = a.iv_w[at];
                    first = (int)(w & 0xffffu); last1 = (int)(w >> 16);
                    int lo = 0, hi = t.nr;               // last read whose records begin at or before i
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sm.so[s][mid] <= i) lo = mid; else hi = mid; }
                    j = lo;
                } else {
                    const int rid = a.iv_rid[at], st = a.iv_s[at], en = a.iv_e[at];
                    j = rid - t.r_a;
                    ok = (unsigned)j < (unsigned)t.nr && (st | en) >= 0 && en > 0;
                    first = (int)win_of(a, (unsigned)st); last1 = (int)win_of(a, (unsigned)(en - 1)) + 1;
                }
                if (!ok) continue;
                const int b0 = sm.roff[j], nb_r = sm.roff[j + 1] - b0;
                const int pf = max(b0 + first, off0), pl1 = min(b0 + min(last1, nb_r), t_end);
                if (pf < pl1) { atomicAdd(&sm.cov[pf], 1); atomicAdd(&sm.cov[pl1], -1); }
            }
        }
        __syncthreads();
        // ---- prefix sum in place: 16 consecutive slots per thread, the threads' totals through a wave scan and LDS
        {
            constexpr int PER = (kDeepSlots + kDeepThreads - 1) / kDeepThreads;
            int v[PER], sum = 0;
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int p = tid * PER + k; v[k] = p < kDeepSlots ? sm.cov[p] : 0; sum += v[k]; }
            const int incl = wave_incl_scan_add(sum);
            if (lane == 63) sm.wsum[wid] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wid; ++w) run += sm.wsum[w];
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int p = tid * PER + k; run += v[k]; if (p < kDeepSlots) sm.cov[p] = run; }
        }
        __syncthreads();
        // ---- coverage out, window by window, in the pass's encoding
        for (int p = off0 + tid; p < t_end; p += kDeepThreads) {
            const int c = sm.cov[p];
            const long long w = a0 + p;
            if (ow == 4) a.cov[w] = c;
            else if (ow == 1) { reinterpret_cast<uint8_t *>(a.covp)[w] = (uint8_t)min((unsigned)c, limit); if ((unsigned)c >= limit) note_exception(a, w, c); }
            else if (ow == 2) { reinterpret_cast<uint16_t *>(a.covp)[w] = (uint16_t)min((unsigned)c, limit); if ((unsigned)c >= limit) note_exception(a, w, c); }
            else {
                // four-bit steps (pack.hpp): the tile's first window is listed with its value (its predecessor is another tile's), and so
                // is every step outside [-7, 7]; the other nibbles of the byte, and of the word, may be a neighbouring tile's
                const int prev = p > off0 ? sm.cov[p - 1] : 0;
                const int step = c - prev;
                unsigned code = 0;
                if (p > off0 && (unsigned)(step + 7) <= 14u) code = (unsigned)(step + 8);
                else note_exception(a, w, c);
                unsigned *const word = reinterpret_cast<unsigned *>(a.covp) + (w >> 3);
                const unsigned sh = (unsigned)(w & 7) * 4u;
                atomicAnd(word, ~(0xFu << sh));
                atomicOr(word, code << sh);
                if (((w + a.d4_shift) & (kD4Block - 1)) == 0) a.cov_anchor[(w + a.d4_shift) >> 10] = prev;
            }
        }
        // ---- runs of windows at or above high_cov: a thread per read (a piece: one thread for the tile)
        long long bp = 0;
        if (t.piece) {
            if (tid == 0) {
                const int r0 = reinterpret_cast<const int32_t *>(a.rep_res_off + t.r_a)[0], r1 = reinterpret_cast<const int32_t *>(a.rep_res_off + t.r_a)[2];
                int S = -1;
                for (int p = off0; p <= t_end; ++p) {
                    const bool high = p < t_end && sm.cov[p] >= a.high_cov;
                    if (high && S < 0) S = p;
                    if (!high && S >= 0) {
                        // (a run that touches an edge of the piece may go on in its neighbour: kept whatever its length, joined later)
                        if ((long long)(p - S) * a.reso >= (long long)a.repeat_length || S == off0 || p == t_end) {
                            const int slot = atomicAdd(&a.rep_cnt[t.r_a], 1);
                            if (slot >= r1 - r0) raise_error(a, kErrInternal, t.r_a);
                            else {
                                const int start = (S - sm.roff[0]) * a.reso;
                                const long long ix = (long long)r0 + slot;
                                a.raw_key[ix] = start; a.raw_s[ix] = start; a.raw_e[ix] = start + (p - S) * a.reso;
                            }
                        }
                        S = -1;
                    }
                }
            }
        } else if (tid < t.nr) {
            const int r = t.r_a + tid;
            const int lo = max(sm.roff[tid], off0), hi = min(sm.roff[tid + 1], t_end);
            const int len = a.read_len[r];
            const int r0 = reinterpret_cast<const int32_t *>(a.rep_res_off + r)[0], r1 = reinterpret_cast<const int32_t *>(a.rep_res_off + r)[2];
            int S = -1, cnt = 0;
            for (int p = lo; p <= hi; ++p) {
                const bool high = p < hi && sm.cov[p] >= a.high_cov;
                if (high && S < 0) S = p;
                if (!high && S >= 0) {
                    const int nwin_r = p - S;
                    if ((long long)nwin_r * a.reso >= (long long)a.repeat_length) {
                        const int start = (S - sm.roff[tid]) * a.reso, end = start + nwin_r * a.reso;
                        int s2 = start - a.flank, e2 = end + a.flank;      // repeat.hpp:129-140
                        if (s2 <= 0) s2 = 0;
                        if (e2 >= len) e2 = len;
                        if (cnt >= r1 - r0) raise_error(a, kErrInternal, r);
                        else {
                            const long long ix = (long long)r0 + cnt;
                            a.raw_key[ix] = start; a.raw_s[ix] = s2; a.raw_e[ix] = e2;
                            ++cnt;
                            bp += end - start;
                        }
                    }
                    S = -1;
                }
            }
            if (cnt) a.rep_cnt[r] = cnt;
        }
        // (the tile's sum of coverage went into its wave worker's total with the records' check; the repeat bases are added here)
        if (__ballot(bp != 0) != 0ull) {
            const long long s = wave_reduce_add64(bp);
            if (lane == 0) atomicAdd(a.deep_rep_total, (unsigned long long)s);
        }
    }
}

} // namespace raft
