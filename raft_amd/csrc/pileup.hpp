// pileup.hpp -- the dominant kernel: binned coverage pileup + prefix scan +
// coalesced coverage store + high-coverage run detection, one tile of reads per
// workgroup.
//
// Reference semantics reproduced (closed forms of SURVEY.md §3.2, checked
// against oracle/raft_oracle.c and the compiled reference):
//   profileCoverage  repeat.hpp:28-79   interval (s,e) adds 1 to windows s/reso .. (e-1)/reso
//   run scan         repeat.hpp:111-168 maximal runs of windows with cov >= high_cov,
//                                       kept when (#windows*reso) >= repeat_length,
//                                       widened by flanking_length and clamped to [0,len]
//
// Work decomposition (MI355X-first, HBM-write bound):
//   * the output coverage array cov[] (4 B per window, all reads concatenated in
//     FASTA order) is cut into tiles by a quantum of Q windows; tile k owns the
//     reads whose first window falls in [kQ,(k+1)Q)  (tile_first[], host-free);
//   * a workgroup stages the tile's windows in LDS as a difference array:
//     +1 at the first window of an interval, -1 one past its last (ds_add_u32),
//     so a plain prefix sum over the concatenated reads yields every read's
//     coverage (each read's +1/-1 balance out before the next read begins);
//   * each wave owns a contiguous quarter of the tile: it prefix-sums rows of
//     256 windows (int4 per lane, DPP wave scan), and streams them to HBM as
//     1 KiB-per-instruction aligned stores -- coverage is written exactly once
//     and never read back;
//   * the >= high_cov predicate of a row is four 64-bit ballots; run starts /
//     run ends are found with scalar bit logic on those masks, so rows without
//     any high window (the common case) cost no vector work for the repeat scan;
//   * runs crossing wave seams (and chunk seams of reads longer than the LDS
//     capacity) are stitched by one lane from four words per wave.
//
// Algorithmic bytes per launch: 12*I + 4*B (+ 8 per read of offsets).
#pragma once
#include "wave.hpp"

namespace raft {

constexpr int kMaxSeg = 8;

enum : int {
    kErrReadId = 1 << 0,
    kErrCoord = 1 << 1,
    kErrFragment = 1 << 2,
    kErrInternal = 1 << 3,
    kErrLen = 1 << 4
};

struct PileupArgs {
    // intervals: sorted by read id inside each of n_seg segments
    const int32_t *iv_rid, *iv_s, *iv_e;
    int32_t n_seg;
    const long long *tile_iv;   // [n_seg][n_tiles+1] first interval of segment s belonging to tile k
    // reads
    const int32_t *read_len;
    const long long *cov_off;   // [n_reads+1]
    const int32_t *tile_first;  // [n_tiles+1]
    long long n_tiles_p1;       // stride of tile_iv rows
    int32_t n_reads;
    // params
    int32_t reso, high_cov, repeat_length, flank;
    // outputs
    int32_t *cov;
    const long long *rep_res_off; // [n_reads+1] reserved slots for raw repeats
    int32_t *rep_cnt;             // [n_reads], zeroed
    int32_t *raw_key, *raw_s, *raw_e;
    long long *tile_sums;         // [2*n_tiles]: sum of coverage, sum of unclamped repeat bp
    int32_t *err_flags;           // device word, OR of kErr*
    long long *err_index;         // first offending interval index (min)
};

constexpr int kOpen = -2; // run began before this wave's first window
constexpr int kNone = -1;

template <int THREADS, int CAP>
struct PileupSmem {
    static constexpr int NW = THREADS / 64;
    static constexpr int SLOTS = CAP + 256; // window slots: 3 alignment + CAP + 1 sentinel, rounded to rows
    static constexpr int SBW = SLOTS / 32;
    int32_t diff[SLOTS];
    uint32_t sbits[SBW];
    long long iv_lo[kMaxSeg], iv_hi[kMaxSeg];
    unsigned long long acc_cov, acc_rep;
    long long carry_open;
    int32_t carry_hp;
    int32_t wsum[NW];
    int32_t w_rows[NW], w_pclose[NW], w_sfinal[NW], w_hpfinal[NW], w_hpin[NW];
};

__device__ __forceinline__ void raise_error(const PileupArgs &a, int flag, long long idx)
{
    atomicOr(a.err_flags, flag);
    atomicMin((unsigned long long *)a.err_index, (unsigned long long)idx);
}

// read (in [r_a, r_b)) that owns global window g; reads with zero windows are skipped
__device__ __forceinline__ int owner_of_window(const long long *cov_off, int r_a, int r_b, long long g)
{
    int lo = r_a, hi = r_b; // invariant: cov_off[lo] <= g < cov_off[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (cov_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// A closed run of high windows [gS, gT) (global window indices) -> one raw repeat record.
template <class Smem>
__device__ __forceinline__ void emit_run(const PileupArgs &a, Smem &sm, int r_a, int r_b, bool single_read,
                                         long long gS, long long gT)
{
    const long long nwin = gT - gS;
    if (nwin * (long long)a.reso < (long long)a.repeat_length) return; // repeat.hpp:125,150
    const int rid = single_read ? r_a : owner_of_window(a.cov_off, r_a, r_b, gS);
    const long long c0 = a.cov_off[rid];
    const int len = a.read_len[rid];
    const int start = (int)(gS - c0) * a.reso;
    const int end = start + (int)nwin * a.reso;
    int s = start - a.flank, e = end + a.flank;   // repeat.hpp:129-140
    if (s <= 0) s = 0;
    if (e >= len) e = len;
    const int slot = atomicAdd(&a.rep_cnt[rid], 1);
    const long long cap = a.rep_res_off[rid + 1] - a.rep_res_off[rid];
    if (slot >= cap) { raise_error(a, kErrInternal, rid); return; }
    const long long idx = a.rep_res_off[rid] + slot;
    a.raw_key[idx] = start;
    a.raw_s[idx] = s;
    a.raw_e[idx] = e;
    atomicAdd(&sm.acc_rep, (unsigned long long)(end - start)); // repeat.hpp:127,152
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

// One LDS window: global windows [w_lo, w_hi) (at most CAP) belonging to reads [r_a, r_b).
// single_read: the window is a chunk of one long read r_a (intervals are clipped to the chunk).
template <int THREADS, int CAP>
__device__ void pile_window(const PileupArgs &a, PileupSmem<THREADS, CAP> &sm, int r_a, int r_b,
                            long long w_lo, long long w_hi, bool single_read, bool first_chunk, bool last_chunk)
{
    using Smem = PileupSmem<THREADS, CAP>;
    constexpr int NW = Smem::NW;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long a0 = w_lo & ~3LL;          // 16-byte aligned base of the staged range
    const int off0 = (int)(w_lo - a0);         // first valid slot
    const int t_end = off0 + (int)(w_hi - w_lo); // one past the last valid slot
    const int rows = (t_end + 1 + 255) >> 8;   // rows of 256 slots, sentinel slot included

    // 1. clear the difference array and the read-start bits
    for (int i = tid * 4; i < rows * 256; i += THREADS * 4)
        *reinterpret_cast<int4 *>(&sm.diff[i]) = make_int4(0, 0, 0, 0);
    for (int i = tid; i < rows * 8; i += THREADS) sm.sbits[i] = 0u;
    __syncthreads();

    // 2. read-start bits (a run never continues across a read boundary, repeat.hpp:111-112)
    if (single_read) {
        if (first_chunk && tid == 0) sm.sbits[0] = 1u << off0;
    } else {
        for (int rr = r_a + tid; rr < r_b; rr += THREADS) {
            const int p = (int)(a.cov_off[rr] - a0);
            atomicOr(&sm.sbits[p >> 5], 1u << (p & 31));
        }
    }

    // 3. intervals -> +1 / -1 (profileCoverage, closed form)
    for (int s = 0; s < a.n_seg; ++s) {
        const long long lo = sm.iv_lo[s], hi = sm.iv_hi[s];
        for (long long i = lo + tid; i < hi; i += THREADS) {
            const int rid = a.iv_rid[i];
            const int st = a.iv_s[i];
            const int en = a.iv_e[i];
            if ((st | en) < 0) { raise_error(a, kErrCoord, i); continue; }
            const int first = st / a.reso;
            int last = (en > 0) ? (en - 1) / a.reso : -1;
            if (last < first) continue;
            const long long c0 = a.cov_off[rid];
            const int nb_r = (int)(a.cov_off[rid + 1] - c0);
            if (last >= nb_r) {                 // reference writes past its vector here (repeat.hpp:69-72)
                raise_error(a, kErrCoord, i);
                last = nb_r - 1;
                if (last < first) continue;
            }
            long long gf = c0 + first, gl1 = c0 + last + 1;
            if (single_read) {
                if (gf < w_lo) gf = w_lo;
                if (gl1 > w_hi) gl1 = w_hi;
                if (gf >= gl1) continue;
            }
            __hip_atomic_fetch_add(&sm.diff[(int)(gf - a0)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sm.diff[(int)(gl1 - a0)], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();

    // 4. pass A: per-wave sums of the difference array (each wave owns rpw contiguous rows)
    const int rpw = (rows + NW - 1) / NW;
    const int row_b = wid * rpw;
    const int row_e = min(rows, row_b + rpw);
    {
        int s = 0;
        for (int row = row_b; row < row_e; ++row) {
            const int4 d = *reinterpret_cast<const int4 *>(&sm.diff[row * 256 + lane * 4]);
            s += d.x + d.y + d.z + d.w;
        }
        s = wave_reduce_add(s);
        if (lane == 0) sm.wsum[wid] = s;
    }
    __syncthreads();

    // 5. pass B: prefix sum, store, run detection
    int carry = 0;
    for (int w = 0; w < wid; ++w) carry += sm.wsum[w];
    bool hp; // was the window just before this wave's first slot high (and in the same run domain)?
    if (wid == 0) hp = single_read && !first_chunk && sm.carry_hp != 0;
    else hp = (row_b < rows) && (carry >= a.high_cov);
    const bool hp_in = hp;
    int S = hp ? kOpen : kNone;  // start slot of the run currently open
    int pclose = -1;             // slot at which the run inherited from before this wave closed
    long long covsum = 0;

    for (int row = row_b; row < row_e; ++row) {
        const int base = row * 256;
        const int p0 = base + lane * 4;
        const int4 d = *reinterpret_cast<const int4 *>(&sm.diff[p0]);
        const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
        const int incl = wave_incl_scan_add(w);
        const int excl = incl - w + carry;
        carry += __builtin_amdgcn_readlane(incl, 63);
        int c0 = excl + x, c1 = excl + y, c2 = excl + z, c3 = excl + w;
        const bool full = (base >= off0) && (base + 256 <= t_end);
        unsigned long long M0, M1, M2, M3, VE0, VE1, VE2, VE3;
        if (full) {
            *reinterpret_cast<int4 *>(&a.cov[a0 + p0]) = make_int4(c0, c1, c2, c3);
            covsum += (long long)c0 + c1 + c2 + c3;
            M0 = __ballot(c0 >= a.high_cov); M1 = __ballot(c1 >= a.high_cov);
            M2 = __ballot(c2 >= a.high_cov); M3 = __ballot(c3 >= a.high_cov);
            VE0 = VE1 = VE2 = VE3 = ~0ull;
        } else {
            const bool v0 = (p0 + 0 >= off0) && (p0 + 0 < t_end);
            const bool v1 = (p0 + 1 >= off0) && (p0 + 1 < t_end);
            const bool v2 = (p0 + 2 >= off0) && (p0 + 2 < t_end);
            const bool v3 = (p0 + 3 >= off0) && (p0 + 3 < t_end);
            if (v0) { a.cov[a0 + p0 + 0] = c0; covsum += c0; }
            if (v1) { a.cov[a0 + p0 + 1] = c1; covsum += c1; }
            if (v2) { a.cov[a0 + p0 + 2] = c2; covsum += c2; }
            if (v3) { a.cov[a0 + p0 + 3] = c3; covsum += c3; }
            M0 = __ballot(v0 && c0 >= a.high_cov); M1 = __ballot(v1 && c1 >= a.high_cov);
            M2 = __ballot(v2 && c2 >= a.high_cov); M3 = __ballot(v3 && c3 >= a.high_cov);
            VE0 = __ballot(p0 + 0 < t_end); VE1 = __ballot(p0 + 1 < t_end);
            VE2 = __ballot(p0 + 2 < t_end); VE3 = __ballot(p0 + 3 < t_end);
        }
        if ((M0 | M1 | M2 | M3) == 0ull && !hp) continue; // no high window in or just before this row

        const uint32_t word = sm.sbits[p0 >> 5];
        const uint32_t nib = (word >> (p0 & 31)) & 0xFu;
        const unsigned long long SB0 = __ballot(nib & 1u), SB1 = __ballot(nib & 2u),
                                 SB2 = __ballot(nib & 4u), SB3 = __ballot(nib & 8u);
        // P_k: the slot before (lane,k) is a high window
        // the carried-in bit belongs to the first valid slot: slot off0 of row 0, else slot 0 of the row
        const unsigned long long hb = hp ? 1ull : 0ull;
        const int hk = (row == 0) ? off0 : 0;
        const unsigned long long P0 = (M3 << 1) | (hk == 0 ? hb : 0ull), P1 = M0 | (hk == 1 ? hb : 0ull),
                                 P2 = M1 | (hk == 2 ? hb : 0ull), P3 = M2 | (hk == 3 ? hb : 0ull);
        const unsigned long long CL0 = P0 & (~M0 | SB0) & VE0, CL1 = P1 & (~M1 | SB1) & VE1,
                                 CL2 = P2 & (~M2 | SB2) & VE2, CL3 = P3 & (~M3 | SB3) & VE3; // run ends before this slot
        const unsigned long long CA0 = M0 & (~P0 | SB0), CA1 = M1 & (~P1 | SB1),
                                 CA2 = M2 & (~P2 | SB2), CA3 = M3 & (~P3 | SB3);             // run starts at this slot
        if ((CL0 | CL1 | CL2 | CL3) != 0ull) {
            const unsigned long long bit = 1ull << lane, lt = bit - 1ull, le = lt | bit;
            const unsigned long long CLk[4] = {CL0, CL1, CL2, CL3};
            const unsigned long long CAk[4] = {CA0, CA1, CA2, CA3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (CLk[k] & bit) {
                    int best = S;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned long long m = CAk[j] & (j < k ? le : lt);
                        if (m) best = max(best, base + 4 * top_bit(m) + j);
                    }
                    const int t = p0 + k;
                    if (best == kOpen) pclose = t;
                    else emit_run(a, sm, r_a, r_b, single_read, a0 + best, a0 + t);
                }
            }
        }
        if (CA0) S = max(S, base + 4 * top_bit(CA0) + 0);
        if (CA1) S = max(S, base + 4 * top_bit(CA1) + 1);
        if (CA2) S = max(S, base + 4 * top_bit(CA2) + 2);
        if (CA3) S = max(S, base + 4 * top_bit(CA3) + 3);
        if (base + 256 <= t_end) hp = (M3 >> 63) != 0ull;
        else if (t_end > base) {
            const int tl = t_end - 1 - base; // last valid slot of the row
            const unsigned long long Mk = (tl & 3) == 0 ? M0 : (tl & 3) == 1 ? M1 : (tl & 3) == 2 ? M2 : M3;
            hp = ((Mk >> (tl >> 2)) & 1ull) != 0ull;
        }
    }

    // 6. hand the wave's seam state to the stitcher
    {
        const unsigned long long pm = __ballot(pclose >= 0);
        int pc = -1;
        if (pm) pc = __builtin_amdgcn_readlane(pclose, (int)__builtin_ctzll(pm));
        covsum = wave_reduce_add64(covsum);
        if (lane == 0) {
            sm.w_rows[wid] = row_e > row_b ? 1 : 0;
            sm.w_pclose[wid] = pc;
            sm.w_sfinal[wid] = S;
            sm.w_hpfinal[wid] = hp ? 1 : 0;
            sm.w_hpin[wid] = hp_in ? 1 : 0;
            if (covsum) atomicAdd(&sm.acc_cov, (unsigned long long)covsum);
        }
    }
    __syncthreads();

    // 7. stitch runs across wave seams and the window end
    if (tid == 0) {
        long long open = (single_read && !first_chunk) ? sm.carry_open : -1;
        for (int w = 0; w < NW; ++w) {
            if (!sm.w_rows[w]) continue;
            if (sm.w_hpin[w] && sm.w_pclose[w] >= 0 && open >= 0) {
                emit_run(a, sm, r_a, r_b, single_read, open, a0 + sm.w_pclose[w]);
                open = -1;
            }
            if (sm.w_hpfinal[w]) {
                if (sm.w_sfinal[w] != kOpen) open = a0 + sm.w_sfinal[w];
            } else open = -1;
        }
        if (last_chunk && open >= 0) { // end of read closes the run (repeat.hpp:150)
            emit_run(a, sm, r_a, r_b, single_read, open, w_hi);
            open = -1;
        }
        sm.carry_open = open;
        sm.carry_hp = open >= 0 ? 1 : 0;
    }
    __syncthreads();
}

template <int THREADS, int CAP>
__global__ __launch_bounds__(THREADS) void pileup_kernel(PileupArgs a)
{
    using Smem = PileupSmem<THREADS, CAP>;
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int k = blockIdx.x;
    const int tid = threadIdx.x;
    int r = a.tile_first[k];
    const int r_hi = a.tile_first[k + 1];
    if (tid == 0) { sm.acc_cov = 0ull; sm.acc_rep = 0ull; sm.carry_open = -1; sm.carry_hp = 0; }
    if (r >= r_hi) {
        if (tid == 0) { a.tile_sums[2 * (long long)k] = 0; a.tile_sums[2 * (long long)k + 1] = 0; }
        return;
    }
    const long long g_hi_all = a.cov_off[r_hi];
    while (r < r_hi) {
        const long long g_lo = a.cov_off[r];
        int r2;
        if (g_hi_all - g_lo <= CAP) r2 = r_hi;
        else {
            int lo = r, hi = r_hi; // cov_off[lo]-g_lo <= CAP < cov_off[hi]-g_lo
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (a.cov_off[mid] - g_lo <= CAP) lo = mid; else hi = mid;
            }
            r2 = lo;
        }
        const bool single = (r2 == r);
        const int r_b = single ? r + 1 : r2;
        // interval ranges of reads [r, r_b) in every segment
        __syncthreads();
        if (tid < a.n_seg) {
            const long long t_lo = a.tile_iv[(long long)tid * a.n_tiles_p1 + k];
            const long long t_hi = a.tile_iv[(long long)tid * a.n_tiles_p1 + k + 1];
            if (r == a.tile_first[k] && r_b == r_hi) { sm.iv_lo[tid] = t_lo; sm.iv_hi[tid] = t_hi; }
            else {
                sm.iv_lo[tid] = lower_bound_rid(a.iv_rid, t_lo, t_hi, r);
                sm.iv_hi[tid] = lower_bound_rid(a.iv_rid, t_lo, t_hi, r_b);
            }
        }
        __syncthreads();
        if (!single) {
            const long long w_hi = a.cov_off[r_b];
            if (w_hi > g_lo) pile_window<THREADS, CAP>(a, sm, r, r_b, g_lo, w_hi, false, true, true);
        } else {
            const long long g_end = a.cov_off[r + 1];
            for (long long c = g_lo; c < g_end; c += CAP) {
                const long long c_hi = (c + CAP < g_end) ? c + CAP : g_end;
                pile_window<THREADS, CAP>(a, sm, r, r + 1, c, c_hi, true, c == g_lo, c_hi == g_end);
            }
        }
        r = r_b;
    }
    if (tid == 0) {
        a.tile_sums[2 * (long long)k] = (long long)sm.acc_cov;
        a.tile_sums[2 * (long long)k + 1] = (long long)sm.acc_rep;
    }
}

} // namespace raft
