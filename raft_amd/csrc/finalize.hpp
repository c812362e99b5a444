// finalize.hpp -- per-read tail of the path: order the read's repeats, mask the
// cut-point markers, count and emit fragments, reduce the stdout statistics.
//
// Reference: chop.hpp:209-321 (break_reads, integer half) and repeat.hpp:170.
// Work here is O(markers + repeats) per read -- a few integers against the
// hundreds of windows the pileup kernel wrote for the same read -- so one lane per
// read is enough; it is not on the roofline-relevant part of the pass.
#pragma once
#include "pileup.hpp"

namespace raft {

__device__ __forceinline__ void swap3(int32_t *k, int32_t *s, int32_t *e, long long i, long long j)
{
    int32_t t;
    t = k[i]; k[i] = k[j]; k[j] = t;
    t = s[i]; s[i] = s[j]; s[j] = t;
    t = e[i]; e[i] = e[j]; e[j] = t;
}

// in-place sort of a read's raw repeats by run start (keys are distinct: runs are disjoint)
__device__ void sort_repeats(int32_t *k, int32_t *s, int32_t *e, int n)
{
    if (n <= 24) {
        for (int i = 1; i < n; ++i) {
            const int32_t kk = k[i];
            if (k[i - 1] <= kk) continue;         // (in place already -- a read's runs mostly arrive in order: nothing to load or store)
            const int32_t ss = s[i], ee = e[i];
            int j = i - 1;
            while (j >= 0 && k[j] > kk) { k[j + 1] = k[j]; s[j + 1] = s[j]; e[j + 1] = e[j]; --j; }
            k[j + 1] = kk; s[j + 1] = ss; e[j + 1] = ee;
        }
        return;
    }
    // heap sort
    for (int root0 = n / 2 - 1; root0 >= 0; --root0) {
        int root = root0;
        for (;;) {
            int c = 2 * root + 1;
            if (c >= n) break;
            if (c + 1 < n && k[c + 1] > k[c]) ++c;
            if (k[root] >= k[c]) break;
            swap3(k, s, e, root, c);
            root = c;
        }
    }
    for (int end = n - 1; end > 0; --end) {
        swap3(k, s, e, 0, end);
        int root = 0;
        for (;;) {
            int c = 2 * root + 1;
            if (c >= end) break;
            if (c + 1 < end && k[c + 1] > k[c]) ++c;
            if (k[root] >= k[c]) break;
            swap3(k, s, e, root, c);
            root = c;
        }
    }
}

// repeat.hpp:170 is std::sort by the flanked, CLAMPED start.  After the ordering by run start above the list is
// non-decreasing in that key too, so there is nothing left to do -- unless several repeats clamp to 0 and the read has
// more than 16 repeats: libstdc++'s introsort (what the reference binary links) then permutes the tied entries, and
// long_repeats.txt/.bed list them in that order (the marker mask does not depend on it: tied entries all start at 0).
// The permutation is a function of the algorithm alone, restated here from its published form (bits/stl_algo.h:
// introsort loop with median-of-3 to first, unguarded partition, recursion on the right part, depth limit
// 2*floor(log2 n), heap-sort fallback; final insertion sort with threshold 16) and pinned against the real std::sort
// in oracle/ (tests/test_oracle_golden.py) and against the oracle on the device (tests/test_gpu_parity.py).
struct RepPair { int32_t s, e; };

struct RepView {                                  // (s[i], e[i]) of one read as one sequence
    int32_t *s, *e;
    __device__ RepPair get(int i) const { return RepPair{s[i], e[i]}; }
    __device__ void set(int i, RepPair v) const { s[i] = v.s; e[i] = v.e; }
    __device__ void swap(int i, int j) const { const RepPair a = get(i), b = get(j); set(i, b); set(j, a); }
    __device__ bool less(int i, int j) const { return s[i] < s[j]; }
};

__device__ inline void rep_adjust_heap(const RepView &v, int first, int hole, int len, RepPair value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (v.less(first + child, first + child - 1)) child--;
        v.set(first + hole, v.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v.set(first + hole, v.get(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                  // __push_heap
    while (hole > top && v.s[first + parent] < value.s) {
        v.set(first + hole, v.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v.set(first + hole, value);
}

__device__ inline void rep_unguarded_linear_insert(const RepView &v, int last)
{
    const RepPair val = v.get(last);
    int next = last - 1;
    while (val.s < v.s[next]) { v.set(last, v.get(next)); last = next; --next; }
    v.set(last, val);
}

__device__ inline void rep_insertion_sort(const RepView &v, int first, int last)
{
    for (int i = first + 1; i < last; ++i) {
        if (v.less(i, first)) {
            const RepPair val = v.get(i);
            for (int j = i; j > first; --j) v.set(j, v.get(j - 1));
            v.set(first, val);
        } else rep_unguarded_linear_insert(v, i);
    }
}

// (stk: 3 x 64 ints of the caller's -- LDS in the kernels: as arrays of its own they were 784 bytes of scratch per LANE of
// finalize_count_kernel, 50 KB per wave whether the wave met such a read or not)
constexpr int kSortStack = 3 * 64;
__device__ __noinline__ void rep_std_sort(int32_t *s, int32_t *e, int n, int32_t *stk)
{
    const RepView v{s, e};
    if (n <= 0) return;
    int lg = 0;
    for (int m = n; m > 1; m >>= 1) ++lg;
    // __introsort_loop: the recursive call (right part) becomes a stack entry; depth <= 2*lg <= 62
    int32_t *const stk_first = stk, *const stk_last = stk + 64, *const stk_depth = stk + 128;
    int sp = 0;
    stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = 2 * lg; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        // the callee's own loop: partition, hand the right part to a (pending) recursive call, continue on the left.
        // Right parts are independent of what happens on the left afterwards, so their order of execution is free.
        while (last - first > 16) {
            if (depth == 0) {                     // __partial_sort(first, last, last): make_heap + sort_heap
                const int len = last - first;
                for (int parent = (len - 2) / 2;; --parent) {
                    rep_adjust_heap(v, first, parent, len, v.get(first + parent));
                    if (parent == 0) break;
                }
                for (int end = last; end - first > 1;) {
                    --end;
                    const RepPair value = v.get(end);
                    v.set(end, v.get(first));
                    rep_adjust_heap(v, first, 0, end - first, value);
                }
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            if (v.less(a, b)) {
                if (v.less(b, c)) v.swap(first, b);
                else if (v.less(a, c)) v.swap(first, c);
                else v.swap(first, a);
            } else if (v.less(a, c)) v.swap(first, a);
            else if (v.less(b, c)) v.swap(first, c);
            else v.swap(first, b);
            int lo = first + 1, hi = last;
            for (;;) {
                while (v.less(lo, first)) ++lo;
                --hi;
                while (v.less(first, hi)) --hi;
                if (!(lo < hi)) break;
                v.swap(lo, hi);
                ++lo;
            }
            if (sp < 64) { stk_first[sp] = lo; stk_last[sp] = last; stk_depth[sp] = depth; ++sp; }
            last = lo;
        }
    }
    if (n > 16) {
        rep_insertion_sort(v, 0, 16);
        for (int i = 16; i < n; ++i) rep_unguarded_linear_insert(v, i);
    } else rep_insertion_sort(v, 0, n);
}

// Walks the candidate markers 0, L, 2L, ..., (+len) of one read (chop.hpp:209-223) and keeps
// the first, the last, and every marker not inside a flanked repeat [s,e] (chop.hpp:225-246).
// Calls keep(m) for each kept marker in ascending order; returns their number.
// A read's ordered, flanked repeats: in memory, or -- up to four of them, which is nearly every read that has any -- in registers
// (finalize_fill_kernel asks for them in one batch; one after the other, every repeat a sweep passes was a round trip of its thread)
struct RepMem {
    const int32_t *s, *e;
    __device__ __forceinline__ int S(int k) const { return s[k]; }
    __device__ __forceinline__ int E(int k) const { return e[k]; }
};
struct RepReg {
    int s0, s1, s2, s3, e0, e1, e2, e3;
    __device__ __forceinline__ int S(int k) const { return k == 0 ? s0 : k == 1 ? s1 : k == 2 ? s2 : s3; }
    __device__ __forceinline__ int E(int k) const { return k == 0 ? e0 : k == 1 ? e1 : k == 2 ? e2 : e3; }
};

template <class Rep, class Keep>
__device__ __forceinline__ int walk_cuts(int len, int L, const FastDiv &by_L, const Rep &rep, int n, Keep keep)
{
    const int parts = fdiv(by_L, len);
    const int nm = parts + 1 + ((len - parts * L) ? 1 : 0);
    int kept = 0, k = 0;
    for (int j = 0; j < nm; ++j) {
        const int m = (j <= parts) ? j * L : len;
        bool covered = false;
        if (j > 0 && j < nm - 1) {
            while (k < n && rep.E(k) < m) ++k; // repeats are ordered by start and by end
            covered = (k < n) && (rep.S(k) <= m);
        }
        if (!covered) { keep(m); ++kept; }
    }
    return kept;
}

// one read: order (and, for a long read, join) its repeats; count its kept markers and fragments
// (rep_bp_out: unclamped repeat bases of a read joined from pieces -- repeat.hpp:127,152 -- for the caller to add to the total:
// one atomic per READ on the one word serialised to 0.4 ms once a human-scale set had 1e4 such reads)
// (`live`: the lane has a read; every lane of the wave calls -- the lanes whose read needs the tie order below take turns with the
// wave's one stack.  stk: kSortStack ints of LDS per wave)
__device__ __forceinline__ void finalize_count_one(const FinalizeArgs &a, bool live, int r, int &n_out, int &nF_out, int &nf_out, long long &rep_bp_out,
                                                   int32_t *stk)
{
    n_out = nF_out = nf_out = 0;
    if (!live) r = 0;                             // (any read: nothing is stored for it)
    // (the read's three scalars before anything is stored: the sorts below write int32 arrays, which the compiler must assume to
    // overlap read_len -- a load behind them waits for them)
    int n = live ? a.rep_cnt[r] : 0;
    const long long base = a.rep_res_off[r];
    const int len = a.read_len[r];
    const int nbq = fdiv(a.by_reso, len);
    const int nb = nbq + ((len - nbq * a.reso) ? 1 : 0);
    const bool pieces = nb > a.long_windows;      // a long read, piled up in pieces
    // (Measured and dropped, round 5: up to four repeats asked for at once, ordered by a network in registers and swept from there --
    // 48.1 against 45.7 us at human scale, 102.7 against 100.2 on the ultralong set: nearly every wave holds a read with a repeat or
    // two, so every wave executed the network; the kernel's time is the instructions its waves execute, not its threads' chains.)
    if (n > 1) sort_repeats(a.raw_key + base, a.raw_s + base, a.raw_e + base, n);
    if (live) {
        if (pieces && n > 0) {
            // a long read, piled up in pieces: runs that meet at a piece boundary are one run (repeat.hpp:111-168 on the
            // whole read); then the length test, the flanks and the clamp, as pileup.hpp emit_run_of does for other reads
            int32_t *S = a.raw_s + base, *E = a.raw_e + base, *K = a.raw_key + base;
            int m = 0;
            long long rep_bp = 0;
            for (int i = 0; i < n;) {
                const int start = S[i];
                int end = E[i];
                int j = i + 1;
                while (j < n && S[j] == end) { end = E[j]; ++j; }
                if (end - start >= a.repeat_length) {
                    rep_bp += end - start;
                    int s = start - a.flank, e = end + a.flank;
                    if (s <= 0) s = 0;
                    if (e >= len) e = len;
                    K[m] = start; S[m] = s; E[m] = e; ++m;
                }
                i = j;
            }
            n = m;
            a.rep_cnt_rw[r] = m;
            rep_bp_out += rep_bp;
        }
    }
    {   // tied starts: repeat.hpp:170
        unsigned long long need = __ballot(n > 16 && a.raw_s[base + 1] == 0);
        while (need) {
            const int l = (int)__builtin_ctzll(need);
            need &= need - 1ull;
            if ((int)(threadIdx.x & 63u) == l) rep_std_sort(a.raw_s + base, a.raw_e + base, n, stk);
        }
    }
    if (!live) return;
    // Number of markers walk_cuts() keeps, without walking them: the interior markers are L, 2L, .., J*L; a flanked
    // repeat [s,e] covers the multiples of L inside it; repeats are ordered by start and by end, so the union is
    // counted in one sweep over the read's (few) repeats.
    const int L = a.interval_length;
    const int parts = fdiv(a.by_L, len);
    const int tail = (len - parts * L) ? 1 : 0;
    const int J = tail ? parts : parts - 1;       // last interior marker is J * L
    int covered = 0, done = 0;                    // multiples 1 .. done are accounted for
    auto sweep = [&](int s, int e) {
        int lo = s <= 0 ? 0 : fdiv(a.by_L, s + L - 1);
        lo = max(lo, done + 1);
        const int hi = e < 0 ? -1 : min(fdiv(a.by_L, e), J);
        if (hi >= lo) { covered += hi - lo + 1; done = hi; }
    };
    for (int k = 0; k < n && done < J; ++k) sweep(a.raw_s[base + k], a.raw_e[base + k]);
    const int nF = parts + 1 + tail - covered;
    int nf = 1;                                   // chop.hpp:250-276
    if (nF > a.div + 1) nf = fdiv(a.by_div, nF - 1 + a.div - 1);
    a.cut_cnt[r] = nF;
    a.frag_cnt[r] = nf;
    n_out = n; nF_out = nF; nf_out = nf;
}

// (Measured and dropped, round 3: a long read's raw runs ordered and joined by its whole wave -- a lane per run, rank by 64
// compare steps, joined through ballots -- instead of its thread's loops in global memory.  The long reads of a wave then
// take their turns, where the threads' loops had run side by side: finalize_count 0.11 -> 0.21 ms on the ultralong set.)
__global__ __launch_bounds__(256) void finalize_count_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
    __shared__ int32_t sort_stk[256 / 64][kSortStack];
    __shared__ long long scr[4][4];
    long long bp = 0;
    int n, nF, nf;
    finalize_count_one(a, r < a.n_reads, r, n, nF, nf, bp, sort_stk[threadIdx.x >> 6]);
    // one atomic per wave that has anything to add
    if (__ballot(bp != 0) != 0ull) {
        const long long s = wave_reduce_add64(bp);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.total_repeat, (unsigned long long)s);
    }
    // what the offsets and the totals are made of: this workgroup's sums (the kernel's one barrier, at its end)
    const long long v[4] = {wave_reduce_add64(n), wave_reduce_add64(nF), wave_reduce_add64(nf), wave_reduce_add64(r < a.n_reads ? a.read_len[r] : 0)};
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) scr[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) a.tail_part[(long long)threadIdx.x * a.tail_blocks + blockIdx.x] = scr[threadIdx.x][0] + scr[threadIdx.x][1] + scr[threadIdx.x][2] + scr[threadIdx.x][3];
}

// tail_part -> tail_prefix (exclusive) and the totals.  A workgroup per 1024 entries (N / 256 entries per array: 13 k for 3.3 M
// reads): it adds up the entries before its own by itself -- a few KB from L2, all workgroups at once -- and scans its 1024.  The
// last one also adds up what totals_kernel added up until round 5: the pileup workers' sums, the reads' lengths (the count kernel's
// fourth sum).  (ONE workgroup walking the array 1024 entries at a time took 53 us at that size; with its entries held in registers
// between two looks at them, 64.)
__global__ __launch_bounds__(1024) void tail_prefix_kernel(FinalizeArgs a, TailPublish tp)
{
    __shared__ long long scr[4][16], bas[4][16];
    const int lane = (int)threadIdx.x & 63, wid = (int)threadIdx.x >> 6;
    const bool stop = (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) != 0;
    const int nb = stop || a.n_reads == 0 ? 0 : a.tail_blocks;
    const int first = (int)blockIdx.x * 1024, i = first + (int)threadIdx.x;
    const bool closing = blockIdx.x == gridDim.x - 1;
    long long base[4] = {0, 0, 0, 0}, v[4], inc[4];
    for (int j = (int)threadIdx.x; j < min(first, nb); j += 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k) base[k] += a.tail_part[(long long)k * a.tail_blocks + j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = i < nb ? a.tail_part[(long long)k * a.tail_blocks + i] : 0;
        inc[k] = wave_incl_scan_add64(v[k]);
        base[k] = wave_reduce_add64(base[k]);
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { scr[k][wid] = inc[k]; bas[k][wid] = base[k]; }
    }
    __syncthreads();
    long long run[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long x = lane < 16 ? scr[k][lane] : 0, y = lane < 16 ? bas[k][lane] : 0;
        const long long all = wave_reduce_add64(x), before = wave_reduce_add64(lane < wid ? x : 0), b0 = wave_reduce_add64(y);
        if (k < 3 && i < nb) a.tail_prefix[(long long)k * a.tail_blocks + i] = b0 + before + inc[k] - v[k];
        run[k] = b0 + all;
    }
    if (!closing) return;
    long long v0 = 0, v1 = 0;                     // coverage, repeat bp: the sums of the pileup workers
    for (long long j = threadIdx.x; j < tp.n_tiles; j += 1024) { v0 += tp.tile_sums[2 * j]; v1 += tp.tile_sums[2 * j + 1]; }
    v0 = wave_reduce_add64(v0); v1 = wave_reduce_add64(v1);
    __syncthreads();
    if (lane == 0) { scr[0][wid] = v0; scr[1][wid] = v1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long c0 = 0, c1 = 0;
        for (int w = 0; w < 16; ++w) { c0 += scr[0][w]; c1 += scr[1][w]; }
        tp.totals[0] += (unsigned long long)c0; tp.totals[1] += (unsigned long long)c1;     // ([1]: finalize_count_kernel's joined long reads are in already)
        tp.totals[2] += (unsigned long long)run[3];
        tp.tails[0] = run[0]; tp.tails[1] = run[1]; tp.tails[2] = run[2];
        tp.tails[3] = tp.bucket_off ? tp.bucket_off[a.n_reads] : 0;
        a.rep_off_w[a.n_reads] = run[0]; a.cut_off_w[a.n_reads] = run[1]; a.frag_off_w[a.n_reads] = run[2];
    }
}

// the pass's last kernel: one wave copies the control block -- everything raft_hip_finish reports -- into the context's page-locked
// block, stamped with the pass's number: raft_hip_finish looks for it itself instead of sleeping in the runtime's wait (whose
// wake-up is 20-30 us of a pass that may take 200)
__global__ __launch_bounds__(64) void publish_ctrl_kernel(TailPublish tp);

// Compact repeats and the fragments of every read.  The cut points themselves (chop.hpp's final_stars, 4 B per
// marker: 0.4 GB on the human-scale set) are neither stored nor walked here: fragment j begins at the kept marker
// with index (j-1)*div and ends at the one with index j*div, the first marker is 0 and the last is the read length.
// finalize_cuts_kernel materialises them when a caller asks for them.
template <class Rep>
__device__ __forceinline__ void finalize_fill_one(const FinalizeArgs &a, int r, long long ro, long long fo, int n, int nF, int len, const Rep &rep)
{
    for (int i = 0; i < n; ++i) { a.rep_s[ro + i] = rep.S(i); a.rep_e[ro + i] = rep.E(i); }
    if (nF <= a.div + 1) {                        // chop.hpp:250-267: the read is kept whole
        a.frag_read[fo] = r; a.frag_begin[fo] = 0; a.frag_end[fo] = len;
        return;
    }
    const int nf = fdiv(a.by_div, nF - 1 + a.div - 1);  // chop.hpp:280-321
    // Fragment j ends, and fragment j + 1 begins, at the kept marker with index t = j * div (an interior marker).
    // Without repeats that is the multiple t * L; each flanked repeat removes the multiples inside it, so the t-th kept
    // multiple is t plus the sizes of the covered ranges that begin at or before it -- the same sweep as in
    // finalize_count_one, and ONE sweep for all fragments: t grows with j, so a range passed for one fragment is passed for
    // every later one (a 1.5 Mb read has 75 fragments and dozens of repeats, and its thread is what the kernel waits for).
    const int L = a.interval_length;
    const int parts = fdiv(a.by_L, len);
    const int J = (len - parts * L) ? parts : parts - 1;
    a.frag_read[fo] = r; a.frag_begin[fo] = 0;
    int k = 0, done = 0, passed = 0;              // next repeat, last covered multiple, covered multiples passed so far
    for (int j = 1; j < nf; ++j) {
        int v = j * a.div + passed;
        while (k < n && done < J) {
            const int s = rep.S(k), e = rep.E(k);
            int lo = s <= 0 ? 0 : fdiv(a.by_L, s + L - 1);
            lo = max(lo, done + 1);
            const int hi = e < 0 ? -1 : min(fdiv(a.by_L, e), J);
            if (hi >= lo) {
                if (v < lo) break;                // (this fragment ends before the range; a later one may pass it)
                v += hi - lo + 1; passed += hi - lo + 1;
                done = hi;
            }
            ++k;
        }
        const int m = v * L;
        const int begin = m - a.overlap_length;
        if (begin < 0 || begin > len) {
            atomicOr(a.err_flags, kErrFragment);
            atomicMin((unsigned long long *)a.err_index, (unsigned long long)r);
        }
        a.frag_end[fo + j - 1] = m;
        a.frag_read[fo + j] = r; a.frag_begin[fo + j] = begin;
    }
    a.frag_end[fo + nf - 1] = len;                // the last kept marker is the read's end
}

// CUTS: the pass also writes the cut points themselves (chop.hpp:225-246 final_stars) -- the default of a
// context (raft_hip_set_emit_cuts); the host pipelines, whose outputs hold no cut points, leave them to finalize_cuts_kernel.
// Round 6: the kernel makes the three offset arrays itself (see FinalizeArgs::tail_part), adds up what totals_kernel added up and
// its last workgroup publishes the control block -- the pass's tail is count -> fill.
template <bool CUTS>
__global__ __launch_bounds__(256) void finalize_fill_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
    const int lane = (int)threadIdx.x & 63, wid = (int)threadIdx.x >> 6;
    const bool live = r < a.n_reads;
    __shared__ long long wtot[3][4];
    // everything the read's thread needs, asked for at once: behind the first store a load of an int32 array would wait for the
    // stores (they may overlap, as far as the compiler knows), and a thread's time here is its chain of dependent loads
    const int n = live ? a.rep_cnt[r] : 0, nF = live ? a.cut_cnt[r] : 0, nf = live ? a.frag_cnt[r] : 0, len = live ? a.read_len[r] : 0;
    const long long rbase = live ? a.rep_res_off[r] : 0;
    // where this read's outputs begin: the workgroup's base (tail_prefix_kernel), the reads of the workgroup before this one -- a scan
    // in the wave, the waves' totals through LDS (the kernel's one barrier, near its start)
    long long ex[3];
    {
        const long long v[3] = {n, nF, nf};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long inc = wave_incl_scan_add64(v[k]);
            ex[k] = inc - v[k] + a.tail_prefix[(long long)k * a.tail_blocks + blockIdx.x];
            if (lane == 63) wtot[k][wid] = inc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int w = 0; w < 3; ++w) ex[k] += w < wid ? wtot[k][w] : 0;
    if (!live) return;
    const long long ro = ex[0], co = ex[1], fo = ex[2];
    a.rep_off_w[r] = ro; a.cut_off_w[r] = co; a.frag_off_w[r] = fo;
    int32_t *F = a.cuts + co;
    int wr = 0;
    if (n <= 4) {                                 // (the reads without a repeat too: one instruction stream for nearly every wave)
        const int32_t *S = a.raw_s + rbase, *E = a.raw_e + rbase;
        const int i0 = n > 0 ? 0 : -1, i1 = min(1, n - 1), i2 = min(2, n - 1), i3 = n - 1;
        auto at = [&](const int32_t *p, int i) { return i >= 0 ? p[i] : 0; };
        const RepReg rep{at(S, i0), at(S, i1), at(S, i2), at(S, i3), at(E, i0), at(E, i1), at(E, i2), at(E, i3)};
        finalize_fill_one(a, r, ro, fo, n, nF, len, rep);
        if (CUTS) (void)walk_cuts(len, a.interval_length, a.by_L, rep, n, [&](int m) { F[wr++] = m; });
    } else {
        const RepMem rep{a.raw_s + rbase, a.raw_e + rbase};
        finalize_fill_one(a, r, ro, fo, n, nF, len, rep);
        if (CUTS) (void)walk_cuts(len, a.interval_length, a.by_L, rep, n, [&](int m) { F[wr++] = m; });
    }
}

__global__ __launch_bounds__(64) void publish_ctrl_kernel(TailPublish tp)
{
    publish_stamped(tp.host_block, [&](int i) { return reinterpret_cast<const volatile long long *>(tp.ctrl_words)[i]; }, tp.n_ctrl_words, tp.pass_seq, (int)threadIdx.x);
    __threadfence_system();
}

// (Measured and dropped: count, offsets and fill in ONE launch -- a single-pass scan with decoupled look-back over
// per-workgroup sums, 256 or 1024 reads per workgroup.  Bit-exact on the whole suite, but 0.07-0.17 ms SLOWER than the
// five-launch chain on the human-scale set: every resident workgroup reaches its look-back at about the same time and
// walks over all the others, and the fill cannot start before that.)

// cut points of every read (final_stars), on demand
__global__ __launch_bounds__(256) void finalize_cuts_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
    int32_t *F = a.cuts + a.cut_off[r];
    int w = 0;
    (void)walk_cuts(a.read_len[r], a.interval_length, a.by_L, RepMem{a.raw_s + a.rep_res_off[r], a.raw_e + a.rep_res_off[r]}, a.rep_cnt[r],
                    [&](int m) { F[w++] = m; });
}

// totals[0] = sum coverage, [1] = sum unclamped repeat bp, [2] = sum read length.  One atomic per workgroup and
// total (a few hundred per launch): with one per wave the three counters saw 12 k serialised atomics and the kernel
// took 56 us for 13 MB of input.
} // namespace raft
