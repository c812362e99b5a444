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

// n / d for 0 <= n < 2^31 and d >= 1 without a hardware divide (a 32-bit signed division is ~35 instructions; the sweeps
// below do two per repeat and fragment, and the longest read's thread is what a long-read set waits for): with
// L = ceil(log2 d) and m = floor(2^(31+L) / d) + 1 (< 2^32), n / d == mulhi(n, m) >> (L - 1)  (the identity pileup.hpp's
// win_of uses for the windows)
struct FastDiv {
    uint32_t magic;
    int32_t shift;                        // -1: d == 1
};
inline FastDiv make_fast_div(int d)
{
    FastDiv f{0u, -1};
    if (d > 1) {
        int L = 0;
        while ((1ull << L) < (unsigned long long)d) ++L;
        f.magic = (uint32_t)((1ull << (31 + L)) / (unsigned)d + 1ull);
        f.shift = L - 1;
    }
    return f;
}
__device__ __forceinline__ int fdiv(const FastDiv &f, int n) { return f.shift < 0 ? n : (int)(__umulhi((unsigned)n, f.magic) >> f.shift); }

struct FinalizeArgs {
    int32_t n_reads;
    const int32_t *read_len;
    const long long *rep_res_off;
    const int32_t *rep_cnt;
    int32_t *raw_key, *raw_s, *raw_e;     // sorted in place by finalize_count_kernel
    int32_t interval_length, div, overlap_length;
    FastDiv by_L, by_div, by_reso;        // interval_length, div, reso as divisors
    // reads with more than long_windows windows were piled up in pieces (pileup_fast.hpp emit_piece_run): their raw
    // records are unflanked [start, end) runs per piece, to be joined, tested, flanked and clamped here
    int32_t long_windows, reso, repeat_length, flank;
    int32_t *rep_cnt_rw;                  // (rep_cnt, writable: the joined count replaces the pieces' count)
    unsigned long long *total_repeat;     // repeat.hpp:127,152 for those reads
    int32_t *cut_cnt, *frag_cnt;          // [n_reads]
    const long long *rep_off, *cut_off, *frag_off; // [n_reads+1] (fill kernel)
    int32_t *rep_s, *rep_e, *cuts, *frag_read, *frag_begin, *frag_end;
    int32_t *err_flags;
    long long *err_index;
};

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
            const int32_t kk = k[i], ss = s[i], ee = e[i];
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

__device__ __noinline__ void rep_std_sort(int32_t *s, int32_t *e, int n)
{
    const RepView v{s, e};
    if (n <= 0) return;
    int lg = 0;
    for (int m = n; m > 1; m >>= 1) ++lg;
    // __introsort_loop: the recursive call (right part) becomes a stack entry; depth <= 2*lg <= 62
    int stk_first[64], stk_last[64], stk_depth[64], sp = 0;
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
template <class Keep>
__device__ __forceinline__ int walk_cuts(int len, int L, const int32_t *s, const int32_t *e, int n, Keep keep)
{
    const int parts = len / L;
    const int nm = parts + 1 + ((len % L) ? 1 : 0);
    int kept = 0, k = 0;
    for (int j = 0; j < nm; ++j) {
        const int m = (j <= parts) ? j * L : len;
        bool covered = false;
        if (j > 0 && j < nm - 1) {
            while (k < n && e[k] < m) ++k;     // repeats are ordered by start and by end
            covered = (k < n) && (s[k] <= m);
        }
        if (!covered) { keep(m); ++kept; }
    }
    return kept;
}

// one read: order (and, for a long read, join) its repeats; count its kept markers and fragments
// (rep_bp_out: unclamped repeat bases of a read joined from pieces -- repeat.hpp:127,152 -- for the caller to add to the total:
// one atomic per READ on the one word serialised to 0.4 ms once a human-scale set had 1e4 such reads)
__device__ __forceinline__ void finalize_count_one(const FinalizeArgs &a, int r, int &n_out, int &nF_out, int &nf_out, long long &rep_bp_out)
{
    int n = a.rep_cnt[r];
    const long long base = a.rep_res_off[r];
    if (n > 1) sort_repeats(a.raw_key + base, a.raw_s + base, a.raw_e + base, n);
    {
        const int len = a.read_len[r];
        const int nbq = fdiv(a.by_reso, len);
        const int nb = nbq + ((len - nbq * a.reso) ? 1 : 0);
        if (nb > a.long_windows && n > 0) {
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
    if (n > 16 && a.raw_s[base + 1] == 0) rep_std_sort(a.raw_s + base, a.raw_e + base, n);   // tied starts: repeat.hpp:170
    // Number of markers walk_cuts() keeps, without walking them: the interior markers are L, 2L, .., J*L; a flanked
    // repeat [s,e] covers the multiples of L inside it; repeats are ordered by start and by end, so the union is
    // counted in one sweep over the read's (few) repeats.
    const int len = a.read_len[r], L = a.interval_length;
    const int parts = fdiv(a.by_L, len);
    const int tail = (len - parts * L) ? 1 : 0;
    const int J = tail ? parts : parts - 1;       // last interior marker is J * L
    int covered = 0, done = 0;                    // multiples 1 .. done are accounted for
    for (int k = 0; k < n && done < J; ++k) {
        const int s = a.raw_s[base + k], e = a.raw_e[base + k];
        int lo = s <= 0 ? 0 : fdiv(a.by_L, s + L - 1);
        lo = max(lo, done + 1);
        const int hi = e < 0 ? -1 : min(fdiv(a.by_L, e), J);
        if (hi >= lo) { covered += hi - lo + 1; done = hi; }
    }
    const int nF = parts + 1 + tail - covered;
    int nf = 1;                                   // chop.hpp:250-276
    if (nF > a.div + 1) nf = fdiv(a.by_div, nF - 1 + a.div - 1);
    a.cut_cnt[r] = nF;
    a.frag_cnt[r] = nf;
    n_out = n; nF_out = nF; nf_out = nf;
}

// finalize_count_one as the loader of the output scan's first pass (device_scan.hpp exclusive_scan2): that pass walks the
// reads one per thread, coalesced, exactly as finalize_count_kernel does, so the count rides in it for free (one launch
// less; the second pass re-reads the three counts).
struct FinalizeCountLoader {
    FinalizeArgs a;
    __device__ void operator()(long long i, long long (&v)[3]) const
    {
        v[0] = v[1] = v[2] = 0;
        if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
        int n, nF, nf;
        long long bp = 0;
        finalize_count_one(a, (int)i, n, nF, nf, bp);
        if (bp) atomicAdd(a.total_repeat, (unsigned long long)bp);
        v[0] = n; v[1] = nF; v[2] = nf;
    }
};

// (Measured and dropped, round 3: a long read's raw runs ordered and joined by its whole wave -- a lane per run, rank by 64
// compare steps, joined through ballots -- instead of its thread's loops in global memory.  The long reads of a wave then
// take their turns, where the threads' loops had run side by side: finalize_count 0.11 -> 0.21 ms on the ultralong set.)
__global__ __launch_bounds__(256) void finalize_count_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
    long long bp = 0;
    if (r < a.n_reads) {
        int n, nF, nf;
        finalize_count_one(a, r, n, nF, nf, bp);
    }
    // one atomic per wave that has anything to add
    if (__ballot(bp != 0) != 0ull) {
        const long long s = wave_reduce_add64(bp);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.total_repeat, (unsigned long long)s);
    }
}

// Compact repeats and the fragments of every read.  The cut points themselves (chop.hpp's final_stars, 4 B per
// marker: 0.4 GB on the human-scale set) are neither stored nor walked here: fragment j begins at the kept marker
// with index (j-1)*div and ends at the one with index j*div, the first marker is 0 and the last is the read length.
// finalize_cuts_kernel materialises them when a caller asks for them.
__device__ __forceinline__ void finalize_fill_one(const FinalizeArgs &a, int r, long long ro, long long fo, int n, int nF)
{
    const long long base = a.rep_res_off[r];
    for (int i = 0; i < n; ++i) { a.rep_s[ro + i] = a.raw_s[base + i]; a.rep_e[ro + i] = a.raw_e[base + i]; }
    const int len = a.read_len[r];
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
            const int s = a.raw_s[base + k], e = a.raw_e[base + k];
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
template <bool CUTS>
__global__ __launch_bounds__(256) void finalize_fill_kernel(FinalizeArgs a)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_reads) return;
    if (*(volatile int32_t *)a.err_flags & (kErrStop | kErrLen)) return;   // (sizes or lengths are not what the pass was built on)
    const int n = a.rep_cnt[r];
    finalize_fill_one(a, r, a.rep_off[r], a.frag_off[r], n, a.cut_cnt[r]);
    if (CUTS) {
        int32_t *F = a.cuts + a.cut_off[r];
        int w = 0;
        (void)walk_cuts(a.read_len[r], a.interval_length, a.raw_s + a.rep_res_off[r], a.raw_e + a.rep_res_off[r], n, [&](int m) { F[w++] = m; });
    }
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
    (void)walk_cuts(a.read_len[r], a.interval_length, a.raw_s + a.rep_res_off[r], a.raw_e + a.rep_res_off[r], a.rep_cnt[r],
                    [&](int m) { F[w++] = m; });
}

// totals[0] = sum coverage, [1] = sum unclamped repeat bp, [2] = sum read length.  One atomic per workgroup and
// total (a few hundred per launch): with one per wave the three counters saw 12 k serialised atomics and the kernel
// took 56 us for 13 MB of input.
constexpr int kSeqWord = 256;     // (in 8-byte words behind the control block's copy in the page-locked block)
__global__ __launch_bounds__(256) void totals_kernel(long long n_tiles, const long long *tile_sums, int32_t n_reads,
                                                     const int32_t *read_len, unsigned long long *totals,
                                                     const long long *rep_off, const long long *cut_off,
                                                     const long long *frag_off, const long long *bucket_off, long long *tails,
                                                     unsigned *done_blocks, const long long *ctrl_words, int n_ctrl_words,
                                                     long long *host_block, long long pass_seq)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {    // output sizes, so that the host reads one block back
        tails[0] = rep_off[n_reads]; tails[1] = cut_off[n_reads]; tails[2] = frag_off[n_reads];
        tails[3] = bucket_off ? bucket_off[n_reads] : 0;
    }
    __shared__ long long part[3][4];
    long long c = 0, rp = 0, l = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = i0; i < n_tiles; i += stride) { c += tile_sums[2 * i]; rp += tile_sums[2 * i + 1]; }
    for (long long i = i0; i < n_reads; i += stride) l += read_len[i];
    c = wave_reduce_add64(c); rp = wave_reduce_add64(rp); l = wave_reduce_add64(l);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wid] = c; part[1][wid] = rp; part[2][wid] = l; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const long long v = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        if (v) atomicAdd(&totals[threadIdx.x], (unsigned long long)v);
    }
    // The workgroup that finishes last copies the control block -- everything raft_hip_finish reports -- into the context's
    // page-locked block (a one-wave kernel of its own before: one launch less at the end of every pass).
    __shared__ int last;
    __syncthreads();                                     // (this workgroup's three atomics are issued)
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(done_blocks, 1u) == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (last && (int)threadIdx.x < 64) {
        __threadfence();
        if ((int)threadIdx.x < n_ctrl_words) host_block[threadIdx.x] = reinterpret_cast<const volatile long long *>(ctrl_words)[threadIdx.x];
        __threadfence_system();
        // ... and then the pass's number, behind the block: raft_hip_finish spins on it instead of sleeping in the runtime's wait
        // (whose wake-up is 20-30 us of a pass that may take 200)
        if (threadIdx.x == 0) { host_block[kSeqWord] = pass_seq; __threadfence_system(); }
    }
}

} // namespace raft
