// bucket.hpp -- everything that turns the PAF record stream into per-read
// interval ranges for the pileup kernel.
//
// Reference: create_pileup (chop.hpp:133-191) heap-allocates one Overlap per
// record and pushes its pointer into the query's bucket and, while the symmetric
// flag is 0 and query != target, into the target's bucket; profileCoverage
// (repeat.hpp:48-58) then keeps the query side of every record of a read, and the
// target side only when the FINAL flag is 0.  Equivalent multiset of intervals:
//     symmetric == 1 : { (qid, qs, qe) }                         I = n_rec
//     symmetric == 0 : { (qid, qs, qe) } + { (tid, ts, te) : qid != tid }
//
// Two ways to hand those to the pileup kernel:
//   * sorted-segment fast path: hifiasm writes its PAF grouped by ascending query
//     index and the workflow concatenates two such files (README.md:36-38), so the
//     record stream is a handful of sorted runs.  In symmetric mode the original
//     columns (qid,qs,qe) ARE the bucketed intervals: only the per-tile start
//     positions are needed (one binary search per tile and run).  No copy.
//   * bucketing path: counting sort by read id -- histogram, exclusive scan,
//     scatter -- with per-wave run aggregation so that grouped input costs one
//     atomic per run, not per record.
#pragma once
#include "pileup.hpp"

namespace raft {

struct InspectOut {            // device words written by inspect_kernel
    int32_t sym_found;         // a record i >= 1 mirrors record 0 (chop.hpp:175-184)
    int32_t n_desc;            // positions i with qid[i] < qid[i-1]
    int32_t err_flags;
    int32_t pad;
    long long err_index;
    long long desc_pos[kMaxSeg]; // first kMaxSeg descent positions (unordered)
};

// One pass over qid (and over the other columns only where qid matches tid[0]): 16-byte loads, four records
// per lane, so that the pass streams at HBM rate (it reads 4 B per record of the 12-24 B the pileup reads).
// (Measured and dropped: writing a table "first record of every read in every run" from this pass, to turn the tile
// cuts into look-ups -- the extra stores and index arithmetic took the pass from 220 to 310-440 us, more than the
// searches in tile_desc_kernel cost.)
__device__ __forceinline__ void inspect_one(long long i, int32_t q, int32_t prev, int32_t n_reads, int detect_sym,
                                            const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts,
                                            const int32_t *te, int32_t q0, int32_t t0, int32_t qs0, int32_t qe0,
                                            int32_t ts0, int32_t te0, InspectOut *out, bool &many)
{
    if (q < 0 || q >= n_reads) {
        atomicOr(&out->err_flags, kErrReadId);
        atomicMin((unsigned long long *)&out->err_index, (unsigned long long)i);
    }
    if (i > 0) {
        // (`many`: this thread has seen the count beyond what anybody asks for -- a shuffled stream has a descent at every other
        // record, and 1.5e8 returning atomics on the one word were 50 ms of a pass)
        if (q < prev && !many) {
            const int slot = atomicAdd(&out->n_desc, 1);
            if (slot < kMaxSeg) out->desc_pos[slot] = i;
            else many = true;
        }
        if (detect_sym && q == t0) {
            if (tid[i] == q0 && ts[i] == qs0 && te[i] == qe0 && qs[i] == ts0 && qe[i] == te0) out->sym_found = 1;
        }
    }
}

__global__ __launch_bounds__(256) void inspect_kernel(long long n_rec, int32_t n_reads, int detect_sym,
                                                      const int32_t *qid, const int32_t *qs, const int32_t *qe,
                                                      const int32_t *tid, const int32_t *ts, const int32_t *te,
                                                      InspectOut *out)
{
    const int32_t q0 = qid[0], t0 = tid[0], qs0 = qs[0], qe0 = qe[0], ts0 = ts[0], te0 = te[0];
    // records [0, head) bring qid to a 16-byte boundary, then groups of four, then a tail
    long long head = (long long)(((16u - (unsigned)(reinterpret_cast<unsigned long long>(qid) & 15u)) & 15u) >> 2);
    if (head > n_rec) head = n_rec;
    const long long n_groups = (n_rec - head) >> 2;
    const long long tail = head + (n_groups << 2);
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool many = *(volatile int32_t *)&out->n_desc > kMaxSeg;
    for (long long g = t0i; g < n_groups; g += stride) {
        const long long i = head + (g << 2);
        const int4 v = *reinterpret_cast<const int4 *>(qid + i);
        const int32_t prev = i > 0 ? qid[i - 1] : 0;
        inspect_one(i + 0, v.x, prev, n_reads, detect_sym, qs, qe, tid, ts, te, q0, t0, qs0, qe0, ts0, te0, out, many);
        inspect_one(i + 1, v.y, v.x, n_reads, detect_sym, qs, qe, tid, ts, te, q0, t0, qs0, qe0, ts0, te0, out, many);
        inspect_one(i + 2, v.z, v.y, n_reads, detect_sym, qs, qe, tid, ts, te, q0, t0, qs0, qe0, ts0, te0, out, many);
        inspect_one(i + 3, v.w, v.z, n_reads, detect_sym, qs, qe, tid, ts, te, q0, t0, qs0, qe0, ts0, te0, out, many);
    }
    if (blockIdx.x == 0) {
        for (long long i = threadIdx.x; i < head + (n_rec - tail); i += blockDim.x) {
            const long long j = i < head ? i : tail + (i - head);
            inspect_one(j, qid[j], j > 0 ? qid[j - 1] : 0, n_reads, detect_sym, qs, qe, tid, ts, te, q0, t0, qs0, qe0, ts0, te0, out, many);
        }
    }
}

// Sorted runs of the record stream as up to 16 k evenly spaced samples show them: where one sample is smaller than the one
// before, a run ends in between, and a bisection (left part >= the earlier sample, right part below it) finds the first
// record of the next run.  inspect_kernel, which looks at every record, confirms or refutes it.
// The samples themselves are kept (64 KB): they are a coarse index of the stream.  tile_desc_kernel bisects them first
// -- cache hits -- and then only the 9 k records between two samples, instead of 28 dependent probes spread over a
// GB-sized column (every one of them a TLB miss).
struct GuessOut {
    int32_t n_desc, pad;
    long long desc_pos[kMaxSeg];
};


__global__ __launch_bounds__(256) void guess_runs_kernel(long long n_rec, const int32_t *qid, GuessOut *out, int32_t *samples)
{
    const int sh = sample_shift(n_rec);                   // (out->n_desc was zeroed with the control block)
    const long long S = n_samples(n_rec, sh);
    const long long i = 1 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 1 && samples) samples[0] = qid[0];
    if (i >= S) return;
    const long long p0 = sample_pos(i - 1, n_rec, sh), p1 = sample_pos(i, n_rec, sh);
    const int32_t v = qid[p0], w = qid[p1];
    if (samples) samples[i] = w;
    if (w < v) {
        long long lo = p0, hi = p1;
        while (hi - lo > 1) {
            const long long mid = lo + (hi - lo) / 2;
            if (qid[mid] >= v) lo = mid; else hi = mid;
        }
        const int slot = atomicAdd(&out->n_desc, 1);
        if (slot < kMaxSeg) out->desc_pos[slot] = hi;
    }
}

// tile_first[k] = first read whose first window lies in tile k or later (tiles of Q windows).  For grouped input the same
// threads -- one per read, coalesced -- check the caller's offsets: they must not step back and the runs must chain from
// record 0 to record n_rec (kErrGroup; every later kernel of the pass then returns at once).
// It also clears the reads' repeat counters (a fill command of its own before) and, in a pass that was sized by the caller's
// window count, compares that count with the scan's (kErrHint; a fill command and a one-wave kernel of their own cost
// ~15 us of a 0.5 ms pass on an eighth of the human-scale set).
__global__ __launch_bounds__(256) void tile_first_kernel(int32_t n_reads, const long long *cov_off, int Q,
                                                         long long n_tiles, int32_t *tile_first, int32_t *err_flags,
                                                         long long *err_index, GroupedOff grp, int32_t n_runs, long long n_rec,
                                                         int32_t *rep_cnt, const long long *scan_totals, long long hint_bins)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    if (hint_bins >= 0 && scan_totals[0] != hint_bins) {           // (every thread sees the same: nothing below runs)
        if (r == 0) atomicOr(err_flags, kErrHint);
        return;
    }
    if (r < n_reads) rep_cnt[r] = 0;
    if (grp.off) {
        bool bad = false;
        if (r < n_reads)
            for (int s = 0; s < n_runs; ++s) bad |= grp.at(s, r) > grp.at(s, r + 1);
        if (r == 0) {
            long long at = 0;
            for (int s = 0; s < n_runs; ++s) { bad |= grp.at(s, 0) != at; at = grp.at(s, n_reads); }
            bad |= at != n_rec;
        }
        if (bad) {
            atomicOr(err_flags, kErrGroup);
            atomicMin((unsigned long long *)err_index, (unsigned long long)r);
        }
    }
    const long long t_r = (r < n_reads) ? cov_off[r] / Q : n_tiles;
    const long long t_p = (r > 0) ? cov_off[r - 1] / Q : -1;
    for (long long k = t_p + 1; k <= t_r; ++k) tile_first[k] = (int32_t)r;
}

// ---- grouped input of more runs than the pileup kernels take (kMaxSeg): merged into ONE run first ----------------------
// (a PAF concatenated from many files; the intervals a rank of a pre-split job receives from its peers, two runs each).
// With the offsets at hand this needs no histogram and no atomics: read r's records of run j go behind its records of the
// runs before, at sum_j' (off_j'[r] - off_j'[0]) + sum_{j' < j} count_j'(r).  24 bytes of traffic per record.
constexpr int kMaxRuns = 16;

__global__ __launch_bounds__(256) void check_offsets_kernel(int32_t n_reads, int32_t n_runs, const long long *off, long long stride,
                                                            long long n_rec, int32_t *err_flags, long long *err_index)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;
    bool bad = false;
    if (r < n_reads)
        for (int s = 0; s < n_runs; ++s) bad |= off[s * stride + r] > off[s * stride + r + 1];
    if (r == 0) {
        long long at = 0;
        for (int s = 0; s < n_runs; ++s) { bad |= off[s * stride] != at; at = off[s * stride + n_reads]; }
        bad |= at != n_rec;
    }
    if (bad) {
        atomicOr(err_flags, kErrGroup);
        atomicMin((unsigned long long *)err_index, (unsigned long long)r);
    }
}

__global__ __launch_bounds__(256) void merge_runs_kernel(int32_t n_reads, int32_t n_runs, const long long *off, long long stride,
                                                         const int32_t *qs, const int32_t *qe, long long *m_off, int32_t *b_rid,
                                                         int32_t *b_s, int32_t *b_e, const int32_t *err_flags)
{
    if (*(volatile const int32_t *)err_flags & kErrStop) return;   // (offsets that step back or leave [0, n_rec]: check_offsets_kernel)
    const int lane = threadIdx.x & 63;
    const long long n_groups = ((long long)n_reads + 1 + 63) >> 6;          // (entry n_reads closes the merged run)
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave0; w < n_groups; w += n_waves) {
        const long long r0 = w << 6, r = r0 + lane;
        long long lo[kMaxRuns];
        int cnt[kMaxRuns];
        long long mo = 0;
        int tot = 0;
#pragma unroll
        for (int j = 0; j < kMaxRuns; ++j) {
            lo[j] = 0; cnt[j] = 0;
            if (j < n_runs && r <= n_reads) {
                const long long a = off[j * stride + r];
                mo += a - off[j * stride];
                if (r < n_reads) { lo[j] = a; cnt[j] = (int)(off[j * stride + r + 1] - a); tot += cnt[j]; }
            }
        }
        if (r <= n_reads) m_off[r] = mo;
        const int n_in = (int)min(64LL, (long long)n_reads - r0);
        for (int l = 0; l < n_in; ++l) {
            const int T = __builtin_amdgcn_readlane(tot, l);
            if (T == 0) continue;
            const long long base = ((long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)mo, l)) |
                                   ((long long)__builtin_amdgcn_readlane((int)((unsigned long long)mo >> 32), l) << 32);
            for (int i0 = 0; i0 < T; i0 += 64) {
                const int i = i0 + lane;
                int k = i < T ? i : -1;                       // index inside the read; negative once placed (or idle)
                long long src = -1;
#pragma unroll
                for (int j = 0; j < kMaxRuns; ++j) {
                    const int c = __builtin_amdgcn_readlane(cnt[j], l);
                    const long long s = ((long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned long long)lo[j], l)) |
                                        ((long long)__builtin_amdgcn_readlane((int)((unsigned long long)lo[j] >> 32), l) << 32);
                    if (k >= 0 && k < c) { src = s + k; k = -1; }
                    else if (k >= 0) k -= c;
                }
                if (src >= 0) {
                    b_rid[base + i] = (int32_t)(r0 + l);
                    b_s[base + i] = qs[src];
                    b_e[base + i] = qe[src];
                }
            }
        }
    }
}

// Grouped input without a query column: the ids are what the offsets say.  One wave per 64 consecutive reads and run: the
// lanes hold their reads' ranges, and the wave writes each read's id over its range (a read has ~45 records in a run:
// one store instruction per read, two for a read inside a repeat).
__global__ __launch_bounds__(256) void expand_ids_kernel(int32_t n_reads, int32_t n_runs, GroupedOff grp, int32_t *qid,
                                                         const int32_t *err_flags)
{
    if (*(volatile const int32_t *)err_flags & kErrStop) return;   // (offsets that step back or leave [0, n_rec])
    const int lane = threadIdx.x & 63;
    const long long n_groups = ((long long)n_reads + 63) >> 6;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave0; w < n_groups * n_runs; w += n_waves) {
        const int s = (int)(w / n_groups);
        const long long r0 = (w - (long long)s * n_groups) << 6;
        const long long r = r0 + lane;
        const long long lo = r < n_reads ? grp.at(s, r) : 0, hi = r < n_reads ? grp.at(s, r + 1) : 0;
        const int n_in = (int)min(64LL, (long long)n_reads - r0);
        for (int l = 0; l < n_in; ++l) {
            const long long a = __shfl(lo, l, kWave), b = __shfl(hi, l, kWave);
            for (long long i = a + lane; i < b; i += kWave) qid[i] = (int32_t)(r0 + l);
        }
    }
}

// Window records (pileup_fast.hpp IN = 1) for the passes that need coordinate columns -- the general pileup kernel, the merge
// of more than kMaxSeg runs: coordinates that fall into the same windows (first * reso, last1 * reso; an empty record
// becomes (0, 0)).  reso <= 32767 keeps 65535 * reso inside int32 (checked by the engine).
__global__ __launch_bounds__(256) void unpack_windows_kernel(long long n, const uint32_t *__restrict__ w, int32_t reso, int32_t *__restrict__ qs,
                                                             int32_t *__restrict__ qe)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t v = w[i];
        qs[i] = (int32_t)(v & 0xffffu) * reso;
        qe[i] = (int32_t)(v >> 16) * reso;
    }
}

// ---- counting sort by read id ------------------------------------------------

// Lanes that hold the same key as the previous lane are folded into the run's
// head lane: one atomic per run.  Returns, for every lane, the head lane of its
// run and (in *run_len) the length of the run when called on the head.
__device__ __forceinline__ int run_head(bool valid, int key, int lane, int *run_len, bool *is_head)
{
    const int prev_key = __shfl_up(key, 1, kWave);
    const bool prev_valid = __shfl_up((int)valid, 1, kWave) != 0;
    const bool head = valid && (lane == 0 || !prev_valid || prev_key != key);
    const unsigned long long hm = __ballot(head);
    const unsigned long long vm = __ballot(valid);
    // head lane of my run: highest head bit at or below my lane
    const unsigned long long below = hm & ((2ull << lane) - 1ull);
    const int h = below ? top_bit(below) : lane;
    // run length for a head: distance to the next head above it, or to the end of the valid lanes
    const unsigned long long above = (lane < 63) ? (hm >> (lane + 1)) : 0ull;
    const int nvalid = __popcll(vm);              // valid lanes are a prefix of the wave by construction
    const int next = above ? lane + 1 + (int)__builtin_ctzll(above) : nvalid;
    *run_len = next - lane;
    *is_head = head;
    return h;
}

__global__ __launch_bounds__(256) void bucket_hist_kernel(long long n_rec, int32_t n_reads, int symmetric,
                                                          const int32_t *qid, const int32_t *tid,
                                                          int32_t *cnt, int32_t *err_flags, long long *err_index)
{
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_round = (n_rec + 63) & ~63LL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool valid = i < n_rec;
        int q = valid ? qid[i] : -1;
        bool okq = valid && q >= 0 && q < n_reads;
        if (valid && !okq) {                         // (inspect_kernel reports the same when it runs; a verified pass has no inspect)
            atomicOr(err_flags, kErrReadId);
            atomicMin((unsigned long long *)err_index, (unsigned long long)i);
        }
        int len; bool head;
        run_head(okq, q, lane, &len, &head);
        // the fold assumes valid lanes form a prefix; a bad id in the middle breaks that, so fall back per lane
        const unsigned long long vm = __ballot(okq);
        const bool prefix = (vm & (vm + 1ull)) == 0ull;
        if (prefix) { if (head) atomicAdd(&cnt[q], len); }
        else if (okq) atomicAdd(&cnt[q], 1);
        if (!symmetric && valid) {
            const int t = tid[i];
            if (t < 0 || t >= n_reads) {
                atomicOr(err_flags, kErrReadId);
                atomicMin((unsigned long long *)err_index, (unsigned long long)i);
            } else if (t != q) atomicAdd(&cnt[t], 1);
        }
    }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(long long n_rec, int32_t n_reads, int symmetric,
                                                             const int32_t *qid, const int32_t *qs, const int32_t *qe,
                                                             const int32_t *tid, const int32_t *ts, const int32_t *te,
                                                             const long long *iv_off, int32_t *cursor,
                                                             int32_t *b_rid, int32_t *b_s, int32_t *b_e)
{
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n_round = (n_rec + 63) & ~63LL;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool valid = i < n_rec;
        int q = valid ? qid[i] : -1;
        const bool okq = valid && q >= 0 && q < n_reads;
        int len; bool head;
        const int h = run_head(okq, q, lane, &len, &head);
        const unsigned long long vm = __ballot(okq);
        const bool prefix = (vm & (vm + 1ull)) == 0ull;
        int slot = 0;
        if (prefix) {
            int base = 0;
            if (head) base = atomicAdd(&cursor[q], len);
            base = __shfl(base, h, kWave);
            slot = base + (lane - h);
        } else if (okq) slot = atomicAdd(&cursor[q], 1);
        if (okq) {
            const long long d = iv_off[q] + slot;
            b_rid[d] = q; b_s[d] = qs[i]; b_e[d] = qe[i];
        }
        if (!symmetric && valid) {
            const int t = tid[i];
            if (t >= 0 && t < n_reads && t != q) {
                const long long d = iv_off[t] + atomicAdd(&cursor[t], 1);
                b_rid[d] = t; b_s[d] = ts[i]; b_e[d] = te[i];
            }
        }
    }
}

// ---- bucketing by read id for streams of any order: partition twice, never sort (round 5) --------------------------------------
// The counting sort above -- one atomic per record on its read's counter, then a 12-byte write per record wherever its read's
// bucket lies -- is fine where the stream is almost sorted and took 87 ms for 2.9e8 SHUFFLED records (every write a partial
// line somewhere in 3.5 GB).  Round 4 sorted instead (a library radix sort of (id, start | end << 32) pairs: three passes over
// 12-byte pairs plus histograms, an expansion before and an unzip behind it: 9.6 ms).  But create_pileup (chop.hpp:147-187) needs
// no ORDER: profileCoverage adds up a read's intervals in whatever order they come (repeat.hpp:48-79).  What is needed is that a
// read's intervals lie together and the reads in index order -- a partition, and one that can be made in two steps, each of which
// writes into few places at a time:
//   (A) coarse_partition_kernel: every side goes to the COARSE bucket of its read, 2^rshift consecutive reads each, chosen so that
//       a bucket's intervals are ~2 MB -- a few thousand buckets.  A workgroup ranks its 8 k sides by bucket in LDS (a returning
//       ds_add per side), reserves room in every bucket it has sides for with ONE global atomic per bucket, and writes; the
//       buckets' open ends are a few thousand lines that stay in L2 until they are full.
//   (B) bucket_finish_kernel: one workgroup per coarse bucket.  The bucket's reads are few enough for an LDS counter each: count,
//       scan (the reads' offsets: pileup input), and scatter -- random 12-byte writes, but inside 2 MB that sit in the XCD's L2.
// Bytes: (A0) the ids once for the buckets' sizes; (A) 12-24 B read + 12 B written per side; (B) 12 B read twice (the second time
// from L2) + 12 B written.  Sides that do not exist (a bad id: reported; the target side of a symmetric PAF or of a self overlap,
// chop.hpp:165-169) are dropped in (A).
constexpr int kCoarseMax = 4096;              // coarse buckets (LDS counters of (A), 16 KB)
constexpr int kFineMaxShift = 13;             // reads per coarse bucket at most 2^13 (LDS counters of (B), 32 KB)
constexpr int kPartThreads = 1024;

// (A0) sides per coarse bucket.  cnt[k] zeroed by the caller.
__global__ __launch_bounds__(256) void coarse_hist_kernel(long long n_rec, int32_t n_reads, int symmetric, int rshift, int n_coarse,
                                                          const int32_t *__restrict__ qid, const int32_t *__restrict__ tid,
                                                          unsigned long long *__restrict__ cnt, int32_t *err_flags, long long *err_index)
{
    __shared__ int32_t h[kCoarseMax];
    for (int i = threadIdx.x; i < n_coarse; i += 256) h[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (long long)gridDim.x * blockDim.x) {
        const int q = qid[i];
        const bool okq = q >= 0 && q < n_reads;
        bool bad = !okq;
        if (okq) atomicAdd(&h[q >> rshift], 1);
        if (!symmetric) {
            const int t = tid[i];
            const bool okt = t >= 0 && t < n_reads;
            bad = bad || !okt;
            if (okt && t != q) atomicAdd(&h[t >> rshift], 1);
        }
        if (bad) {
            atomicOr(err_flags, kErrReadId);
            atomicMin((unsigned long long *)err_index, (unsigned long long)i);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_coarse; i += 256)
        if (h[i]) atomicAdd(&cnt[i], (unsigned long long)h[i]);
}

// where every coarse bucket begins (base[0 .. n_coarse]) and the cursors of (A); one workgroup
__global__ __launch_bounds__(1024) void coarse_scan_kernel(int n_coarse, const unsigned long long *__restrict__ cnt, long long *__restrict__ base,
                                                           unsigned long long *__restrict__ cursor)
{
    __shared__ long long part[1024];
    const int per = (n_coarse + 1023) / 1024;
    const int lo = min((int)threadIdx.x * per, n_coarse), hi = min(lo + per, n_coarse);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += (long long)cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; }
        base[n_coarse] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { base[i] = run; cursor[i] = (unsigned long long)run; run += (long long)cnt[i]; }
}

// (A) the sides, bucket by bucket.  A workgroup takes TILES of kPartTile records: it counts the tile's sides per bucket (the id
// columns only), reserves the tile's room in every bucket with one global atomic per bucket -- a few million for 3e8 records; a
// tile of 4 k records would make one per SIDE of them -- and walks the tile again (its ids come from L2 now), ranking every side
// within its bucket by a returning ds_add.  Loads are issued kPartUnroll records ahead per thread: the kernel lives on the
// memory system's parallelism, not on its threads' arithmetic.
constexpr int kPartTile = 1 << 17;          // records per tile at most (the host shrinks it for small inputs: 512 tiles at the least)
constexpr int kPartUnroll = 4;
__global__ __launch_bounds__(kPartThreads) void coarse_partition_kernel(long long n_rec, int tile, int32_t n_reads, int symmetric, int rshift, int n_coarse,
                                                                        const int32_t *__restrict__ qid, const int32_t *__restrict__ qs, const int32_t *__restrict__ qe,
                                                                        const int32_t *__restrict__ tid, const int32_t *__restrict__ ts, const int32_t *__restrict__ te,
                                                                        unsigned long long *__restrict__ cursor,
                                                                        int32_t *__restrict__ o_rid, int32_t *__restrict__ o_s, int32_t *__restrict__ o_e)
{
    __shared__ int32_t h[kCoarseMax];           // sides of this tile per bucket; then the rank counters of the second walk
    __shared__ long long first[kCoarseMax];     // where this tile's sides of the bucket go
    const long long n_tiles = (n_rec + tile - 1) / tile;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long long t0 = t * tile, t1 = min(t0 + (long long)tile, n_rec);
        for (int i = threadIdx.x; i < n_coarse; i += kPartThreads) h[i] = 0;
        __syncthreads();
        for (long long i0 = t0 + threadIdx.x; i0 < t1; i0 += (long long)kPartThreads * kPartUnroll) {
            int q[kPartUnroll], tt[kPartUnroll];
#pragma unroll
            for (int k = 0; k < kPartUnroll; ++k) {
                const long long i = i0 + (long long)k * kPartThreads;
                q[k] = i < t1 ? qid[i] : -1;
                tt[k] = (!symmetric && i < t1) ? tid[i] : -1;
            }
#pragma unroll
            for (int k = 0; k < kPartUnroll; ++k) {
                if (q[k] >= 0 && q[k] < n_reads) atomicAdd(&h[q[k] >> rshift], 1);
                if (tt[k] >= 0 && tt[k] < n_reads && tt[k] != q[k]) atomicAdd(&h[tt[k] >> rshift], 1);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n_coarse; i += kPartThreads) {
            const int c = h[i];
            if (c) first[i] = (long long)atomicAdd(&cursor[i], (unsigned long long)c);
            h[i] = 0;
        }
        __syncthreads();
        for (long long i0 = t0 + threadIdx.x; i0 < t1; i0 += (long long)kPartThreads * kPartUnroll) {
            int q[kPartUnroll], a[kPartUnroll], b[kPartUnroll], tt[kPartUnroll], ta[kPartUnroll], tb[kPartUnroll];
#pragma unroll
            for (int k = 0; k < kPartUnroll; ++k) {
                const long long i = i0 + (long long)k * kPartThreads;
                q[k] = -1; tt[k] = -1; a[k] = b[k] = ta[k] = tb[k] = 0;
                if (i < t1) {
                    q[k] = qid[i]; a[k] = qs[i]; b[k] = qe[i];
                    if (!symmetric) { tt[k] = tid[i]; ta[k] = ts[i]; tb[k] = te[i]; }
                }
            }
#pragma unroll
            for (int k = 0; k < kPartUnroll; ++k) {
                if (q[k] >= 0 && q[k] < n_reads) {
                    const int bk = q[k] >> rshift;
                    const long long d = first[bk] + atomicAdd(&h[bk], 1);
                    o_rid[d] = q[k]; o_s[d] = a[k]; o_e[d] = b[k];
                }
                if (tt[k] >= 0 && tt[k] < n_reads && tt[k] != q[k]) {
                    const int bk = tt[k] >> rshift;
                    const long long d = first[bk] + atomicAdd(&h[bk], 1);
                    o_rid[d] = tt[k]; o_s[d] = ta[k]; o_e[d] = tb[k];
                }
            }
        }
        __syncthreads();
    }
}

// (B) one workgroup per coarse bucket: its reads' counts, their offsets, and every interval to its read's place.
// off[r] for the bucket's reads (and off[n_reads] by the last bucket): where read r's intervals begin.
__global__ __launch_bounds__(kPartThreads) void bucket_finish_kernel(int32_t n_reads, int rshift, int n_coarse, const long long *__restrict__ base,
                                                                     const int32_t *__restrict__ i_rid, const int32_t *__restrict__ i_s, const int32_t *__restrict__ i_e,
                                                                     int32_t *__restrict__ o_rid, int32_t *__restrict__ o_s, int32_t *__restrict__ o_e,
                                                                     long long *__restrict__ off)
{
    __shared__ int32_t cnt[1 << kFineMaxShift];
    __shared__ int32_t wsum[kPartThreads / 64];
    const int R = 1 << rshift;
    const int per = max(1, R / kPartThreads);           // consecutive counters per thread in the scan (R <= 8 k: at most 8)
    constexpr int UN = 8;
    for (int b = blockIdx.x; b < n_coarse; b += gridDim.x) {
        const long long lo = base[b], hi = base[b + 1];
        const int r0 = b << rshift;
        const int nr = min(R, n_reads - r0);
        for (int i = threadIdx.x; i < R; i += kPartThreads) cnt[i] = 0;
        __syncthreads();
        for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += (long long)kPartThreads * UN) {
            int r[UN];
#pragma unroll
            for (int k = 0; k < UN; ++k) { const long long i = i0 + (long long)k * kPartThreads; r[k] = i < hi ? i_rid[i] : -1; }
#pragma unroll
            for (int k = 0; k < UN; ++k) if (r[k] >= 0) atomicAdd(&cnt[r[k] - r0], 1);
        }
        __syncthreads();
        // exclusive scan of cnt[0 .. R): thread t owns counters [t * per, t * per + per)
        int mine = 0;
        const int c0 = (int)threadIdx.x * per;
        if (c0 < R) for (int k = 0; k < per; ++k) mine += cnt[c0 + k];
        const int incl = wave_incl_scan_add(mine);
        const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        if (threadIdx.x < kPartThreads / 64) {            // (16 wave sums: one partial wave scans them)
            const int v = wsum[threadIdx.x];
            int acc = v;
            for (int d = 1; d < kPartThreads / 64; d <<= 1) { const int o = __shfl_up(acc, d, kWave); if ((int)threadIdx.x >= d) acc += o; }
            wsum[threadIdx.x] = acc - v;
        }
        __syncthreads();
        int run = wsum[wid] + incl - mine;
        if (c0 < R) for (int k = 0; k < per; ++k) {
            const int v = cnt[c0 + k];
            cnt[c0 + k] = run;                             // the read's cursor from here on
            if (c0 + k < nr) off[r0 + c0 + k] = lo + run;
            run += v;
        }
        if (b == n_coarse - 1 && threadIdx.x == 0) off[n_reads] = hi;
        __syncthreads();
        constexpr int U2 = 4;
        for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += (long long)kPartThreads * U2) {
            int r[U2], a[U2], e[U2];
#pragma unroll
            for (int k = 0; k < U2; ++k) {
                const long long i = i0 + (long long)k * kPartThreads;
                r[k] = -1; a[k] = 0; e[k] = 0;
                if (i < hi) { r[k] = i_rid[i]; a[k] = i_s[i]; e[k] = i_e[i]; }
            }
#pragma unroll
            for (int k = 0; k < U2; ++k) {
                if (r[k] < 0) continue;
                const long long d = lo + atomicAdd(&cnt[r[k] - r0], 1);
                o_rid[d] = r[k]; o_s[d] = a[k]; o_e[d] = e[k];
            }
        }
        __syncthreads();
    }
}

} // namespace raft
