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

// Sorted runs of the record stream as up to 256 k evenly spaced samples show them: where one sample is smaller than the one
// before, a run ends in between, and a bisection (left part >= the earlier sample, right part below it) finds the first
// record of the next run.  inspect_kernel, which looks at every record, confirms or refutes it.
// The samples themselves are kept (1 MB): they are a coarse index of the stream.  tile_desc_kernel bisects them first
// -- cache hits -- and then only the ~1 k records between two samples, instead of 28 dependent probes spread over a
// GB-sized column (every one of them a TLB miss).
// (struct GuessOut: pileup.hpp, beside tile_desc_kernel, which checks a speculative pass against it)


__device__ __forceinline__ void guess_runs_body(long long block, long long n_rec, const int32_t *qid, GuessOut *out, int32_t *samples)
{
    const int sh = sample_shift(n_rec);                   // (out->n_desc was zeroed with the control block)
    const long long S = n_samples(n_rec, sh);
    const long long i = 1 + block * blockDim.x + threadIdx.x;
    if (i == 1 && samples) samples[0] = qid[0];
    if (i >= S) return;
    const long long p0 = sample_pos(i - 1, n_rec, sh), p1 = sample_pos(i, n_rec, sh);
    const int32_t v = qid[p0], w = qid[p1];
    if (samples) samples[i] = w;
    // (a stream that is not a handful of sorted runs: a wave that alone sees more descents than the pass accepts says so with ONE
    // atomic and looks no further -- with one per descent, the 128 k of a shuffled stream's 256 k samples would queue up on the word)
    const unsigned long long dm = __ballot(w < v);
    if (__popcll(dm) > kMaxSeg) {
        if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(dm)) atomicAdd(&out->n_desc, (int)__popcll(dm));
        return;
    }
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

__global__ __launch_bounds__(256) void guess_runs_kernel(long long n_rec, const int32_t *qid, GuessOut *out, int32_t *samples)
{
    guess_runs_body((long long)blockIdx.x, n_rec, qid, out, samples);
}
// ... riding in the geometry scan's first launch (device_scan.hpp Beside): the workgroups behind the scan's own
struct GuessBeside {
    long long n_rec;
    const int32_t *qid;
    GuessOut *out;
    int32_t *samples;
    __device__ void operator()(int block) const { guess_runs_body(block, n_rec, qid, out, samples); }
};

// A speculative pass over reads whose geometry the context still holds (engine_ctx.hpp geom_id): the lengths against the copy the scan
// that made it kept; any difference refutes the pass (kErrHint: raft_hip_finish runs it again the long way, which scans).  The repeat
// counters are cleared on the way; `Beside`: as in scan_partials_kernel.
constexpr int kVerifyReads = 1024;
template <class Beside>
__global__ __launch_bounds__(256) void verify_lengths_kernel(int32_t n_reads, const int32_t *len, const int32_t *seen, int32_t *rep_cnt, int32_t *err_flags,
                                                             int n_blocks, Beside beside)
{
    if ((int)blockIdx.x >= n_blocks) { beside((int)blockIdx.x - n_blocks); return; }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < kVerifyReads / 256; ++j) {
        const long long i = (long long)blockIdx.x * kVerifyReads + j * 256 + threadIdx.x;
        if (i < n_reads) { bad |= len[i] != seen[i]; rep_cnt[i] = 0; }
    }
    if (bad) atomicOr(err_flags, kErrHint);
}

// tile_first[k] = first read whose first window lies in tile k or later (tiles of Q windows).  For grouped input the same
// threads -- one per read, coalesced -- check the caller's offsets: they must not step back and the runs must chain from
// record 0 to record n_rec (kErrGroup; every later kernel of the pass then returns at once).
// It also clears the reads' repeat counters (a fill command of its own before) and, in a pass that was sized by the caller's
// window count, compares that count with the scan's (kErrHint; a fill command and a one-wave kernel of their own cost
// ~15 us of a 0.5 ms pass on an eighth of the human-scale set).
__global__ __launch_bounds__(256) void tile_first_kernel(int32_t n_reads, const long long *cov_off, Quantum qz,
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
    const long long t_r = (r < n_reads) ? min(qz.idx(cov_off[r]), n_tiles) : n_tiles;
    const long long t_p = (r > 0) ? min(qz.idx(cov_off[r - 1]), n_tiles) : -1;
    for (long long k = t_p + 1; k <= t_r; ++k) tile_first[k] = (int32_t)r;
}

// tile_first_kernel's work riding in the geometry scan's second launch (device_scan.hpp Post), for a pass whose sizes the host
// knows -- or assumes -- before anything has run (the caller's window count; what the context's last pass over a stream of this
// shape found): every read as the scan writes its offsets -- its repeat counter cleared, the caller's offsets checked, the tiles
// that begin inside it (read r owns the windows [off, off + nb): tile k begins at window k Q, and its first read is r + 1 when
// off < k Q <= off + nb) -- and on the last workgroup the sizes the pass was built on against the ones the scan found (kErrHint:
// every later kernel of the pass returns at once, raft_hip_finish runs it again with the host wait).
struct PrepPost {
    int32_t n_reads;
    Quantum qz;
    long long n_tiles;
    int32_t *tile_first, *rep_cnt, *err_flags;
    long long *err_index;
    GroupedOff grp;
    int32_t n_runs;
    long long n_rec;
    long long want_bins, cap_rep, cap_cut;        // the sizes the host built the pass on: windows (exact), reserved repeat slots and markers (at most)
    __device__ void operator()(long long r, const long long (&off)[3], const long long (&v)[3]) const
    {
        rep_cnt[r] = 0;
        if (grp.off) {
            bool bad = false;
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
        if (r == 0) tile_first[0] = 0;
        const long long k0 = qz.idx(off[0]) + 1, k1 = r + 1 == n_reads ? n_tiles : min(qz.idx(off[0] + v[0]), n_tiles);
        for (long long k = k0; k <= k1; ++k) tile_first[k] = (int32_t)(r + 1);
    }
    __device__ void closing(const long long (&tot)[3]) const
    {
        if (tot[0] != want_bins || tot[1] > cap_rep || tot[2] > cap_cut) atomicOr(err_flags, kErrHint);
    }
};

// ---- grouped input of more runs than the pileup kernels take (kMaxSeg): merged into ONE run first ----------------------
// (a PAF concatenated from many files; the intervals a rank of a pre-split job receives from its peers, two runs each).
// With the offsets at hand this needs no histogram and no atomics: read r's records of run j go behind its records of the
// runs before, at sum_j' (off_j'[r] - off_j'[0]) + sum_{j' < j} count_j'(r).  24 bytes of traffic per record.
// (kMaxRuns: raft_types.hpp)

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

// Window records (pileup_wave.hpp IN = 1) for the passes that need coordinate columns -- more runs than its instantiations take, the merge
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

// ---- bucketing by read id for streams of any order: sort, do not scatter (round 4) --------------------------------------------
// The counting sort above -- one atomic per record on its read's counter, then a 12-byte write per record wherever its read's
// bucket lies -- is fine where the stream is almost sorted and took 87 ms for 2.9e8 SHUFFLED records (every write a partial
// line somewhere in 3.5 GB).  For large inputs the intervals are sorted instead: expand_sides_kernel writes (read id,
// start | end << 32) per side -- query sides, and target sides of records whose two reads differ while the PAF is not
// symmetric (chop.hpp:165-169); sides that do not exist get the key n_reads and sort behind everything -- a radix sort by
// the id's bits (sort_pairs.hpp radix_sort_by_key: hand-written since round 5 -- LDS-staged digits of eight bits, every store part
// of a run, tiles dealt to the XCDs in contiguous eighths; round 4 called rocprim's device radix sort here), and unzip_sorted_kernel
// writes the three columns the pileup kernels read plus where every read's intervals begin.
// (Measured and dropped, round 5: two partition steps instead of a sort -- every side to a coarse bucket of 2048 reads (LDS ranks,
// one global atomic per bucket and 128 k-record tile), then one workgroup per bucket with an LDS counter per read.  Bit-exact, no
// library, 48 B of traffic per side -- and 25 ms where the sort takes 9.6: a wave's 64 lanes store to 64 different lines, and this
// part retires such a store at ~34 ps per LANE whether the lines sit in L2 or not (coarse step 14.8 ms, fine step 9.8 ms for
// 2.9e8 sides; profiles/r05_partition_kernel_stats.txt).  Stores have to leave in runs, which takes LDS-staged digits of 8 bits:
// the three passes of sort_pairs.hpp.)
__global__ __launch_bounds__(256) void expand_sides_kernel(long long n_rec, int32_t n_reads, int symmetric, const int32_t *qid, const int32_t *qs, const int32_t *qe,
                                                           const int32_t *tid, const int32_t *ts, const int32_t *te, uint32_t *key, unsigned long long *val,
                                                           int32_t *err_flags, long long *err_index)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (long long)gridDim.x * blockDim.x) {
        const int q = qid[i];
        const bool okq = q >= 0 && q < n_reads;
        bool bad = !okq;
        key[i] = okq ? (uint32_t)q : (uint32_t)n_reads;
        val[i] = (unsigned long long)(uint32_t)qs[i] | ((unsigned long long)(uint32_t)qe[i] << 32);
        if (!symmetric) {
            const int t = tid[i];
            const bool okt = t >= 0 && t < n_reads;
            bad = bad || !okt;
            key[n_rec + i] = (okt && t != q) ? (uint32_t)t : (uint32_t)n_reads;
            val[n_rec + i] = (unsigned long long)(uint32_t)ts[i] | ((unsigned long long)(uint32_t)te[i] << 32);
        }
        if (bad) {
            atomicOr(err_flags, kErrReadId);
            atomicMin((unsigned long long *)err_index, (unsigned long long)i);
        }
    }
}

// Reads without intervals begin where the next read with intervals does.  A thread that meets a step of the sorted keys fills
// the offsets of the reads between itself -- unless they are many (a rank's slice of a pre-split job names an eighth of the reads:
// one thread would write millions of entries one after the other): such a gap goes to a short list that
// fill_gaps_kernel serves with a workgroup per gap.
constexpr int kGapInline = 32, kGapList = 1024;
struct GapList { int32_t n; int32_t pad; long long lo[kGapList], hi[kGapList], val[kGapList]; };   // off[lo .. hi] = val

__device__ __forceinline__ void fill_or_list(long long lo, long long hi, long long val, long long *__restrict__ off, GapList *gaps)
{
    if (hi < lo) return;
    if (hi - lo >= kGapInline && gaps) {
        const int slot = atomicAdd(&gaps->n, 1);
        if (slot < kGapList) { gaps->lo[slot] = lo; gaps->hi[slot] = hi; gaps->val[slot] = val; return; }
    }
    for (long long r = lo; r <= hi; ++r) off[r] = val;
}

__global__ __launch_bounds__(256) void unzip_sorted_kernel(long long n_ent, int32_t n_reads, const uint32_t *__restrict__ key, const unsigned long long *__restrict__ val,
                                                           int32_t *__restrict__ b_rid, int32_t *__restrict__ b_s, int32_t *__restrict__ b_e, long long *__restrict__ off,
                                                           GapList *gaps)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_ent; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t k = key[i];
        const long long prev = i == 0 ? -1 : (long long)key[i - 1];
        if ((long long)k != prev) fill_or_list(prev + 1, (long long)k, i, off, gaps);      // (r == n_reads: the end)
        if (k < (uint32_t)n_reads) {
            const unsigned long long v = val[i];
            b_rid[i] = (int32_t)k; b_s[i] = (int32_t)(uint32_t)v; b_e[i] = (int32_t)(uint32_t)(v >> 32);
        }
        if (i == n_ent - 1) fill_or_list((long long)k + 1, (long long)n_reads, n_ent, off, gaps);
    }
}

__global__ __launch_bounds__(256) void fill_gaps_kernel(const GapList *__restrict__ gaps, long long *__restrict__ off)
{
    const int n = min(gaps->n, kGapList);
    for (int g = blockIdx.x; g < n; g += gridDim.x) {
        const long long lo = gaps->lo[g], hi = gaps->hi[g], v = gaps->val[g];
        for (long long r = lo + threadIdx.x; r <= hi; r += blockDim.x) off[r] = v;
    }
}

// ---- the same with WINDOW RECORDS (round 5): what profileCoverage uses of an interval is the windows it touches (repeat.hpp:69-72),
// two 16-bit indices where reads stay below 65,535 windows.  A side is then ONE 64-bit item -- read id | first window << 32 | one past
// the last << 48 -- 8 bytes through every pass of the sort instead of 12, and what comes out is the pileup kernel's leanest input
// (pileup_wave.hpp IN = 1: a word per record, the reads' offsets).  A side whose windows do not fit 16 bits raises kErrWide and the
// pass is run again with the coordinate pairs above; a negative coordinate is reported here (the record's index), an interval past
// its read's last window by the pileup kernel (the index into the bucketed array), as on the coordinate route.
// (kErrWide: raft_types.hpp)
__device__ __forceinline__ unsigned long long side_item(int rid, int s, int e, const FastDiv &by_reso, bool &neg, bool &wide)
{
    neg = neg || (s | e) < 0;
    unsigned first = 0, last1 = 0;
    // (multiply-high divisions: two hardware divisions per side were most of what the first pass of the sort executed)
    if (e > 0 && (s | e) >= 0) { first = (unsigned)fdiv(by_reso, s); last1 = (unsigned)fdiv(by_reso, e - 1) + 1u; }
    wide = wide || first > 65535u || last1 > 65535u;
    return (unsigned long long)(uint32_t)rid | ((unsigned long long)(first & 0xffffu) << 32) | ((unsigned long long)(last1 & 0xffffu) << 48);
}
// The sides of the record columns as the item sort's first pass takes them (sort_pairs.hpp: Src): side j < n_rec is the query side of
// record j, side n_rec + i the target side of record i (a non-symmetric PAF only; a side that does not exist -- a target on the query's
// own read, an id out of range -- has the key n_reads and sorts behind every read).  Until round 6 expand_sides_win_kernel wrote
// these items out and the first pass read them back; now the histogram reads the id columns and the scatter makes the items.
// What the input can have wrong is reported by the scatter (item<true>), per side: an id outside [0, n_reads) (kErrReadId), a
// negative coordinate (kErrCoord; neither for a side that does not exist), windows beyond 16 bits (kErrWide) -- the record's index.
struct SideSource {
    long long n_rec;
    int32_t n_reads, symmetric;
    FastDiv by_reso;
    const int32_t *qid, *qs, *qe, *tid, *ts, *te;
    int32_t *err_flags;
    long long *err_index;
    __device__ __forceinline__ uint32_t key(long long j) const
    {
        if (j < n_rec) { const int q = qid[j]; return q >= 0 && q < n_reads ? (uint32_t)q : (uint32_t)n_reads; }
        const long long i = j - n_rec;
        const int t = tid[i];
        return t >= 0 && t < n_reads && t != qid[i] ? (uint32_t)t : (uint32_t)n_reads;
    }
    struct Raw { int r, s, e, q; };
    __device__ __forceinline__ Raw load(long long j) const      // (plain loads, nothing dependent: a batch of them is in flight at once)
    {
        const bool query = j < n_rec;
        const long long i = query ? j : j - n_rec;
        const int32_t *ids = query ? qid : tid, *ss = query ? qs : ts, *ee = query ? qe : te;
        return Raw{ids[i], ss[i], ee[i], qid[i]};
    }
    template <bool REPORT> __device__ __forceinline__ unsigned long long make(const Raw &v, long long j) const
    {
        const bool query = j < n_rec;
        const long long i = query ? j : j - n_rec;
        const bool ok = v.r >= 0 && v.r < n_reads;
        const bool use = ok && (query || v.r != v.q);
        bool neg = false, wide = false;
        const unsigned long long x = side_item(use ? v.r : n_reads, v.s, v.e, by_reso, neg, wide);
        if (REPORT) {
            if (!ok) {
                atomicOr(err_flags, kErrReadId);
                atomicMin((unsigned long long *)err_index, (unsigned long long)i);
            } else if (use && neg) {
                atomicOr(err_flags, kErrCoord);
                atomicMin((unsigned long long *)err_index, (unsigned long long)i);
            } else if (use && wide) atomicOr(err_flags, kErrWide);
        }
        return use ? x : (unsigned long long)(uint32_t)n_reads;
    }
};

__global__ __launch_bounds__(256) void unzip_items_kernel(long long n_ent, int32_t n_reads, const unsigned long long *__restrict__ item, uint32_t *__restrict__ b_win,
                                                          long long *__restrict__ off, GapList *gaps)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_ent; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long x = item[i];
        const uint32_t k = (uint32_t)x;
        const long long prev = i == 0 ? -1 : (long long)(uint32_t)item[i - 1];
        if ((long long)k != prev) fill_or_list(prev + 1, (long long)k, i, off, gaps);
        if (k < (uint32_t)n_reads) b_win[i] = (uint32_t)(x >> 32);
        if (i == n_ent - 1) fill_or_list((long long)k + 1, (long long)n_reads, n_ent, off, gaps);
    }
}

} // namespace raft
