// pileup.hpp -- the dominant kernel: binned coverage pileup + prefix scan +
// coalesced coverage store + high-coverage run detection.
//
// Reference semantics reproduced (closed forms of SURVEY.md §3.2, checked
// against oracle/raft_oracle.c and the compiled reference):
//   profileCoverage  repeat.hpp:28-79   interval (s,e) adds 1 to windows s/reso .. (e-1)/reso
//   run scan         repeat.hpp:111-168 maximal runs of windows with cov >= high_cov,
//                                       kept when (#windows*reso) >= repeat_length,
//                                       widened by flanking_length and clamped to [0,len]
//
// Work decomposition (MI355X-first; the kernel is bound by the HBM write of cov[]):
//   * cov[] (4 B per window, all reads concatenated in FASTA order) is cut into tiles by a
//     quantum of Q windows; tile k owns the reads whose first window falls in [kQ,(k+1)Q).
//     A 72-byte TileDesc per tile (reads, windows, interval ranges) is built on the device.
//   * PERSISTENT workgroups (CUs x resident workgroups) walk the tiles.  While tile i is being
//     processed, the descriptor of tile i+2 and the read offsets + first U x THREADS intervals of
//     tile i+1 are already in flight into registers: a wave waits for them once, just before
//     it starts storing tile i (vmcnt is one in-order counter for loads and stores, so a wait
//     placed after the stores would also wait for the stores -- measured, tools/stamp_probe.py).
//   * the tile's windows are staged in LDS as a difference array: +1 at the first window of
//     an interval, -1 one past its last (ds_add_u32); a plain prefix sum over the
//     concatenated reads yields every read's coverage (each read's +1/-1 balance out before
//     the next read begins).  Barriers order LDS only (s_waitcnt lgkmcnt(0); s_barrier).
//   * each wave owns a contiguous quarter of the tile: rows of 256 windows (int4 per lane)
//     are prefix-summed with a DPP wave scan and a scalar carry and stored as aligned
//     1 KiB-per-instruction wave stores -- coverage is written exactly once, never re-read.
//   * the >= high_cov predicate of a row is four 64-bit ballots held in SGPRs; run starts /
//     ends are scalar bit logic on them, so rows without a high window (the common case) cost
//     no vector work for the repeat scan; runs crossing wave seams (or chunk seams of a read
//     longer than the LDS window) are stitched by one wave from five words per wave.
//
// Algorithmic bytes per launch: 12*I + 4*B + 4*N + 8*R.
#pragma once
#include "wave.hpp"

namespace raft {

constexpr int kMaxSeg = 4;   // sorted runs of the record stream the fast path accepts (hifiasm cis+trans = 2)

enum : int {
    kErrReadId = 1 << 0,
    kErrCoord = 1 << 1,
    kErrFragment = 1 << 2,
    kErrInternal = 1 << 3,
    kErrLen = 1 << 4,
    kErrExtra = 1 << 6,         // the list of extra tiles (split tiles, pieces of long reads) overflowed: the engine runs
                               // the pass again with the general kernel for those tiles
    kErrOrder = 1 << 5,         // a pass that trusted a sampled guess of the sorted runs met a record that refutes it (not an
                               // error of the input: the engine runs the pass again after looking at every record)
    kErrHint = 1 << 7,          // the number of windows the caller announced (a pass without a host wait is sized by it) is not
                               // what the read lengths give: every later kernel of the pass returns at once, the engine
                               // runs the pass again with the host wait
    kErrGroup = 1 << 8,         // grouped input (raft_hip_run_device_grouped): the per-read record offsets step back or do not
                               // chain from 0 to n_rec
    kErrStop = kErrHint | kErrGroup   // what a pass without a host wait cannot go on after: its later kernels return at once
};

// Grouped input (include/raft_hip.h raft_hip_run_device_grouped): the record stream is n_runs runs sorted by query id and
// the caller says where each read's records begin in each run -- off[s * stride + r] (+ adj[s]: a chunk of the host
// pipelines uploads slices of the caller's arrays and of its record columns) is the first record of read r in run s,
// entry n_reads closes the run.  Tile cuts are then look-ups, not searches.
struct GroupedOff {
    const long long *off;      // nullptr: not grouped
    long long stride;          // n_reads + 1
    long long adj[kMaxSeg];
    __device__ __forceinline__ long long at(int s, long long r) const { return off[s * stride + r] + adj[s]; }
};

constexpr int kExcPerTile = 16;   // delta4: listed windows a tile keeps in slots of its own (a HiFi tile lists ~12: its first window, large steps at read boundaries)

struct SegStarts { long long start[kMaxSeg + 1]; int32_t n_seg; };

struct TileDesc {              // written by tile_desc_kernel; 18 dwords
    int32_t r_lo, r_hi;        // reads [r_lo, r_hi) start in this tile
    int32_t n_iv[kMaxSeg];     // intervals of those reads in segment s
    long long g_lo, g_hi;      // their windows [g_lo, g_hi) in cov[]
    long long iv_lo[kMaxSeg];  // first interval in segment s
};
constexpr int kDescDwords = (int)(sizeof(TileDesc) / 4);

// The same tiling as seen by pileup_fast_kernel (pileup_fast.hpp): boundary k says where tile k begins; entry
// n_tiles closes the last tile.  Two adjacent cuts describe a tile and arrive as one 64-byte scalar load.
struct TileCut {
    int32_t r_lo;              // first read of tile k
    int32_t flags;             // kCutFast: tile k holds whole reads that fit one LDS window
    int32_t iv_lo[kMaxSeg];    // first interval of tile k in segment s (absolute index, < 2^31 checked by the host)
    long long g_lo;            // first window of tile k in cov[]
};
static_assert(sizeof(TileCut) == 32, "two adjacent cuts are one 64-byte scalar load");
enum : int {
    kCutFast = 1,              // tile k holds whole reads that fit one LDS window
    kCutPiece = 2,             // (extra entries only) the tile is a piece of ONE read longer than the LDS window
};
static_assert(sizeof(TileDesc) == 72, "descriptor is fetched as 18 dwords, one per lane");

struct PileupArgs {
    // intervals: sorted by read id inside each of n_seg segments
    const int32_t *iv_rid, *iv_s, *iv_e;
    // ... or "window records" (pileup_fast.hpp IN = 1): one word per record, first window | one past the last << 16, no read
    // ids -- the reads' records are where the caller's offsets (grp) say
    const uint32_t *iv_w;
    GroupedOff grp;
    int32_t n_seg;
    // tiles and reads
    const TileDesc *td;
    long long n_tiles;
    const int32_t *read_len;
    const long long *cov_off;   // [n_reads+1]
    int32_t n_reads;
    // params
    int32_t reso, high_cov, repeat_length, flank;
    uint32_t div_magic;           // n / reso == mulhi(n, div_magic) >> div_shift for 0 <= n < 2^31 (reso > 1)
    int32_t div_shift;            // -1: reso == 1
    // outputs
    int32_t *cov;
    // pileup_fast_kernel instantiated with OW = 1 or 2 writes the transfer encoding of cov[] instead (pack.hpp: OW bytes per
    // window, min(cov, 255 / 65535), plus the list of the windows at or above that limit) and leaves `cov` alone
    void *covp;
    int32_t *cov_anchor;          // OW = 8 (pack.hpp kCovDelta4, four bits per window): cov[1024 k - 1] per block of 1024 windows
    int32_t d4_shift;             //     ... blocks counted from d4_shift windows before this pass's first (a chunk of a larger array: multiple of 4, < 1024)
    long long *exc_pidx;          // ... and the windows it lists, kExcPerTile slots per tile (regular tiles, then the extra ones)
    int32_t *exc_pval;
    int32_t *exc_tile_n;          //     how many of its slots a tile used (zeroed before the pass)
    unsigned long long *n_exc;    // windows at or above the limit (counted even when the list is full)
    long long exc_cap;
    long long *exc_idx;
    int32_t *exc_val;
    const long long *rep_res_off; // [n_reads+1] reserved slots for raw repeats
    int32_t *rep_cnt;             // [n_reads], zeroed
    int32_t *raw_key, *raw_s, *raw_e;
    long long *block_sums;        // [2*gridDim.x]: sum of coverage, sum of unclamped repeat bp
    int32_t *err_flags;           // device word, OR of kErr*
    long long *err_index;         // first offending interval index (min)
    unsigned long long *dbg;      // diagnostic build only: [n_tiles][16] s_memtime stamps
    // When pileup_fast_kernel takes the tiles of whole reads, this kernel walks only the others:
    const int32_t *slow_list;     // tile ids (any order), or nullptr: every tile is handled here
    const int32_t *n_slow;        // device count of slow_list
    int32_t *tile_counter;        // pileup_fast_kernel: tiles are handed out through this counter (zeroed by the host)
    int32_t tile_batch;           // ... in batches of this many consecutive tiles
    // Tiles that do not fit the fast kernel as they are (more windows than the LDS window, more reads than its tables, a
    // read longer than the window) are re-cut by tile_desc_kernel into EXTRA tiles that do: groups of whole reads, and
    // pieces of piece_w windows of a long read.  They follow the regular boundaries in the cut array as explicit
    // (begin, end) pairs: extra tile j = cuts[n_tiles + 1 + 2j], cuts[n_tiles + 2 + 2j].
    const int32_t *n_extra;       // device count of extra tiles (nullptr: none, the general kernel takes those tiles)
    int32_t piece_w;
    int32_t *slow_counter;        // list mode of this kernel: the items of slow_list are handed out one by one
    // tiles too deep for the wave kernel's 16-bit difference array (pileup_deep.hpp): listed by it, piled up by pileup_deep_kernel
    void *deep_list;              // DeepTile[deep_cap]
    int32_t *n_deep;              // device count (may exceed deep_cap: kErrDeep, the pass is run again with room)
    int32_t deep_cap, deep_min;   // deep_min: intervals on a tile from which it goes that way (2^15; tests lower it)
    unsigned long long *deep_rep_total;   // where pileup_deep_kernel adds its tiles' unclamped repeat bases (Ctrl::totals[1])
};

// Coarse index of the record stream (bucket.hpp guess_runs_kernel writes it, tile_desc_kernel reads it): the read id of
// every 2^sh-th record and of the last one, with sh the smallest shift that leaves at most kSamples strides.
constexpr int kGuessBlocks = 1024;                         // x 256 threads: one sample per thread (64 blocks, 16 k samples until round 5:
                                                           // tile_desc_kernel's time is the lines its probes BEHIND the samples pull in)
constexpr int kSamples = kGuessBlocks * 256;
__host__ __device__ __forceinline__ int sample_shift(long long n)
{
    int sh = 0;
    while (((n - 1) >> sh) >= kSamples) ++sh;
    return sh;
}
// samples j = 0 .. n_samples - 1 sit at min(j << sh, n - 1)
__host__ __device__ __forceinline__ long long n_samples(long long n, int sh) { return ((n - 1) >> sh) + 2; }
__host__ __device__ __forceinline__ long long sample_pos(long long j, long long n, int sh)
{
    const long long p = j << sh;
    return p < n - 1 ? p : n - 1;
}

constexpr int kRunQ = 16; // parked runs per wave and window before falling back to immediate emission
constexpr int kOpen = -2; // run began before this wave's first window
constexpr int kNone = -1;

template <int THREADS, int CAP>
struct PileupSmem {
    static constexpr int NW = THREADS / 64;
    static constexpr int SLOTS = CAP + 256; // window slots: 3 alignment + CAP + 1 sentinel, rounded to rows
    static constexpr int SBW = SLOTS / 32;
    static constexpr int MAXR = THREADS - 2; // reads per window: their offsets, lengths, repeat slots live in LDS
    static_assert(NW * 8 <= 64, "wave seam words are read by one wave instruction");
    int32_t diff[SLOTS];
    uint32_t sbits[SBW];
    int32_t roff[MAXR + 2];                 // first slot of read r_a+j relative to a0 (j <= nr)
    int32_t rlen[MAXR + 2];                 // read length
    int32_t rcnt[MAXR + 2];                 // raw repeats emitted so far for the read
    int32_t rres[MAXR + 2];                 // its first reserved raw-repeat slot (rep_res_off, < 2^31 checked by the host)
    unsigned long long acc_cov, acc_rep;
    long long carry_open[2];                // a long read's open run across chunks: chunk c reads [c & 1], writes [~c & 1]
    int32_t carry_hp[2];
    int32_t wsum[NW];
    int32_t wst[NW * 8];                    // per wave: rows, pclose, sfinal, hpfinal, hpin, need
    unsigned long long stamps[16];          // diagnostic build
    int32_t item;                           // list mode: the item thread 0 drew for the workgroup
    int32_t runq_n[NW];                     // per wave: closed runs parked for emission (slots relative to a0)
    int32_t runq[NW * 2 * kRunQ];
};

// window index of base n (0 <= n < 2^31): n / reso without a hardware divide
__device__ __forceinline__ unsigned win_of(const PileupArgs &a, unsigned n)
{
    return a.div_shift < 0 ? n : (__umulhi(n, a.div_magic) >> a.div_shift);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's global
// stores (s_waitcnt vmcnt(0)), which would park every wave behind its own coverage stores.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// A closed run of high windows [sS, sT) (slots of the current LDS window, i.e. relative to a0) of the window's
// j-th read -> one raw repeat record.  Everything it needs about the read sits in LDS tables staged with the
// window (four independent LDS reads = one round trip); the only global traffic is three fire-and-forget stores,
// so nothing here waits behind the wave's coverage stores (vmcnt is in-order).
template <class Smem>
__device__ __forceinline__ void emit_run_of(const PileupArgs &a, Smem &sm, int j, int sS, int sT)
{
    const int nwin = sT - sS;
    if ((long long)nwin * a.reso < (long long)a.repeat_length) return; // repeat.hpp:125,150
    const int len = sm.rlen[j], off = sm.roff[j], r0 = sm.rres[j], r1 = sm.rres[j + 1];
    const int slot = atomicAdd(&sm.rcnt[j], 1);
    const int start = (sS - off) * a.reso;
    const int end = start + nwin * a.reso;
    int s = start - a.flank, e = end + a.flank;   // repeat.hpp:129-140
    if (s <= 0) s = 0;
    if (e >= len) e = len;
    if (slot >= r1 - r0) { raise_error(a, kErrInternal, j); return; }
    const long long idx = (long long)r0 + slot;
    a.raw_key[idx] = start;
    a.raw_s[idx] = s;
    a.raw_e[idx] = e;
    atomicAdd(&sm.acc_rep, (unsigned long long)(end - start)); // repeat.hpp:127,152
}

// read of the window (0 .. nr-1) that owns slot sS; reads without windows are skipped
template <class Smem>
__device__ __forceinline__ int owner_slot(const Smem &sm, int nr, int sS)
{
    int lo = 0, hi = nr;                         // invariant: roff[lo] <= sS < roff[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sm.roff[mid] <= sS) lo = mid; else hi = mid;
    }
    return lo;
}

template <class Smem>
__device__ __forceinline__ void emit_run(const PileupArgs &a, Smem &sm, int nr, int sS, int sT)
{
    if ((long long)(sT - sS) * a.reso < (long long)a.repeat_length) return;
    // (a run handed over in the concatenated windows may span a read boundary -- pileup_fast.hpp LS: it is one run per read,
    // repeat.hpp:111-112, each judged by its own length)
    int j = owner_slot(sm, nr, sS);
    while (sS < sT) {
        const int e = min(sT, sm.roff[j + 1]);
        if (e > sS) emit_run_of(a, sm, j, sS, e);
        sS = max(sS, e); ++j;
    }
}

// lower bound of read id `r` in iv_rid[lo, hi); per-lane and wave-uniform forms
__device__ __forceinline__ long long lower_bound_rid(const int32_t *iv_rid, long long lo, long long hi, int r)
{
    while (lo < hi) {
        long long mid = (lo + hi) >> 1;
        if (iv_rid[mid] < r) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ long long lower_bound_rid_uni(const int32_t *iv_rid, long long lo, long long hi, int r)
{
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (uni(iv_rid[mid]) < r) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// What a lane holds of a window before the window is processed: two read offsets and four intervals.
template <int U>
struct Prefetch {
    int rid[U], st[U], en[U];
    long long cv;           // cov_off of read r_a + thread-id (thread-id <= nr)
    int rr, rl;             // its first reserved raw-repeat slot (low 32 bits; the host checks the total) and length
};

struct TileRegs {            // descriptor unpacked into scalars
    int r_lo, r_hi;
    long long g_lo, g_hi;
    long long seg_lo[kMaxSeg];
    int seg_cum[kMaxSeg + 1];
};

// v-th interval of a window whose segments are (seg_lo[s], seg_cum[s] .. seg_cum[s+1]): its index relative to
// seg_lo[0], in 32 bits (the host guarantees fewer than 2^30 intervals, so the byte offset fits 32 bits and the
// loads use scalar base + 32-bit vector offset addressing)
__device__ __forceinline__ unsigned iv_rel_of(const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1], int v)
{
    int d = 0;
#pragma unroll
    for (int s = 1; s < kMaxSeg; ++s)
        if (v >= seg_cum[s]) d = (int)(seg_lo[s] - seg_lo[0]) - seg_cum[s];
    return (unsigned)(v + d);
}

template <int THREADS, int U>
__device__ __forceinline__ void load_intervals(const PileupArgs &a, int v0, const long long (&seg_lo)[kMaxSeg],
                                               const int (&seg_cum)[kMaxSeg + 1], Prefetch<U> &g)
{
    const int n_iv = seg_cum[kMaxSeg];
    const char *b_rid = reinterpret_cast<const char *>(a.iv_rid + seg_lo[0]);
    const char *b_s = reinterpret_cast<const char *>(a.iv_s + seg_lo[0]);
    const char *b_e = reinterpret_cast<const char *>(a.iv_e + seg_lo[0]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int v = v0 + u * THREADS;
        const bool ok = v < n_iv;
        const unsigned off = iv_rel_of(seg_lo, seg_cum, v) * 4u;
        g.rid[u] = ok ? *reinterpret_cast<const int32_t *>(b_rid + off) : -1;
        g.st[u] = ok ? *reinterpret_cast<const int32_t *>(b_s + off) : 0;
        g.en[u] = ok ? *reinterpret_cast<const int32_t *>(b_e + off) : 0;
    }
}

template <int THREADS, int U>
__device__ __forceinline__ void issue_prefetch(const PileupArgs &a, int tid, int r_a, int nr,
                                               const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1],
                                               Prefetch<U> &g)
{
    g.cv = 0; g.rr = 0; g.rl = 0;               // nr <= MAXR = THREADS - 2: thread j covers read r_a + j (j <= nr)
    if (tid <= nr) {
        g.cv = a.cov_off[r_a + tid];
        g.rr = reinterpret_cast<const int32_t *>(a.rep_res_off)[2 * (long long)(r_a + tid)];
        g.rl = (tid < nr) ? a.read_len[r_a + tid] : 0;
    }
    load_intervals<THREADS, U>(a, tid, seg_lo, seg_cum, g);
}

// Diagnostic build only: s_memtime stamps are parked in LDS (a global store would queue behind the coverage stores
// and measure the store queue instead of the phase) and dumped once per tile.
#define RAFT_STAMP(slot)                                                                      \
    do {                                                                                      \
        if (DIAG && threadIdx.x == 0 && a.dbg) sm.stamps[(slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// One LDS window: global windows [w_lo, w_hi) (at most CAP) belonging to reads [r_a, r_b).
// single_read: the window is a chunk of one long read r_a (intervals are clipped to the chunk).
// g holds the window's prefetched loads (issue_prefetch with the same arguments).
template <int THREADS, int CAP, int U, bool DIAG>
__device__ void pile_window(const PileupArgs &a, PileupSmem<THREADS, CAP> &sm, int r_a, int r_b,
                            long long w_lo, long long w_hi, bool single_read, bool first_chunk, bool last_chunk, int cpar,
                            const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1], Prefetch<U> &g,
                            long long stamp_row)
{
    using Smem = PileupSmem<THREADS, CAP>;
    constexpr int NW = Smem::NW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = uni((int)(threadIdx.x >> 6)); // wave-uniform: keeps row counters, carries and ballots in SGPRs
    const long long a0 = w_lo & ~3LL;          // 16-byte aligned base of the staged range
    const int off0 = (int)(w_lo - a0);         // first valid slot
    const int t_end = off0 + (int)(w_hi - w_lo); // one past the last valid slot
    const int rows = (t_end + 1 + 255) >> 8;   // rows of 256 slots, sentinel slot included
    const int nr = r_b - r_a;
    const int n_iv = seg_cum[kMaxSeg];

    // 1. clear the difference array and the read-start bits; stage the reads' first slots
    for (int i = tid * 4; i < rows * 256; i += THREADS * 4)
        *reinterpret_cast<int4 *>(&sm.diff[i]) = make_int4(0, 0, 0, 0);
    for (int i = tid; i < rows * 8; i += THREADS) sm.sbits[i] = 0u;
    if (tid <= nr) {
        sm.roff[tid] = (int)(g.cv - a0);
        sm.rlen[tid] = g.rl;
        sm.rres[tid] = g.rr;
        if (!(single_read && !first_chunk)) sm.rcnt[tid] = 0;   // a long read keeps counting across its chunks
    }
    lds_barrier();
    RAFT_STAMP(2);

    // 2. read-start bits (a run never continues across a read boundary, repeat.hpp:111-112)
    if (single_read) {
        if (first_chunk && tid == 0) sm.sbits[0] = 1u << off0;
    } else {
        for (int j = tid; j < nr; j += THREADS) {
            const int p = sm.roff[j];
            atomicOr(&sm.sbits[p >> 5], 1u << (p & 31));
        }
    }

    // 3. intervals -> +1 / -1 (profileCoverage, closed form); U records in flight per lane.  One predicated region
    //    per record (the two LDS adds); malformed records only set a lane flag that is looked at once per window.
    int covsum = 0;
    int bad_v = -1;                              // a record of this lane with a coordinate error (virtual index)
    bool bad_order = false;                      // a record whose read is not one of this window's (see kErrOrder)
    for (int v0 = tid; v0 < n_iv; v0 += THREADS * U) {
        if (v0 != tid) load_intervals<THREADS, U>(a, v0, seg_lo, seg_cum, g);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int st = g.st[u], en = g.en[u];
            // (a record of a read outside this window can only come from a mis-speculated pass -- engine.hip run_pass --
            // whose results are thrown away; it must not index the tables)
            const bool valid = (unsigned)(g.rid[u] - r_a) < (unsigned)nr;
            if (!valid && v0 + u * THREADS < n_iv) bad_order = true;     // a real record (not an empty slot) of a foreign read
            const int j = valid ? g.rid[u] - r_a : 0;
            const int b0 = sm.roff[j], nb_r = sm.roff[j + 1] - b0;
            const int first = (int)win_of(a, (unsigned)st);
            int last = (en > 0) ? (int)win_of(a, (unsigned)(en - 1)) : -1;
            const bool neg = (st | en) < 0;
            const bool over = !neg && last >= first && last >= nb_r; // reference writes past its vector (repeat.hpp:69-72)
            if (valid && (neg || over)) bad_v = v0 + u * THREADS;
            last = min(last, nb_r - 1);
            int pf = b0 + first, pl1 = b0 + last + 1;   // slots relative to a0
            if (single_read) { pf = max(pf, off0); pl1 = min(pl1, t_end); }
            if (valid && !neg && pf < pl1) {
                __hip_atomic_fetch_add(&sm.diff[pf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&sm.diff[pl1], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                covsum += pl1 - pf;        // sum of coverage over the window == windows touched by its intervals
            }
        }
    }
    if (__ballot(bad_v >= 0) != 0ull) {          // rare
        if (bad_v >= 0) raise_error(a, kErrCoord, seg_lo[0] + (long long)iv_rel_of(seg_lo, seg_cum, bad_v));
    }
    if (__ballot(bad_order) != 0ull && lane == 0) atomicOr(a.err_flags, kErrOrder);
    {
        const long long cs = wave_reduce_add64((long long)covsum);
        if (lane == 0 && cs) atomicAdd(&sm.acc_cov, (unsigned long long)cs);
    }
    lds_barrier();
    RAFT_STAMP(3);

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
    lds_barrier();
    RAFT_STAMP(4);

    // Every load issued so far -- including the NEXT tile's prefetch -- must land before this wave's first
    // coverage store: after the stores, any vmcnt wait would also wait for the stores.
    wait_all_loads();
    RAFT_STAMP(15);

    // 5. pass B: prefix sum, store, run detection.  Rows that lie entirely inside the window (all but the first
    //    and last of a window) take the lean path: no per-slot validity masks.
    int carry = 0;
    for (int w = 0; w < wid; ++w) carry += uni(sm.wsum[w]);
    bool hp; // was the window just before this wave's first slot high (and in the same run domain)?
    if (wid == 0) hp = single_read && !first_chunk && uni(sm.carry_hp[cpar]) != 0;
    else hp = (row_b < rows) && (carry >= a.high_cov);
    const bool hp_in = hp;
    int S = hp ? kOpen : kNone;  // start slot of the run currently open
    int pclose = -1;             // slot at which the run inherited from before this wave closed

    // the repeat scan of one row, given its four >= high_cov ballots (already masked to valid slots)
    auto scan_row = [&](int row, int base, int p0, unsigned long long M0, unsigned long long M1, unsigned long long M2,
                        unsigned long long M3) {
        const unsigned long long VE0 = __ballot(p0 + 0 < t_end), VE1 = __ballot(p0 + 1 < t_end),
                                 VE2 = __ballot(p0 + 2 < t_end), VE3 = __ballot(p0 + 3 < t_end);
        const uint32_t word = sm.sbits[p0 >> 5];
        const uint32_t nib = (word >> (p0 & 31)) & 0xFu;
        const unsigned long long SB0 = __ballot(nib & 1u), SB1 = __ballot(nib & 2u),
                                 SB2 = __ballot(nib & 4u), SB3 = __ballot(nib & 8u);
        // P_k: the slot before (lane,k) is a high window; the carried-in bit belongs to the first valid slot:
        // slot off0 of row 0, else slot 0 of the row
        const unsigned long long hb = hp ? 1ull : 0ull;
        const int hk = (row == 0) ? off0 : 0;
        const unsigned long long P0 = (M3 << 1) | (hk == 0 ? hb : 0ull), P1 = M0 | (hk == 1 ? hb : 0ull),
                                 P2 = M1 | (hk == 2 ? hb : 0ull), P3 = M2 | (hk == 3 ? hb : 0ull);
        const unsigned long long CL0 = P0 & (~M0 | SB0) & VE0, CL1 = P1 & (~M1 | SB1) & VE1,
                                 CL2 = P2 & (~M2 | SB2) & VE2, CL3 = P3 & (~M3 | SB3) & VE3; // run ends before this slot
        const unsigned long long CA0 = M0 & (~P0 | SB0), CA1 = M1 & (~P1 | SB1),
                                 CA2 = M2 & (~P2 | SB2), CA3 = M3 & (~P3 | SB3);             // run starts at this slot
        if ((CL0 | CL1 | CL2 | CL3) != 0ull) {
            const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
            unsigned cl4 = (unsigned)((CL0 >> lane) & 1ull) | (unsigned)(((CL1 >> lane) & 1ull) << 1) |
                           (unsigned)(((CL2 >> lane) & 1ull) << 2) | (unsigned)(((CL3 >> lane) & 1ull) << 3);
#pragma unroll 1
            while (cl4) {                        // rare: this lane sees the end of a run
                const int k = __builtin_ctz(cl4);
                cl4 &= cl4 - 1u;
                int best = S;                    // latest run start at a slot before (lane, k)
                unsigned long long m;
                m = CA0 & (0 < k ? le : lt); if (m) best = max(best, base + 4 * top_bit(m) + 0);
                m = CA1 & (1 < k ? le : lt); if (m) best = max(best, base + 4 * top_bit(m) + 1);
                m = CA2 & (2 < k ? le : lt); if (m) best = max(best, base + 4 * top_bit(m) + 2);
                m = CA3 & lt;                if (m) best = max(best, base + 4 * top_bit(m) + 3);
                const int t = p0 + k;
                if (best == kOpen) pclose = t;
                else if ((long long)(t - best) * a.reso >= (long long)a.repeat_length) { // repeat.hpp:125
                    // park the run; all parked runs become repeat records at once after the seams are resolved
                    const int q = atomicAdd(&sm.runq_n[wid], 1);
                    if (q < kRunQ) { sm.runq[(wid * kRunQ + q) * 2] = best; sm.runq[(wid * kRunQ + q) * 2 + 1] = t; }
                    else emit_run(a, sm, nr, best, t);
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
    };
    // prefix sum of one row of 256 slots, given the row's four differences per lane
    auto row_values = [&](const int4 d, int &c0, int &c1, int &c2, int &c3) {
        const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
        const int incl = wave_incl_scan_add(w);
        const int excl = incl - w + carry;
        carry += __builtin_amdgcn_readlane(incl, 63);
        c0 = excl + x; c1 = excl + y; c2 = excl + z; c3 = excl + w;
    };
    auto partial_row = [&](int row) {          // first / last rows of a window: per-slot validity
        const int base = row * 256, p0 = base + lane * 4;
        int c0, c1, c2, c3;
        row_values(*reinterpret_cast<const int4 *>(&sm.diff[p0]), c0, c1, c2, c3);
        const unsigned q0 = (unsigned)(p0 - off0), nbw_u = (unsigned)(t_end - off0);
        const unsigned long long M0 = __ballot(c0 >= a.high_cov && q0 + 0u < nbw_u), M1 = __ballot(c1 >= a.high_cov && q0 + 1u < nbw_u),
                                 M2 = __ballot(c2 >= a.high_cov && q0 + 2u < nbw_u), M3 = __ballot(c3 >= a.high_cov && q0 + 3u < nbw_u);
        if (q0 + 0u < nbw_u) a.cov[a0 + p0 + 0] = c0;
        if (q0 + 1u < nbw_u) a.cov[a0 + p0 + 1] = c1;
        if (q0 + 2u < nbw_u) a.cov[a0 + p0 + 2] = c2;
        if (q0 + 3u < nbw_u) a.cov[a0 + p0 + 3] = c3;
        if ((M0 | M1 | M2 | M3) != 0ull || hp) scan_row(row, base, p0, M0, M1, M2, M3);
    };
    {
        int row = row_b;
        // rows [full_b, full_e) are entirely inside [off0, t_end)
        const int full_b = max(row_b, (off0 + 255) >> 8), full_e = min(row_e, t_end >> 8);
        for (; row < min(full_b, row_e); ++row) partial_row(row);
        // the next row's LDS read is issued before the current row's scan: the read latency hides under the DPP chain
        int4 dn = make_int4(0, 0, 0, 0);
        if (row < full_e) dn = *reinterpret_cast<const int4 *>(&sm.diff[row * 256 + lane * 4]);
        for (; row < full_e; ++row) {
            const int base = row * 256, p0 = base + lane * 4;
            const int4 dc = dn;
            if (row + 1 < full_e) dn = *reinterpret_cast<const int4 *>(&sm.diff[p0 + 256]);
            int c0, c1, c2, c3;
            row_values(dc, c0, c1, c2, c3);
            *reinterpret_cast<int4 *>(&a.cov[a0 + p0]) = make_int4(c0, c1, c2, c3);
            const unsigned long long M0 = __ballot(c0 >= a.high_cov), M1 = __ballot(c1 >= a.high_cov),
                                     M2 = __ballot(c2 >= a.high_cov), M3 = __ballot(c3 >= a.high_cov);
            if ((M0 | M1 | M2 | M3) != 0ull || hp) scan_row(row, base, p0, M0, M1, M2, M3);
        }
        for (; row < row_e; ++row) partial_row(row);
    }

    // 6. publish the wave's seam state: rows, slot where the inherited run closed, start of the run open at the
    //    end, high at the end, high at the start
    {
        const unsigned long long pm = __ballot(pclose >= 0);
        int pc = -1;
        if (pm) pc = __builtin_amdgcn_readlane(pclose, (int)__builtin_ctzll(pm));
        pclose = pc;
        if (lane == 0) {
            *reinterpret_cast<int4 *>(&sm.wst[wid * 8]) = make_int4(row_e > row_b ? 1 : 0, pc, S, hp ? 1 : 0);
            sm.wst[wid * 8 + 4] = hp_in ? 1 : 0;
        }
    }
    RAFT_STAMP(5);
    lds_barrier();
    RAFT_STAMP(6);

    // 7. seams, resolved by every wave for itself (no serial walk): a run inherited from earlier waves starts at
    //    the run-start of the nearest earlier wave that saw one (or at the chunk carry); the wave holding the last
    //    valid slot closes the run that reaches the window end, or carries it into the next chunk.
    {
        const int v = (lane < NW * 8) ? sm.wst[lane] : 0;
        // (read from the slot of this chunk's parity: the wave holding the last slot writes the next chunk's value in
        // this same phase, and a slower wave must not pick that up -- it did, on 1 run in 3e5, before the slots existed)
        const long long carry_open = (single_read && !first_chunk) ? uni(sm.carry_open[cpar]) : -1;
        auto run_start_before = [&](int w) -> long long { // start (global window) of the run open at the end of wave w
#pragma unroll
            for (int y = NW - 1; y >= 0; --y) {
                if (y > w) continue;
                if (!__builtin_amdgcn_readlane(v, y * 8 + 0)) continue;
                const int sf = __builtin_amdgcn_readlane(v, y * 8 + 2);
                if (sf != kOpen) return a0 + sf;
            }
            return carry_open;
        };
        auto park = [&](long long gS, long long gT) {
            if (gS < 0 || (gT - gS) * (long long)a.reso < (long long)a.repeat_length) return;
            if (lane == 0) {
                const int q = atomicAdd(&sm.runq_n[wid], 1);
                if (q < kRunQ) { sm.runq[(wid * kRunQ + q) * 2] = (int)(gS - a0); sm.runq[(wid * kRunQ + q) * 2 + 1] = (int)(gT - a0); }
                else emit_run(a, sm, nr, (int)(gS - a0), (int)(gT - a0));
            }
        };
        if (row_e > row_b) {
            if (hp_in && pclose >= 0) park(wid == 0 ? carry_open : run_start_before(wid - 1), a0 + pclose);
            const bool last_wave = (row_e == rows);           // this wave holds the last valid slot
            if (last_wave) {
                const long long open = hp ? ((S != kOpen) ? a0 + S : (wid == 0 ? carry_open : run_start_before(wid - 1))) : -1;
                if (last_chunk) { if (open >= 0) park(open, w_hi); }   // end of read closes the run (repeat.hpp:150)
                if (single_read && lane == 0) { sm.carry_open[cpar ^ 1] = last_chunk ? -1 : open; sm.carry_hp[cpar ^ 1] = (!last_chunk && open >= 0) ? 1 : 0; }
            }
        }
    }
    RAFT_STAMP(11);

    // 8. every run this wave parked becomes a repeat record, one lane per run, so their LDS look-ups overlap
    {
        const int nq = uni(sm.runq_n[wid]);
        if (nq > 0) {
            const int m = min(nq, kRunQ);
            int sS = 0, sT = 0, j = 0;
            if (lane < m) { sS = sm.runq[(wid * kRunQ + lane) * 2]; sT = sm.runq[(wid * kRunQ + lane) * 2 + 1]; }
            if (nr <= 64) {
                // owner of each run by one compare + ballot per run against the (register-held) read offsets
                const int ro = (lane < nr) ? sm.roff[lane] : 0x7fffffff;
#pragma unroll 1
                for (int q = 0; q < m; ++q) {
                    const int jq = __popcll(__ballot(ro <= __builtin_amdgcn_readlane(sS, q))) - 1;
                    if (lane == q) j = jq;
                }
            } else if (lane < m) j = owner_slot(sm, nr, sS);
            if (lane < m) emit_run_of(a, sm, j, sS, sT);
            if (lane == 0) sm.runq_n[wid] = 0;
        }
    }
    RAFT_STAMP(13);
    lds_barrier();
    RAFT_STAMP(14);
    // publish the repeat counts of the reads that are complete (rep_cnt[] was zeroed by the host)
    if (tid < nr && (!single_read || last_chunk)) {
        const int c = sm.rcnt[tid];
        if (c) a.rep_cnt[r_a + tid] = c;
    }
}

// unpacks a descriptor that lane l holds as dword l (l < 18) into scalars
__device__ __forceinline__ void unpack_desc(int raw, TileRegs &t)
{
    auto d = [&](int i) -> int { return __builtin_amdgcn_readlane(raw, i); };
    auto q = [&](int i) -> long long { return (long long)(((unsigned long long)(unsigned)d(i + 1) << 32) | (unsigned)d(i)); };
    t.r_lo = d(0); t.r_hi = d(1);
    t.seg_cum[0] = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) t.seg_cum[s + 1] = t.seg_cum[s] + d(2 + s);
    t.g_lo = q(2 + kMaxSeg); t.g_hi = q(4 + kMaxSeg);
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) t.seg_lo[s] = q(6 + kMaxSeg + 2 * s);
}

template <int THREADS, int CAP, int MINW, int U, bool DIAG>
__global__ __launch_bounds__(THREADS, MINW) void pileup_kernel(PileupArgs a)
{
    using Smem = PileupSmem<THREADS, CAP>;
    __shared__ __attribute__((aligned(16))) Smem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const long long nb = gridDim.x;
    if (uni(*(volatile int32_t *)a.err_flags) & kErrStop) return;   // (see kErrStop)
    const int32_t *td_words = reinterpret_cast<const int32_t *>(a.td);
    if (tid == 0) { sm.acc_cov = 0ull; sm.acc_rep = 0ull; sm.carry_open[0] = sm.carry_open[1] = -1; sm.carry_hp[0] = sm.carry_hp[1] = 0; }
    if (tid < Smem::NW) sm.runq_n[tid] = 0;

    // Descriptors travel as ONE VGPR (lane l holds dword l): raw_n = tile k+nb (landed), raw_nn = tile k+2nb (in
    // flight).  Only the current tile's descriptor is kept unpacked in scalars across the window code.
    auto desc_word = [&](long long tile) -> int {
        // 32-bit dword index (n_tiles * 18 < 2^31 is checked by the host): base in SGPRs + one 32-bit VGPR offset.
        // A per-lane 64-bit address gets hoisted out of the loop, spilled, and its scratch reload then waits on
        // vmcnt(0) -- behind every prefetch just issued and every coverage store still draining.
        const unsigned idx = (unsigned)tile * (unsigned)kDescDwords + (unsigned)lane;
        return (lane < kDescDwords) ? td_words[idx] : 0;
    };
    auto is_simple = [&](const TileRegs &t) -> bool {
        return (t.r_hi > t.r_lo) && (t.g_hi > t.g_lo) && (t.g_hi - t.g_lo <= CAP) && (t.r_hi - t.r_lo <= Smem::MAXR);
    };
    // items of this kernel: all tiles, or the tiles listed in slow_list
    const long long n_items = a.slow_list ? (long long)uni(*a.n_slow) : a.n_tiles;
    auto tile_at = [&](long long i) -> long long { return a.slow_list ? (long long)uni(a.slow_list[i]) : i; };
    // List mode (the tiles pileup_fast_kernel leaves): the items are heavy and very unequal -- reads longer than the LDS
    // window, tiles with hundreds of reads -- so they are handed out one at a time from a device counter and every one
    // takes the splitting path below with synchronous loads; one returning atomic and one descriptor fetch per item are
    // nothing against the item.  Otherwise: fixed stride over all tiles, descriptors and intervals prefetched.
    const bool dyn = a.slow_list != nullptr;
    auto draw = [&]() -> long long {
        lds_barrier();                          // every wave is done with the previous item (and with sm.item)
        if (tid == 0) sm.item = atomicAdd(a.slow_counter, 1);
        lds_barrier();
        return (long long)uni(sm.item);
    };
    long long k = dyn ? draw() : (long long)blockIdx.x;
    TileRegs cur{};
    Prefetch<U> g{}, gn{};
    bool simple = false, nsimple = false;
    int raw_n = 0, raw_nn = 0;
    if (k < n_items) {
        unpack_desc(desc_word(tile_at(k)), cur);
        simple = !dyn && is_simple(cur);
        if (simple) issue_prefetch<THREADS, U>(a, tid, cur.r_lo, cur.r_hi - cur.r_lo, cur.seg_lo, cur.seg_cum, g);
        if (!dyn && k + nb < n_items) raw_n = desc_word(tile_at(k + nb));
    }
    wait_all_loads(); // loop invariant: nothing is pending at the loop head on any incoming edge
    while (k < n_items) {
        const long long stamp_row = a.slow_list ? tile_at(k) : k;
        if (DIAG && tid == 0 && a.dbg) { sm.stamps[0] = __builtin_amdgcn_s_memtime(); sm.stamps[9] = __builtin_amdgcn_s_memrealtime(); }
        // next tile: its descriptor was requested one iteration ago; start its loads now
        nsimple = false;
        if (!dyn && k + nb < n_items) {
            const long long kn = k + nb;
            TileRegs nxt;
            unpack_desc(raw_n, nxt);
            nsimple = is_simple(nxt);
            if (nsimple) issue_prefetch<THREADS, U>(a, tid, nxt.r_lo, nxt.r_hi - nxt.r_lo, nxt.seg_lo, nxt.seg_cum, gn);
            if (kn + nb < n_items) raw_nn = desc_word(tile_at(kn + nb));
        }
        if (DIAG && tid == 0 && a.dbg) sm.stamps[8] = (unsigned long long)(cur.g_hi - cur.g_lo);
        RAFT_STAMP(1);

        if (simple) {
            pile_window<THREADS, CAP, U, DIAG>(a, sm, cur.r_lo, cur.r_hi, cur.g_lo, cur.g_hi, false, true, true, 0, cur.seg_lo,
                                            cur.seg_cum, g, stamp_row);
        } else if (cur.r_hi > cur.r_lo) {
            // A tile holding a read longer than the LDS window (or very many reads) is split on the fly:
            // sub-batches of whole reads, long reads in chunks of CAP windows; loads are issued synchronously.
            int r = cur.r_lo;
            long long chunk_pos = -1, g_first = 0, g_end = 0;
            int chunk_idx = 0;                  // chunks of the long read in hand so far (its parity picks the carry slot)
            for (;;) {
                int r_a, r_b;
                long long w_lo, w_hi;
                bool single, first, last;
                int cpar = 0;
                if (chunk_pos < 0) {
                    if (r >= cur.r_hi) break;
                    const long long gl = (r == cur.r_lo) ? cur.g_lo : uni(a.cov_off[r]);
                    // largest r2 in (r, r + MAXR] with all windows of reads [r, r2) inside one LDS window
                    const int hi_lim = min(cur.r_hi, r + Smem::MAXR);
                    const long long g_lim = (hi_lim == cur.r_hi) ? cur.g_hi : uni(a.cov_off[hi_lim]);
                    int r2;
                    if (g_lim - gl <= CAP) r2 = hi_lim;
                    else {
                        int lo = r, hi = hi_lim; // cov_off[lo]-gl <= CAP < cov_off[hi]-gl
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (uni(a.cov_off[mid]) - gl <= CAP) lo = mid; else hi = mid;
                        }
                        r2 = lo;
                    }
                    if (r2 > r) {
                        r_a = r; r_b = r2; w_lo = gl; w_hi = (r2 == cur.r_hi) ? cur.g_hi : uni(a.cov_off[r2]);
                        single = false; first = true; last = true;
                        r = r2;
                        // (only reads without windows: piled up all the same -- on an empty window -- because their records
                        // still have to be looked at: an interval on a read without windows is repeat.hpp:69-72's write past
                        // the vector, and a record of some other read here refutes the order the pass relies on)
                    } else {
                        g_first = gl; g_end = uni(a.cov_off[r + 1]); chunk_pos = gl; chunk_idx = 0;
                    }
                }
                if (chunk_pos >= 0) {
                    r_a = r; r_b = r + 1; w_lo = chunk_pos;
                    w_hi = (chunk_pos + CAP < g_end) ? chunk_pos + CAP : g_end;
                    single = true; first = (chunk_pos == g_first); last = (w_hi == g_end);
                    cpar = chunk_idx & 1; ++chunk_idx;
                    if (last) { chunk_pos = -1; r = r + 1; } else chunk_pos = w_hi;
                }
                long long s_lo[kMaxSeg];
                int s_cum[kMaxSeg + 1];
                s_cum[0] = 0;
#pragma unroll
                for (int s = 0; s < kMaxSeg; ++s) {
                    long long lo = cur.seg_lo[s], hi = cur.seg_lo[s] + (cur.seg_cum[s + 1] - cur.seg_cum[s]);
                    if (s < a.n_seg && !(r_a == cur.r_lo && r_b == cur.r_hi)) {
                        // The sub-ranges tile the tile's range: the first begins where the tile's does, the last ends where
                        // it ends, neighbours meet at the same search result.  Every record is therefore looked at by
                        // exactly one sub-batch, which flags it if it belongs to a read outside (kErrOrder).
                        const long long l2 = (r_a == cur.r_lo) ? lo : lower_bound_rid_uni(a.iv_rid, lo, hi, r_a);
                        long long h2 = (r_b == cur.r_hi) ? hi : lower_bound_rid_uni(a.iv_rid, lo, hi, r_b);
                        if (h2 < l2) { if (tid == 0) atomicOr(a.err_flags, kErrOrder); h2 = l2; }
                        hi = h2; lo = l2;
                    }
                    s_lo[s] = lo;
                    s_cum[s + 1] = s_cum[s] + (int)(hi - lo);
                }
                Prefetch<U> gs;
                issue_prefetch<THREADS, U>(a, tid, r_a, r_b - r_a, s_lo, s_cum, gs);
                pile_window<THREADS, CAP, U, DIAG>(a, sm, r_a, r_b, w_lo, w_hi, single, first, last, cpar, s_lo, s_cum, gs, stamp_row);
            }
            wait_all_loads(); // keep the "no load pending after a tile" invariant on this path too
        } else {
            wait_all_loads(); // tile without reads: same invariant (else the loop head waits behind the last stores)
        }
        RAFT_STAMP(7);
        if (DIAG && tid == 0 && a.dbg) {
            sm.stamps[10] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 16; ++i) a.dbg[stamp_row * 16 + i] = sm.stamps[i];
        }

        if (dyn) {
            k = draw();
            if (k < n_items) { unpack_desc(desc_word(tile_at(k)), cur); wait_all_loads(); }
        } else {
            k += nb; simple = nsimple; g = gn;
            if (k < n_items) unpack_desc(raw_n, cur);
            raw_n = raw_nn;
        }
    }
    lds_barrier();
    if (tid == 0) {
        a.block_sums[2 * (long long)blockIdx.x] = (long long)sm.acc_cov;
        a.block_sums[2 * (long long)blockIdx.x + 1] = (long long)sm.acc_rep;
    }
}

// the sorted runs of the record stream as its samples show them (bucket.hpp guess_runs_kernel)
struct GuessOut {
    int32_t n_desc, pad;
    long long desc_pos[kMaxSeg];
};

// What a pass that assumes a symmetric PAF (engine.hip run_pass, detecting contexts) still has to find: the mirror of
// record 0 (chop.hpp:171-184).  It can only sit among the records of record 0's target read, and where those lie in each
// sorted run is one more pair of the boundary searches tile_desc_kernel does anyway: the thread behind the closing
// boundary searches for "tile" [tid[0], tid[0] + 1) and its wave then looks at those few records.
struct MirrorArgs {
    const int32_t *qs, *qe, *tid, *ts, *te;   // the record columns besides the id column the kernel searches (tid == nullptr: no search)
    int32_t *found;                          // set to 1 when a record i > 0 mirrors record 0
};

// One thread per tile boundary: the descriptor the general pileup workgroups fetch (reads, windows, interval
// ranges) and, when `cuts` is given, the compact boundary record of pileup_fast_kernel plus the list of tiles that
// kernel leaves to the general one.  A tile's interval range ends where the next tile's begins, so each lane
// searches once per segment and takes the end from its neighbour lane (the last lane of a wave searches twice).
__global__ __launch_bounds__(256) void tile_desc_kernel(long long n_tiles, SegStarts sb, const long long *seg_end_dev,
                                                        const int32_t *iv_rid, const int32_t *tile_first,
                                                        const long long *cov_off, TileDesc *td, TileCut *cuts,
                                                        int fast_cap, int fast_max_reads, int32_t *slow_list,
                                                        int32_t *n_slow, const int32_t *samples, long long n_rec,
                                                        const long long *bucket_off, int32_t *err_flags, TileCut *extra,
                                                        int32_t *n_extra, int32_t extra_cap, int32_t piece_w, MirrorArgs mir,
                                                        GroupedOff grp, int recut_cap, const GuessOut *verify_guess)
{
    // fast_cap < 0: no tile is taken as it is -- every tile with reads is re-cut into entries of at most recut_cap windows and
    // fast_max_reads reads (pileup_wave.hpp: one wave per entry); otherwise recut_cap == fast_cap
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
    TileDesc d{};
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
    const int nr = d.r_hi - d.r_lo;
    const long long nwin = d.g_hi - d.g_lo;
    const bool fast = cuts && live && nr >= 1 && nr <= fast_max_reads && nwin > 0 && nwin <= fast_cap;
    if (live && !fast) td[k] = d;                   // the general kernel only looks at the tiles the fast one leaves
    if (cuts && edge) {
        TileCut c;
        c.r_lo = d.r_lo; c.flags = fast ? kCutFast : 0; c.g_lo = d.g_lo;
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) c.iv_lo[s] = (int32_t)d.iv_lo[s];
        cuts[k] = c;
    }
    // ---- a tile the fast kernel cannot take as it is becomes EXTRA tiles that it can (see PileupArgs::n_extra): its reads
    // are walked once; a read longer than the LDS window is cut into pieces of piece_w windows (each piece is handed ALL
    // intervals of the read and clips them), the reads between are grouped greedily into LDS-window-sized tiles.  The
    // interval ranges of the entries tile the tile's range: first entry begins where the tile's range begins, the last
    // ends where it ends, neighbours meet at the same search result (what the kernels' kErrOrder check relies on).
    const bool recut = extra && cuts && live && nr >= 1 && !fast;
    {
        long long t_lo[kMaxSeg], t_hi[kMaxSeg];
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) { t_lo[s] = d.iv_lo[s]; t_hi[s] = d.iv_lo[s] + d.n_iv[s]; }
        auto bound = [&](int s, int r) -> long long {      // first interval of read r in run s, inside the tile's range
            if (r <= d.r_lo) return t_lo[s];
            if (r >= d.r_hi) return t_hi[s];
            if (grp.off) return min(max(grp.at(s, r), t_lo[s]), t_hi[s]);
            return lower_bound_rid(iv_rid, t_lo[s], t_hi[s], r);
        };
        // (the interval bounds of a read are searched once: the pieces of a long read share them, and a group begins where
        // the entry before it ended)
        int32_t ib[kMaxSeg], ie[kMaxSeg];
        auto bounds_of = [&](int r, int32_t (&out)[kMaxSeg]) {
#pragma unroll
            for (int s = 0; s < kMaxSeg; ++s) out[s] = s < sb.n_seg ? (int32_t)bound(s, r) : 0;
        };
        auto emit = [&](int slot, int r_a, int r_b, long long g_a, long long g_b, int flags) {
            TileCut b{}, e{};
            b.r_lo = r_a; b.flags = flags; b.g_lo = g_a;
            e.r_lo = r_b; e.flags = 0; e.g_lo = g_b;
#pragma unroll
            for (int s = 0; s < kMaxSeg; ++s) { b.iv_lo[s] = ib[s]; e.iv_lo[s] = ie[s]; }
            extra[2 * (long long)slot] = b; extra[2 * (long long)slot + 1] = e;
        };
        // The tile's reads are walked twice: once to count the entries, once to write them -- in between the wave reserves
        // the room of all its tiles with ONE atomic (a returning atomic per entry, all on one word, cost the kernel two
        // thirds of its time on a long-read set: 195 us for 3e5 tiles).
        auto walk = [&](bool write, int slot) -> int {
            int n_e = 0;
            int r = d.r_lo;
            long long g_r = d.g_lo;                        // cov_off[r]
            if (write) bounds_of(r, ib);
            while (r < d.r_hi) {
                const long long g_n = (r + 1 == d.r_hi) ? d.g_hi : cov_off[r + 1];
                const long long nb = g_n - g_r;
                if (nb > recut_cap) {                      // pieces of one long read
                    const int P = (int)((nb + piece_w - 1) / piece_w);
                    if (write) {
                        bounds_of(r + 1, ie);
                        for (int q = 0; q < P; ++q) {
                            const long long w0 = (long long)q * piece_w, w1 = min(nb, w0 + piece_w);
                            emit(slot + n_e + q, r, r + 1, g_r + w0, g_r + w1, kCutFast | kCutPiece);
                        }
                    }
                    n_e += P;
                    ++r; g_r = g_n;
                } else {                                   // a group of whole reads that fits the window and the tables
                    int r2 = r + 1;
                    long long g2 = g_n;
                    while (r2 < d.r_hi && r2 - r < fast_max_reads) {
                        const long long g3 = (r2 + 1 == d.r_hi) ? d.g_hi : cov_off[r2 + 1];
                        if (g3 - g2 > recut_cap || g3 - g_r > recut_cap) break;   // a long read ends the group; so does a full window
                        ++r2; g2 = g3;
                    }
                    // (also a group of reads without windows: its records still have to be looked at)
                    if (write) { bounds_of(r2, ie); emit(slot + n_e, r, r2, g_r, g2, kCutFast); }
                    ++n_e;
                    r = r2; g_r = g2;
                }
                if (write) {
#pragma unroll
                    for (int s = 0; s < kMaxSeg; ++s) ib[s] = ie[s];   // the next entry begins where this one ended
                }
            }
            return n_e;
        };
        const int n_mine = recut ? walk(false, 0) : 0;
        const int incl = wave_incl_scan_add(n_mine);       // (every lane of the wave is here: no thread has left the kernel)
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(n_extra, total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + total > extra_cap) { if (lane == 0) atomicOr(err_flags, kErrExtra); }
            else if (recut) (void)walk(true, base + incl - n_mine);
        }
    }
    // tiles left to the general kernel: one append per wave (one atomic per tile on the same word serialises)
    const bool slow = cuts && live && nr >= 1 && !fast && !recut;
    const unsigned long long sm = __ballot(slow);
    if (sm) {
        const int leader = (int)__builtin_ctzll(sm);
        int base = 0;
        if (lane == leader) base = atomicAdd(n_slow, (int)__popcll(sm));
        base = __shfl(base, leader, kWave);
        if (slow) slow_list[base + (int)__popcll(sm & ((1ull << lane) - 1ull))] = (int32_t)k;
    }
}

} // namespace raft
