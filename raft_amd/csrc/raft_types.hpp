// raft_types.hpp -- what the engine's translation units share: the plain types the kernels' headers and the host code agree on
// (error flags, argument blocks, the control block's parts, the stamped hand-over).  No kernel is defined here: this header may be
// included by every translation unit; the headers that define kernels (pileup.hpp, bucket.hpp, finalize.hpp, pack.hpp, sort_pairs.hpp)
// are included by engine.hip alone, pileup_wave.hpp's instantiations by wave_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace raft {

constexpr int kMaxSeg = 4;   // sorted runs of the record stream the fast path accepts (hifiasm cis+trans = 2)

enum : int {
    kErrReadId = 1 << 0,
    kErrCoord = 1 << 1,
    kErrFragment = 1 << 2,
    kErrInternal = 1 << 3,
    kErrLen = 1 << 4,
    kErrExtra = 1 << 6,         // (rounds 1-3: the list of re-cut tiles overflowed; no kernel raises it any more)
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

// Boundaries of the quantum tiles (ranges of reads a worker of pileup_wave_kernel draws): boundary k says where range k begins --
// its first read, that read's first record in every sorted run, its first window; entry n_tiles closes the last range.
struct TileCut {
    int32_t r_lo;              // first read of tile k
    int32_t flags;             // (unused)
    int32_t iv_lo[kMaxSeg];    // first interval of tile k in segment s (absolute index, < 2^31 checked by the host)
    long long g_lo;            // first window of tile k in cov[]
};
static_assert(sizeof(TileCut) == 32, "a boundary is eight words, loaded by eight lanes");
enum : int {
    kCutPiece = 2,             // (a wave tile's own flag) the tile is a piece of ONE read longer than the LDS array
};

struct PileupArgs {
    // intervals: sorted by read id inside each of n_seg segments
    const int32_t *iv_rid, *iv_s, *iv_e;
    // ... or "window records" (pileup_wave.hpp IN = 1): one word per record, first window | one past the last << 16, no read
    // ids -- the reads' records are where the caller's offsets (grp) say
    const uint32_t *iv_w;
    GroupedOff grp;
    int32_t n_seg;
    // tiles and reads
    long long n_tiles;            // ranges tile_desc_kernel cut (boundaries 0 .. n_tiles)
    const int32_t *read_len;
    const long long *cov_off;   // [n_reads+1]
    int32_t n_reads;
    // params
    int32_t reso, high_cov, repeat_length, flank;
    uint32_t div_magic;           // n / reso == mulhi(n, div_magic) >> div_shift for 0 <= n < 2^31 (reso > 1)
    int32_t div_shift;            // -1: reso == 1
    // outputs
    int32_t *cov;
    // pileup_wave_kernel instantiated with OW = 1 or 2 writes the transfer encoding of cov[] instead (pack.hpp: OW bytes per
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
    int32_t *tile_counter;        // the ranges are handed out through these counters (pileup_wave.hpp next_range; zeroed before the pass)
    int32_t tile_batch;           // bits 24..28: counters in use - 1; bits 20..23: diagnostic switches (-DRAFT_WAVE_DIAG builds)
    int32_t piece_w;              // delta4: tile ids below this have slots of their own for the windows they list
    int32_t *slow_counter;        // delta4: the counter those tile ids are drawn from
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
constexpr int kNone = -1;


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


struct InspectOut {            // device words written by inspect_kernel
    int32_t sym_found;         // a record i >= 1 mirrors record 0 (chop.hpp:175-184)
    int32_t n_desc;            // positions i with qid[i] < qid[i-1]
    int32_t err_flags;
    int32_t pad;
    long long err_index;
    long long desc_pos[kMaxSeg]; // first kMaxSeg descent positions (unordered)
};


constexpr int kMaxRuns = 16;         // sorted runs a grouped pass is handed at most (more than kMaxSeg: merged into one on the device first)
constexpr int kErrWide = 1 << 10;    // general bucketing: a side whose windows do not fit 16 bits (bucket.hpp side_item)
constexpr int kCovDelta4 = 8;        // width code of the four-bit step encoding (pack.hpp; raft_hip_set_output_width, raft_hip_host_outputs::cov_width)
constexpr int kD4Block = 1024;       // ... windows per anchor
constexpr int kCtrStride = (4096 + 256) / 4;   // int32 words between two hand-out counters of the wave kernel: another 4 KiB block AND another 256-byte slot of it

constexpr int kErrDeep = 1 << 9;     // more tiles too deep for 16-bit coverage than the list for pileup_deep_kernel holds: the pass again, with room

struct DeepTile {                    // what this kernel knows about a tile when it decides not to pile it up (pileup_deep.hpp does)
    int32_t r_a, nr;                 // reads [r_a, r_a + nr)
    int32_t piece, nwin;             // kCutPiece: ONE read longer than a tile, this is a piece of it; windows of the tile
    long long g_lo;                  // first window of the tile in cov[]
    int32_t lo[kMaxSeg], cnt[kMaxSeg];   // its records: [lo, lo + cnt) of every sorted run
};


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

// Where the ranges of reads a worker of pileup_wave_kernel draws may begin: boundary k lies at a window position at(k), and
// tile_first[k] is the first read that begins at or behind it (bucket.hpp PrepPost / tile_first_kernel), its records and first
// window what tile_desc_kernel looks up.  Uniform: at(k) = k q.  Graded (round 6, sets with many tiles per worker): the set is
// eight shares of `share` windows -- what one of the kernel's eight hand-out counters deals out, in order -- and a share's boundaries
// lie q_long apart for its first n_long ranges and q_short apart behind them: few boundaries where a draw's length does not matter,
// short ranges where the kernel's end would otherwise wait for the last long draw (two / three / four / eight tiles' worth per
// draw, uniform: 2.42 / 2.45 / 2.49 / 2.56 ms) -- and tile_desc_kernel's searches, one per boundary and run, are the fixed part's
// largest item.  A share may close with a range shorter than q_short, and boundaries behind the set's last window own nothing.
struct Quantum {
    int32_t q;                            // uniform: boundaries q windows apart (share == 0)
    long long share;                      // graded: windows per share (a multiple of 128), boundaries per share, ...
    int32_t per_share, n_long, q_long, q_short;
    // the largest k with at(k) <= w, for 0 <= w
    __host__ __device__ long long idx(long long w) const
    {
        if (share == 0) return w < (1LL << 31) ? (long long)((unsigned)w / (unsigned)q) : w / q;
        int s = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) s += w >= (long long)i * share ? 1 : 0;
        if (w >= 8 * share) return 8LL * per_share;                       // (behind every share: the closing boundary)
        const unsigned x = (unsigned)(w - (long long)s * share);          // (share < 2^31: checked where the quantum is made)
        const unsigned in_long = (unsigned)n_long * (unsigned)q_long;
        const unsigned j = x < in_long ? x / (unsigned)q_long : (unsigned)n_long + (x - in_long) / (unsigned)q_short;
        return (long long)s * per_share + (long long)(j < (unsigned)per_share ? j : (unsigned)per_share - 1u);
    }
    __host__ __device__ long long n_ranges(long long n_windows) const { return share == 0 ? n_windows / q + 1 : 8LL * per_share; }
};
inline Quantum uniform_quantum(int q) { Quantum z{}; z.q = q; return z; }
// eight shares of ceil(B / 8) windows (rounded up to 128): `frac_long` of a share in ranges of q_long, the rest in ranges of q_short
inline Quantum graded_quantum(long long n_windows, int q_long, int q_short, double frac_long)
{
    Quantum z{};
    z.share = ((n_windows + 7) / 8 + 127) / 128 * 128;
    z.q_long = q_long; z.q_short = q_short;
    z.n_long = (int32_t)((double)z.share * frac_long / q_long);
    const long long rest = z.share - (long long)z.n_long * q_long;
    z.per_share = z.n_long + (int32_t)((rest + q_short - 1) / q_short);
    z.q = q_short;
    return z;
}

struct FinalizeArgs {
    int32_t n_reads;
    const int32_t *read_len;
    const long long *rep_res_off;
    const int32_t *rep_cnt;
    int32_t *raw_key, *raw_s, *raw_e;     // sorted in place by finalize_count_kernel
    int32_t interval_length, div, overlap_length;
    FastDiv by_L, by_div, by_reso;        // interval_length, div, reso as divisors
    // reads with more than long_windows windows were piled up in pieces (pileup_wave.hpp: a tile with kCutPiece): their raw
    // records are unflanked [start, end) runs per piece, to be joined, tested, flanked and clamped here
    int32_t long_windows, reso, repeat_length, flank;
    int32_t *rep_cnt_rw;                  // (rep_cnt, writable: the joined count replaces the pieces' count)
    unsigned long long *total_repeat;     // repeat.hpp:127,152 for those reads
    int32_t *cut_cnt, *frag_cnt;          // [n_reads]
    const long long *rep_off, *cut_off, *frag_off; // [n_reads+1] (fill kernel)
    int32_t *rep_s, *rep_e, *cuts, *frag_read, *frag_begin, *frag_end;
    int32_t *err_flags;
    long long *err_index;
    // the tail's offsets without a scan of their own (round 6): finalize_count_kernel leaves every workgroup's sums of (repeats, cut
    // points, fragments, read length) in tail_part[]; tail_prefix_kernel -- ONE workgroup -- turns them into where every workgroup's
    // reads begin, and into the totals the host is handed; finalize_fill_kernel scans its 256 reads in the workgroup and writes
    // rep_off / cut_off / frag_off itself; publish_ctrl_kernel hands the control block over.  count -> prefix -> fill -> publish:
    // four launches, the middle one over N / 256 words, where there were five with two passes over the reads' counts and a third over
    // their lengths (count, scan_partials, scan_apply, fill, totals).
    // (Measured and dropped on the way, profiles/r06_tail_parts.txt: TWO launches, what crosses workgroups inside a kernel travelling in
    // device-scope atomics -- per wave: non-returning adds into shared sums cost the count kernel 35 us of a pass over 3.3 M reads, a
    // returning one per wave the fill kernel 55 us; per workgroup of 1024 reads: fill 206 us against 100, its sixteen waves waiting
    // for each other at the barrier their common sums need.)
    long long *tail_part;                 // [4][tail_blocks]
    long long *tail_prefix;               // [3][tail_blocks]
    int32_t tail_blocks;
    long long *rep_off_w, *cut_off_w, *frag_off_w;   // (the offset arrays, writable)
};


// What the pass hands the host at its end: the sums of the pileup kernel's workers, the control block, the page-locked block the
// host looks at.
struct TailPublish {
    long long n_tiles;                    // workers of the pileup kernel(s): two sums each in tile_sums
    const long long *tile_sums;
    unsigned long long *totals;           // Ctrl::totals: coverage, repeat bp, read length
    const long long *bucket_off;          // general bucketing: its offsets (the true interval count at [n_reads]), else nullptr
    long long *tails;                     // Ctrl::out_totals
    const long long *ctrl_words;
    int n_ctrl_words;
    long long *host_block;
    long long pass_seq;
};


// What the host reads back without the runtime's wait travels in STAMPED LINES: 64 bytes of the context's page-locked block hold six
// 8-byte data words and, in words 3 and 7, the number of the hand-over; one store instruction of the wave writes all lines, and a
// line (at the least each 32-byte half of it) arrives whole.  The host takes a line when both stamps are the number it waits for.
// (Round 4's form -- the words, a system-scope fence, then the number in a word of its own -- is NOT safe: stores to host memory
// are posted writes that may pass each other.  tools/r05/s27.sh: the sizes hand-over read that way gave a stale window count in
// 1 of 25 runs of the parity tests; none in 25 with the runtime's wait.)
constexpr int kStampData = 6;
__host__ __device__ constexpr int stamped_lines(int n_words) { return (n_words + kStampData - 1) / kStampData; }
// lane t of one wave (0 .. 63) stores its word of the stamped form of src(0 .. n_words - 1); n_words <= 48
template <class Src>
__device__ __forceinline__ void publish_stamped(long long *host, Src src, int n_words, long long seq, int t)
{
    const int line = t >> 3, pos = t & 7;
    long long v = seq;
    if ((pos & 3) != 3) {
        const int idx = line * kStampData + (pos < 3 ? pos : pos - 1);
        v = idx < n_words ? src(idx) : 0;
    }
    if (line < stamped_lines(n_words)) host[t] = v;
}
inline bool stamped_seen(const volatile long long *h, int n_words, long long seq)
{
    for (int line = 0; line < stamped_lines(n_words); ++line)
        if (h[line * 8 + 3] != seq || h[line * 8 + 7] != seq) return false;
    return true;
}
inline void unstamp(const volatile long long *h, int n_words, long long *out)
{
    for (int i = 0; i < n_words; ++i) { const int pos = i % kStampData; out[i] = h[(i / kStampData) * 8 + (pos < 3 ? pos : pos + 1)]; }
}

} // namespace raft
