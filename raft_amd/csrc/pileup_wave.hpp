// pileup_wave.hpp -- the pileup kernel of round 4: one WAVE per tile, a 16-bit difference array, eight windows per lane.
//
// What the counters and stamps of round 3's workgroup-tile kernel said (docs/history.md §5): its time does not follow its bytes -- 53 % of
// the wave cycles are waits, the vector pipe is a quarter busy, and what a tile costs is its dependent chains (LDS read ->
// adds -> six DPP steps -> carry; three workgroup barriers; a pass over the array just to hand every wave its start value)
// executed by four waves per SIMD.  This kernel removes the chains' causes instead of shortening them:
//   * a tile belongs to ONE wave.  No barrier, no pass A (the carry is a scalar that runs along the wave's rows), no seams
//     between waves, no double-buffered tables; waves of a SIMD are in unrelated phases of unrelated tiles, so there is
//     always one that can issue.  A worker streams through a contiguous range of reads and cuts its tiles itself: the longest
//     run of whole reads whose windows fit the array (a ballot over the reads' offsets), their records by a ballot over the id
//     slots that have landed; a read longer than the array goes in pieces (joined by finalize_count_kernel).  The only thing
//     cut in advance are the boundaries of quantum tiles four times a tile's size (tile_desc_kernel): where ranges may begin.
//   * the difference array holds 16 bits per window, two windows per LDS dword.  +1 / -1 land as ds_add_u32 of
//     +-1 or +-65536 on the window's dword; the low halves carry a bias of 0x8000 so that they never borrow from the high
//     ones.  A tile's LDS footprint is a quarter of the int32 window's, which is what lets up to eight waves share a SIMD.
//   * a lane owns eight windows of every 512-window row -- four in each half-row, so that both of the wave's stores are
//     contiguous -- and the prefix sum is packed arithmetic: with P = (L, H) the dword without its bias, P + (P << 16) has
//     L in its low and L + H in its high half; two more packed adds give the lane's prefix, the lanes' totals of BOTH
//     half-rows travel through ONE DPP scan as one 32-bit word, and one packed add per dword puts the start value in.
//     ~22 vector instructions per 512 windows where the int32 rows took 2 x 20 per 512, half the scans, no carry hand-over.
//   * loads: the next tile's (per-read table, record slots) go out at the top of an iteration and are waited for once, before
//     the iteration's first coverage store (vmcnt counts loads and stores in one in-order queue: a wait behind the stores
//     would wait for the stores); the stores are raw buffer stores of whole 16-byte lanes (the compiler merged plain ones
//     with the tile edges' element stores into 12 + 4 bytes).
//   * bound: every intermediate is exact modulo 2^16 and every coverage value must be below 32768.  A tile holds fewer
//     intervals than that, or it is left to pileup_deep_kernel (pileup_deep.hpp: 32-bit, a workgroup per tile, in the same pass).
// Reference semantics: repeat.hpp:28-79 (profileCoverage), repeat.hpp:111-168 (run scan) -- see pileup.hpp; the run scan
// of a half-row is a per-lane scan of four slots (one wave shift + one max-scan), as in round 3's kernel.
#pragma once
#include "pileup.hpp"

namespace raft {

// The empty difference array: the LOW half of every dword is biased by 0x8000, so that a -1 landing on an even slot never
// borrows from the odd slot above it (the ds_add is a 32-bit add); one xor per dword takes the bias off again.
constexpr uint32_t kZero = 0x00008000u;
// (kCtrStride: raft_types.hpp)
constexpr int kWaveMaxReads = 63;    // reads per wave tile: lane j <-> read r_a + j, entry nr closes the table (nr + 1 <= 64)

template <int SLOTS>
struct WaveSmem {                    // ONE wave's LDS
    static_assert(SLOTS % 512 == 0, "rows of 512 slots");
    uint32_t diff[SLOTS / 2];        // two 16-bit slots per dword, slot order; all kZero between tiles
    // ONE array for the two per-read tables, so that what the record pass relies on is in the type: entry 64 -- read for a record that
    // is not the tile's (j == nr == 63), value unused -- exists, and the unmasked pair (roff[j], roff[j + 1]) is one ds_read2
    int32_t rtab[128];               // [0, 64): roff, first slot of read r_a + j relative to a0 (j <= nr); [64, 128): rcnt, raw repeats emitted for the read in this tile
    int32_t runq[2 * kRunQ];         // closed runs parked for emission
};

struct WaveTile {                    // scalars of one tile (SGPRs)
    int r_a, nr, nwin, piece, more, n_total;
    int lo[kMaxSeg], cnt[kMaxSeg];
    long long g_lo;
};

// v_pk_add_u16 with one half of `b` added to BOTH halves of `a` (VOP3P op_sel: free).  HI: b's high half, else its low half.
template <bool HI>
__device__ __forceinline__ unsigned pk_add_bcast(unsigned a, unsigned b)
{
    unsigned r;
    if (HI) asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
    else    asm("v_pk_add_u16 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_add_u16(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int wave_shl1(int v, int last)      // lane l takes lane l + 1's value, lane 63 keeps `last`
{
    return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xf, 0xf, false);         // wave_shl:1 (GFX9 family)
}

// The wave's coverage stores as the instructions they are meant to be: raw buffer stores (a per-tile descriptor, the lane's
// 32-bit byte offset).  (Left to the compiler, the 16-byte store of a row's lanes and the element stores of a tile's edge
// lanes were merged into a 12-byte plus a 4-byte store per half-row: twice the store instructions, none of them a whole
// 16 bytes -- the int32 pass took 3.2 ms where the byte pass took 1.9.  Inline asm stores are not an option either: the
// compiler does not pad the wait states behind them and overwrote their data registers.)
// The int32 coverage stores are NON-TEMPORAL (aux bit 1, `nt`): the kernel never reads a coverage line again, and lines stored that
// way are the first to leave the L2 -- the record columns and per-read tables that neighbouring tiles share stay a little longer.
// Three processes each, `profiles/r05_store_policy_ab.txt`: ultralong columns 2.65 -> 2.58 ms every time, headline 2.04-2.07 where plain
// stores gave 2.05-2.22, the pass 1.5 % shorter.  Write-through (`sc1`: the line is dropped from the L2 at once) fetched 8 % less
// and took 12 % longer; the packed encodings (a quarter / an eighth of the bytes) gain nothing from `nt` and stay plain.
template <int OW> struct CovAux { static constexpr int v = OW == 4 ? 2 : 0; };
// (the record slots' loads stay plain: marked non-temporal like the coverage stores, the column form ran 2-8 % slower in one process --
// 2.175 / 2.197 against 2.136 / 2.031 ms, three contexts each, tools/lib_ab.py -- and the window-record form the same, 1.571 / 1.567)
constexpr int kRecAux = 0;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

template <int U>
struct WaveRegs { int rid[U], st[U], en[U]; };
struct WaveReadRegs { int cv, rr, rl; int so[2]; int n; };      // n: lanes whose table entries were asked for (wave-uniform)

// NSEG sorted runs, U prefetch slots per lane (slot u: record (u / NSEG) * 64 + lane of run u % NSEG), OW bytes per window
// written (4: int32 cov[]; 1 / 2: the transfer encoding; 8: four-bit steps, pack.hpp), IN = 1: window records.
template <int SLOTS, int NSEG, int U, int OW, int IN>
__device__ __forceinline__ void wave_tile_loop(WaveSmem<SLOTS> &sm, const TileCut *__restrict__ cuts, const PileupArgs &a, int wave_id, int n_waves)
{
    constexpr int ITER = U / NSEG;
    static_assert(U % NSEG == 0 && ITER >= 1, "slots are split evenly over the runs");
    static_assert(IN == 0 || NSEG <= 2, "window records: one or two runs");
    constexpr bool D4 = OW == 8;
    const int lane = (int)(threadIdx.x & 63u);
    // (every worker owns two words of the sums tail_prefix_kernel adds up: also the ones that leave without a tile)
    auto leave_empty = [&]() { if (lane == 0) { a.block_sums[2 * (long long)wave_id] = 0; a.block_sums[2 * (long long)wave_id + 1] = 0; } };
    if (uni(*(volatile int32_t *)a.err_flags) & (kErrExtra | kErrStop)) { leave_empty(); return; }
    const int n_seg_tiles = (int)a.n_tiles;      // segments (quantum tiles) tile_desc_kernel cut: boundaries cuts[0 .. n_seg_tiles]
    constexpr int CAP = SLOTS - 4;               // windows of a tile: 3 alignment slots + CAP + 1 sentinel
    const unsigned win_m1 = a.div_shift < 0 ? ~0u : 0u;
    const int win_sh = a.div_shift < 0 ? 0 : a.div_shift;
    constexpr unsigned kLimit = OW == 1 ? 255u : 65535u;
    // threshold test on packed halves: c + (0x8000 - high_cov) has bit 15 set <=> c >= high_cov (0 <= c < 32768; a threshold
    // above 32767 can never be reached by a 16-bit tile, one below 1 is reached by every window: both clamp)
    const unsigned hc16 = (unsigned)min(max(a.high_cov, 0), 0x8000);
    const unsigned kthr = ((0x8000u - hc16) & 0xffffu) * 0x10001u;

    for (int i = lane; i < SLOTS / 2; i += 64) sm.diff[i] = kZero;
    long long lane_cov = 0, lane_rep = 0;

    // ---- A worker STREAMS through contiguous ranges of reads and cuts its tiles itself (round 4, second version).
    // The first version walked a list of tiles tile_desc_kernel had cut in advance -- a sequential walk per quantum tile plus,
    // for six-column input, four bisections of the record stream per tile: 0.3 ms at human scale, and 80 MB of cut records
    // written and read back.  None of that is needed: a range begins at a boundary tile_desc_kernel found anyway (first read,
    // first record of every run, first window); from there the NEXT tile is the longest run of whole reads whose windows fit the
    // LDS array (a ballot over the reads' offsets, which the worker loads for its per-read table in any case) and its records
    // are the ones whose read id is below the tile's last read -- a ballot over the record slots that have already landed,
    // since they are what the tile's interval phase is about to consume.  A read longer than the array is taken in pieces;
    // a tile with more records than slots finds its end with two 64-way probes of the id column.
    const int32_t *cut_words = reinterpret_cast<const int32_t *>(cuts);
    struct Start { int r, q; int pos[NSEG]; long long g; };   // a tile's first read, piece index, first record of every run, first window of the READ
    int R_end = 0, pend[NSEG];                    // the current range: one past its last read, one past its last record of every run
#pragma unroll
    for (int s = 0; s < NSEG; ++s) pend[s] = 0;
    auto load_range = [&](int k0, int k1, Start &t) {   // synchronous: once per range
        const int w0 = lane < 8 ? cut_words[(unsigned)k0 * 8u + (unsigned)lane] : 0;
        const int w1 = lane < 8 ? cut_words[(unsigned)k1 * 8u + (unsigned)lane] : 0;
        wait_all_loads();
        t.r = __builtin_amdgcn_readlane(w0, 0); t.q = 0;
        t.g = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(w0, 7) << 32) | (unsigned)__builtin_amdgcn_readlane(w0, 6));
        R_end = __builtin_amdgcn_readlane(w1, 0);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) { t.pos[s] = __builtin_amdgcn_readlane(w0, 2 + s); pend[s] = __builtin_amdgcn_readlane(w1, 2 + s); }
    };
    // loads of the tile that begins at `t`: the per-read table (64 reads from t.r on) and U record slots from t.pos on
    // (`tables`: the per-read entries of reads t.r .. t.r + 63 are loaded; not when the tile before holds them already -- a tile
    // of long reads uses a handful of the 64, and every line a tile asks for comes from HBM again: the coverage stores turn
    // the L2 over several times per tile.  The next tile then takes the entries by a shift across the lanes, see `reuse`)
    auto issue = [&](const Start &t, WaveRegs<U> &g, WaveReadRegs &rd, bool tables) {
        rd.cv = 0; rd.rr = 0; rd.rl = 0; rd.so[0] = 0; rd.so[1] = 0; rd.n = 64;
        const int idx = t.r + lane;
        if (tables && idx <= a.n_reads) {
            rd.cv = reinterpret_cast<const int32_t *>(a.cov_off + t.r)[2 * lane];
            rd.rr = reinterpret_cast<const int32_t *>(a.rep_res_off + t.r)[2 * lane];
            if (IN == 1) {
#pragma unroll
                for (int s = 0; s < NSEG; ++s) rd.so[s] = reinterpret_cast<const int32_t *>(a.grp.off + s * a.grp.stride + t.r)[2 * lane];
            }
        }
        if (tables && idx < a.n_reads) rd.rl = a.read_len[idx];
        // The record slots: raw buffer loads from a descriptor per column and run that begins at the tile's first record and ends
        // with the range -- the lane's part of the address is a constant (lane * 4 + the slot's 256 bytes), the tile's part is in the
        // descriptor, and so is the end of the range (a lane beyond it gets 0 and touches no memory; every use of a slot is masked
        // by the tile's count).  As plain loads a slot cost a 64-bit add and compare, three 64-bit address computations, three
        // selects and an exec mask -- ~70 of a tile's ~1600 vector instructions, in a kernel whose vector pipe is busy two thirds of
        // the time (SQ counters, round 5).  (No scalar offset: the hardware's range check does not look at it.)
#ifdef RAFT_W_NOBUF
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = u % NSEG, first = (u / NSEG) * 64;
            const long long at = (long long)t.pos[s] + first + lane;
            if (IN == 1) {
                g.st[u] = 0;
                if (at < pend[s]) g.st[u] = (int)a.iv_w[at];
            } else {
                g.rid[u] = 0x7fffffff; g.st[u] = 0; g.en[u] = 0;
                if (at < pend[s]) { g.rid[u] = a.iv_rid[at]; g.st[u] = a.iv_s[at]; g.en[u] = a.iv_e[at]; }
            }
        }
        return;
#endif
        const int lane4 = lane * 4;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int p0 = uni(t.pos[s]);        // (uniform by construction; said again so that the descriptors are built in scalar registers)
            const unsigned bytes = (unsigned)uni(min(max(pend[s] - p0, 0), 64 * ITER)) * 4u;    // (the clamp is a vector instruction)
            if (IN == 1) {
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_w + p0), 0, bytes, 0x00020000);
#pragma unroll
                for (int it = 0; it < ITER; ++it) g.st[it * NSEG + s] = __builtin_amdgcn_raw_buffer_load_b32(rw, lane4 + it * 256, 0, kRecAux);
            } else {
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_rid + p0), 0, bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_s + p0), 0, bytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_e + p0), 0, bytes, 0x00020000);
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    g.rid[it * NSEG + s] = __builtin_amdgcn_raw_buffer_load_b32(rr, lane4 + it * 256, 0, kRecAux);
                    g.st[it * NSEG + s] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane4 + it * 256, 0, kRecAux);
                    g.en[it * NSEG + s] = __builtin_amdgcn_raw_buffer_load_b32(re, lane4 + it * 256, 0, kRecAux);
                }
            }
        }
    };

    // Ranges: a worker draws ONE segment at a time (tile costs differ by what a tile holds and CUs are not equally fast), and it
    // draws it from the counter of its XCD: workgroups go to the XCDs round-robin, so with kCtr a multiple of 8 the workers that
    // share counter c = wave_id % kCtr share an L2, and each counter hands out a contiguous share of the segments.  Neighbouring
    // tiles -- the lines of the per-read tables and of the record columns they both touch, the coverage line their common edge
    // lies in -- then meet in ONE L2 (measured inside one process, tools/mode_probe.py: 2.26 ms against 2.50 with the segments
    // dealt round-robin to all workers; kernel times of separate processes differ by as much with where their buffers
    // happen to lie).  Returning atomics on ONE word serialise at ~12 ns each -- a draw per four tiles from one counter WAS
    // the first version's duration; the counters are 4 KiB + 256 bytes apart (256 bytes apart, the eight of
    // them shared a memory channel or not depending on where the block lay: contexts of ONE process ran at 2.2 or at 2.6 ms).  A worker whose share is used up helps with the next.
    const int kCtr = ((a.tile_batch >> 24) & 31) + 1;
#ifdef RAFT_WAVE_DIAG      // (make DEFS=-DRAFT_WAVE_DIAG, RAFT_WAVE_MODE=<bits>, tools/mode_probe.py: parts of the kernel switched off at run time)
    const int kMode = (a.tile_batch >> 20) & 15;
#else
    constexpr int kMode = 0;
#endif
    typedef __attribute__((address_space(1))) int32_t *global_i32_ptr;
    global_i32_ptr draw_from = (global_i32_ptr)a.tile_counter;
    asm volatile("" : "+v"(draw_from));
    // (which XCD: HW_REG_XCC_ID, id 20, bits 3:0 -- workgroups go to the XCDs round-robin, but which one the first gets is not fixed)
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
    const int per_xcc = max(1, kCtr / 8);
    int ctr = kCtr >= 8 ? xcc * per_xcc + (wave_id / 8) % per_xcc : wave_id % kCtr, ctr_tried = 0;
    // counter c hands out segments [c * q + min(c, r), ...) with q, r = n_seg_tiles divmod kCtr: one 32-bit division per worker
    // (the 64-bit n * c / kCtr this replaces was 800 scalar instructions in every draw)
    const int share_q = (int)((unsigned)n_seg_tiles / (unsigned)kCtr), share_r = n_seg_tiles - share_q * kCtr;
    auto next_range = [&](Start &t) -> bool {     // synchronous (a draw, two boundary records)
        for (;;) {
            if (ctr_tried == kCtr) return false;
            int drawn = 0;
            if (lane == 0) drawn = __hip_atomic_fetch_add(draw_from + ctr * kCtrStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wait_all_loads();
            const int d0 = ctr * share_q + min(ctr, share_r), d1 = d0 + share_q + (ctr < share_r ? 1 : 0);
            int k0 = d0 + uni(drawn);
            if (k0 >= d1) k0 = n_seg_tiles;
            if (k0 >= n_seg_tiles) { ctr = ctr + 1 == kCtr ? 0 : ctr + 1; ++ctr_tried; continue; }
            const int k1 = k0 + 1;
            load_range(k0, k1, t);
            bool left = false;                   // (a range without reads must not own records: they would be walked by nobody)
#pragma unroll
            for (int s = 0; s < NSEG; ++s) left |= t.pos[s] != pend[s];
            if (t.r < R_end) return true;
            if (left && lane == 0) atomicOr(a.err_flags, kErrOrder);
        }
    };
    // delta4: a tile's listed windows go to slots of its own, named by a tile id; ids come in blocks of 32 from a second counter
    int d4_id = 0, d4_id_end = 0;
    auto next_d4_id = [&]() -> int {
        if (d4_id == d4_id_end) {
            int drawn = 0;
            if (lane == 0) drawn = atomicAdd(a.slow_counter, 32);
            d4_id = uni(drawn); d4_id_end = d4_id + 32;
        }
        return d4_id++;
    };
    (void)next_d4_id;

    Start ts, nts;
    if (!next_range(ts)) { leave_empty(); return; }
    WaveTile cur;
    WaveRegs<U> g, gn;
    WaveReadRegs rd, rdn;
    issue(ts, g, rd, true);
    wait_all_loads();

    while (true) {
        // ---- the current tile: its loads have landed.  Which reads fit?  (the reads' first windows are non-decreasing and so is
        // "belongs to this range": the ballot is a prefix of the lanes, lane 0 -- the tile's first read -- always in it)
        const int cv0 = uni(rd.cv);
        const int rel = rd.cv - cv0;
        const unsigned long long okm = __ballot(ts.r + lane <= R_end && (unsigned)rel <= (unsigned)CAP && lane < rd.n);
        int nr = (int)__popcll(okm) - 1;
        int piece = 0, n_pieces = 1, nb_read = 0;
        cur.r_a = ts.r;
        if (nr <= 0) {                           // the first read alone is longer than the array: this tile is piece ts.q of it
            nb_read = __builtin_amdgcn_readlane(rel, 1);
            n_pieces = (nb_read + CAP - 1) / CAP;
            piece = kCutPiece; nr = 1;
            cur.nwin = min(nb_read - ts.q * CAP, CAP);
            cur.g_lo = ts.g + (long long)ts.q * CAP;
        } else {
            cur.nwin = __builtin_amdgcn_readlane(rel, nr);
            cur.g_lo = ts.g;
        }
        // ... and which records are theirs?  The ids are sorted inside a run, so "id below the tile's last read" is a prefix of
        // the slots' lanes; a slot that is full and all of the tile's passes the question on to the next one.  A run whose
        // slots are ALL the tile's (the tile has more records than slots) shrinks the tile to the reads that are complete in the
        // slots -- dense stretches get smaller tiles instead of a wait for more records -- unless that leaves no read at all:
        // then the end is found with two 64-way probes of the id column and the records behind the slots are streamed.
        int r_b = ts.r + nr;
        bool open_run[NSEG];
        auto count_run = [&](int s2, int rb) -> int {      // records of run s2 in the slots whose id is below rb; open_run: all of them and more to come
            const int avail = pend[s2] - ts.pos[s2];
            int c = 0;
            bool open = true;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int L = min(max(avail - 64 * it, 0), 64);
                if (open) {
                    const int n = (int)__popcll(__ballot(lane < L && g.rid[it * NSEG + s2] < rb));
                    c += n;
                    if (n < 64) open = false;
                }
            }
            open_run[s2] = open && avail > 64 * ITER;
            return c;
        };
        cur.more = 0; cur.n_total = 0;
        bool defer = false;
        int avail_s[NSEG];
#pragma unroll
        for (int s = 0; s < NSEG; ++s) { avail_s[s] = pend[s] - ts.pos[s]; open_run[s] = false; }
        if constexpr (IN == 0) {
            int rb_fit = r_b;
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                cur.cnt[s] = count_run(s, r_b);
                if (open_run[s]) rb_fit = min(rb_fit, __builtin_amdgcn_readlane(g.rid[(ITER - 1) * NSEG + s], 63));
            }
            if (rb_fit < r_b && rb_fit > ts.r && !piece) {   // fewer reads, all of whose records are in the slots
                nr = rb_fit - ts.r; r_b = rb_fit;
                cur.nwin = __builtin_amdgcn_readlane(rel, nr);
#pragma unroll
                for (int s = 0; s < NSEG; ++s) cur.cnt[s] = count_run(s, r_b);
            }
        }
        cur.nr = nr; cur.piece = piece;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            cur.lo[s] = ts.pos[s];
            const int avail = pend[s] - ts.pos[s];
            int c = 0;
            if constexpr (IN == 1) {
                c = __builtin_amdgcn_readlane(rd.so[s], nr) - __builtin_amdgcn_readlane(rd.so[s], 0);
                c = min(max(c, 0), avail);       // (offsets that disagree with the range: the order check below refutes the pass)
            } else {
                c = cur.cnt[s];
                // (an open run: every slot is the tile's and the run goes on -- the interval phase streams on until an id says
                // stop, and only then is the next tile's first record known: its loads go out late, see `defer`)
                if (open_run[s]) { cur.more = 1; defer = true; }
            }
            cur.cnt[s] = c;
            cur.n_total += c;
            if (c > 64 * ITER) cur.more = 1;
        }
        // ---- where the next tile begins; the end of the range.  Normally planned here, so that the next tile's loads go out ahead
        // of the interval phase that hides their latency; a tile with an open run plans it behind its interval phase.
        const bool last_piece = !piece || ts.q + 1 == n_pieces;
        bool have_next = true;
        // the next tile's table entries: the ones this tile holds, `shift` lanes up, when they reach far enough for it (twice this
        // tile's reads + 8, at least 16 -- a tile that could have held more reads than it has entries for is cut short)
        bool reuse = false;
        int shift = 0;
        auto plan_next = [&]() {
            if (!last_piece) { nts = ts; nts.q = ts.q + 1; reuse = true; shift = 0; }
            else {
                shift = nr;
                reuse = !(kMode & 1) && r_b != R_end && rd.n - nr >= min(64, max(16, 2 * nr + 8));
                nts.r = r_b; nts.q = 0; nts.g = ts.g + (piece ? nb_read : cur.nwin);
#pragma unroll
                for (int s = 0; s < NSEG; ++s) nts.pos[s] = ts.pos[s] + cur.cnt[s];
                if (r_b == R_end) {
                    // the range is done: every record of it has been handed to a tile, or the stream is not what the pass assumed
                    bool left = false;
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) left |= nts.pos[s] != pend[s];
                    if (left && lane == 0) atomicOr(a.err_flags, kErrOrder);
                    have_next = next_range(nts);
                }
            }
            if (have_next) issue(nts, gn, rdn, !reuse);
        };
        if (!last_piece) defer = false;          // (the next piece begins where this one does: nothing to wait for)
        if (!defer) plan_next();
        int tile_id = 0;
        if (D4) tile_id = next_d4_id();
        (void)tile_id;

        const int r_a = cur.r_a;
        const long long a0 = cur.g_lo & ~3LL;
        const int off0 = (int)(cur.g_lo - a0);
        const int t_end = off0 + cur.nwin;
        const int rows_all = (t_end + 1 + 511) >> 9;
        // ---- per-read table of this tile
        const int ro = rd.cv - (int)a0;                     // 32-bit wrap-around is exact
        if (lane <= nr) sm.rtab[lane] = ro;
        sm.rtab[64 + lane] = 0;
        const int ro_s = (lane < nr) ? ro : 0x7fffffff;     // first slots of the tile's reads, for the run scan and the owner search

        // ---- 1. intervals -> +1 / -1 on 16-bit slots
        int covsum = 0;
        // (as lane masks in scalar registers: as per-lane flags carried across the slots' branches they were five vector
        // instructions a slot)
        unsigned long long bad_any = 0ull, bad_order = 0ull;
        auto add_pm = [&](int pf, int pl1) {
            const unsigned vp = 1u << ((pf & 1) << 4), vm = 0u - (1u << ((pl1 & 1) << 4));
            if (kMode & 4) { covsum += pl1 - pf + (int)vp + (int)vm; return; }      // (diagnostic: no scatter)
            __hip_atomic_fetch_add(&sm.diff[pf >> 1], vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sm.diff[pl1 >> 1], vm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            covsum += pl1 - pf;
        };
        if constexpr (IN == 0) {
            auto win = [&](unsigned n) -> int { return (int)(((n & win_m1) | __umulhi(n, a.div_magic)) >> win_sh); };
            // (`mine`: the slot's lane holds a record of this tile -- the lanes behind the tile's count hold the next tile's)
            auto one = [&](int rid, int st, int en, bool mine) {
                const unsigned jr = (unsigned)(rid - r_a);
                const unsigned j = min(jr, (unsigned)nr);
                // (j + 1 <= 64: entry 64 of the table -- the first repeat counter -- is read only for a record that is not the tile's,
                // j == nr == 63, whose values are not used; unmasked, the two entries are one ds_read2)
                const int b0 = sm.rtab[j], nb_r = sm.rtab[j + 1u] - b0;
                const int first = win((unsigned)st);
                const int last1 = win((unsigned)(en - 1)) + 1;
                const bool valid = mine && jr < (unsigned)nr, sign_ok = (st | en) >= 0, pos = en > 0;
                const bool over = last1 > first && last1 > nb_r;
                const int pf = max(b0 + first, off0), pl1 = min(b0 + min(last1, nb_r), t_end);
                bad_any |= __ballot(valid && (!sign_ok || (pos && over)));
                bad_order |= __ballot(mine && !valid);
                if (valid && sign_ok && pos && pf < pl1) add_pm(pf, pl1);
            };
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int left = cur.cnt[u % NSEG] - (u / NSEG) * 64;
                if (left > 0) one(g.rid[u], g.st[u], g.en[u], lane < left);
            }
            if (cur.more) {
                // records behind the slots: streamed until an id at or beyond the tile's last read (or the range's end) says stop, in
                // groups of kStream batches of 64 whose loads go out together (the slots' registers are free by now): one latency per
                // group.  One batch ahead of the one piled up, as until round 6, left a tile of ultralong reads -- ~830 records where
                // the slots hold 512 -- waiting out a load's latency six times over.  (Loads carried across the loop's back edge
                // -- a ring of batches -- are waited for as if they were the youngest: the compiler's wait counts do not survive
                // the merge.)  Raw buffer loads: the range's end is the descriptor's.
                constexpr int kStream = NSEG <= 2 ? 4 : 2;
#pragma unroll
                for (int s = 0; s < NSEG; ++s) {
                    if (!open_run[s]) continue;
                    const int limit = avail_s[s];
                    const unsigned bytes = (unsigned)max(limit, 0) * 4u;
                    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_rid + cur.lo[s]), 0, bytes, 0x00020000);
                    const __amdgpu_buffer_rsrc_t qs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_s + cur.lo[s]), 0, bytes, 0x00020000);
                    const __amdgpu_buffer_rsrc_t qe = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_e + cur.lo[s]), 0, bytes, 0x00020000);
                    int i0 = ITER * 64, total = cur.cnt[s];
                    bool go = i0 < limit;
                    while (go) {
                        int rq[kStream], sq[kStream], eq[kStream];
#pragma unroll
                        for (int d = 0; d < kStream; ++d) {
                            const int vo = (i0 + d * 64 + lane) * 4;
                            rq[d] = __builtin_amdgcn_raw_buffer_load_b32(qr, vo, 0, kRecAux);
                            sq[d] = __builtin_amdgcn_raw_buffer_load_b32(qs, vo, 0, kRecAux);
                            eq[d] = __builtin_amdgcn_raw_buffer_load_b32(qe, vo, 0, kRecAux);
                        }
#pragma unroll
                        for (int d = 0; d < kStream; ++d) {
                            if (!go) continue;
                            const bool mine = i0 + lane < limit && rq[d] < r_b;        // (sorted: a prefix of the lanes)
                            const int n = (int)__popcll(__ballot(mine));
                            one(rq[d], sq[d], eq[d], mine);
                            total += n;
                            i0 += 64;
                            if (n < 64 || i0 >= limit) go = false;
                        }
                    }
                    cur.cnt[s] = total;
                }
                cur.n_total = 0;
#pragma unroll
                for (int s = 0; s < NSEG; ++s) cur.n_total += cur.cnt[s];
            }
            if (bad_order != 0ull && lane == 0) atomicOr(a.err_flags, kErrOrder);
            if (bad_any != 0ull) {         // rare: find the offending records again and report the first index
#pragma unroll
                for (int s = 0; s < NSEG; ++s) {
                    const long long base = (long long)cur.lo[s];
                    for (int i = lane; i < cur.cnt[s]; i += 64) {
                        const int rid = (a.iv_rid + base)[i], st = (a.iv_s + base)[i], en = (a.iv_e + base)[i];
                        if ((unsigned)(rid - r_a) >= (unsigned)nr) continue;
                        const int j = rid - r_a;
                        const int nb_r = sm.rtab[j + 1] - sm.rtab[j];
                        const int first = (int)win_of(a, (unsigned)st), last1 = (int)win_of(a, (unsigned)(en - 1)) + 1;
                        if ((st | en) < 0 || (en > 0 && last1 > first && last1 > nb_r)) raise_error(a, kErrCoord, base + i);
                    }
                }
            }
        } else {
            // window records: (first window, one past the last) in one word; the read from the caller's offsets.  Boundary
            // lane + 1 of a run: where the records of read r_a + lane + 1 begin (relative to the tile's first record of the run)
            bool bad_w = false;
            int bnd[NSEG];
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                const int nx = wave_shl1(rd.so[s], 0) - uni(rd.so[s]);
                bnd[s] = (lane < nr) ? nx : 0x7fffffff;
            }
            auto one_w = [&](int j, unsigned w) {
                const int b0 = sm.rtab[j], nb_r = sm.rtab[j + 1] - b0;
                const int first = (int)(w & 0xffffu), last1 = (int)(w >> 16);
                const bool over = last1 > first && last1 > nb_r;
                const int pf = max(b0 + first, off0), pl1 = min(b0 + min(last1, nb_r), t_end);
                bad_w |= over;
                if (pf < pl1) add_pm(pf, pl1);
            };
            auto read_of = [&](int s, int i0, int i) -> int {   // read of record i of run s; the wave's records are [i0, i0 + 64)
                const int i_last = min(i0 + 63, cur.cnt[s] - 1);
                const unsigned jf = (unsigned)__popcll(__ballot(bnd[s] <= i0)), jl = (unsigned)__popcll(__ballot(bnd[s] <= i_last));
                int j = (int)jf;
                for (unsigned t = jf; t < jl; ++t) j += (i >= __builtin_amdgcn_readlane(bnd[s], (int)t)) ? 1 : 0;
                return min(j, max(nr - 1, 0));
            };
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = u % NSEG, i0 = (u / NSEG) * 64;
                // (the lanes behind the tile's count hold the next tile's records: an empty word piles nothing up)
                if (cur.cnt[s] > i0) one_w(read_of(s, i0, i0 + lane), i0 + lane < cur.cnt[s] ? (unsigned)g.st[u] : 0u);
            }
            if (cur.more) {                      // records beyond the slots: streamed, in groups of kStream batches of 64 whose loads go out together
                constexpr int kStream = 4;
#pragma unroll
                for (int s = 0; s < NSEG; ++s) {
                    const __amdgpu_buffer_rsrc_t qw = __builtin_amdgcn_make_buffer_rsrc((void *)(a.iv_w + cur.lo[s]), 0, (unsigned)max(cur.cnt[s], 0) * 4u, 0x00020000);
                    for (int i0 = ITER * 64; i0 < cur.cnt[s]; i0 += kStream * 64) {
                        unsigned wq[kStream];
#pragma unroll
                        for (int d = 0; d < kStream; ++d) wq[d] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(qw, (i0 + d * 64 + lane) * 4, 0, kRecAux);
#pragma unroll
                        for (int d = 0; d < kStream; ++d)      // (a lane behind the count: 0 from the descriptor's range check -- an empty word)
                            if (i0 + d * 64 < cur.cnt[s]) one_w(read_of(s, i0 + d * 64, i0 + d * 64 + lane), wq[d]);
                    }
                }
            }
            if (__ballot(bad_w) != 0ull) {
#pragma unroll
                for (int s = 0; s < NSEG; ++s) {
                    const long long base = (long long)cur.lo[s];
                    for (int i0 = 0; i0 < cur.cnt[s]; i0 += 64) {
                        const int i = i0 + lane;
                        const int j = read_of(s, i0, i);
                        if (i < cur.cnt[s]) {
                            const unsigned w = (a.iv_w + base)[i];
                            const int first = (int)(w & 0xffffu), last1 = (int)(w >> 16);
                            if (last1 > first && last1 > sm.rtab[j + 1] - sm.rtab[j]) raise_error(a, kErrCoord, base + i);
                        }
                    }
                }
            }
        }
        lane_cov += covsum;
        // ---- a tile with 2^15 intervals or more: every intermediate below is exact modulo 2^16 only.  Its records have been checked
        // above like any other tile's; what they pile up to is left to pileup_deep_kernel (32-bit, a workgroup per tile), which
        // finds the tile's description in a list.  The array is cleared of what the scatter left.
        const bool deep = cur.n_total >= a.deep_min;
        if (deep) {
            int slot = 0;
            if (lane == 0) slot = atomicAdd(a.n_deep, 1);
            slot = uni(slot);
            if (slot < a.deep_cap) {
                if (lane == 0) {
                    DeepTile dt;
                    dt.r_a = cur.r_a; dt.nr = cur.nr; dt.piece = cur.piece; dt.nwin = cur.nwin; dt.g_lo = cur.g_lo;
#pragma unroll
                    for (int s = 0; s < kMaxSeg; ++s) { dt.lo[s] = s < NSEG ? cur.lo[s] : 0; dt.cnt[s] = s < NSEG ? cur.cnt[s] : 0; }
                    reinterpret_cast<DeepTile *>(a.deep_list)[slot] = dt;
                }
            } else if (lane == 0) atomicOr(a.err_flags, kErrDeep);
            for (int i = lane; i < SLOTS / 2; i += 64) sm.diff[i] = kZero;
        }
        if (defer) plan_next();                  // (late: these loads are waited for right below, with nothing to hide behind)

        // ---- every load issued so far -- the NEXT tile's among them -- must have landed before this tile's first coverage
        // store: vmcnt counts loads and stores in one in-order queue, so a wait placed behind the stores would wait for the
        // stores too (measured: a wave that waited for its loads at the end of the tile spent a third of its time there)
        wait_all_loads();

        // ---- 2. rows: prefix sum, store, run detection; every row is zeroed once it is read
        int carry = 0;
        bool hp = false;                 // the slot before the next one is high
        int S = kNone;                   // start slot of the run currently open
        int nq = 0;
        const int rows = deep ? 0 : rows_all;    // (a deep tile: no rows, no runs -- pileup_deep_kernel's)
        int32_t *const cov0 = OW == 4 ? a.cov + a0 : nullptr;
        char *const covp0 = OW == 4 ? nullptr : reinterpret_cast<char *>(a.covp) + (D4 ? a0 / 2 : a0 * OW);
        // (descriptor of this tile's piece of the coverage array: base = the tile's first aligned window, no stride, raw 32-bit data)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(OW == 4 ? (void *)cov0 : (void *)covp0, 0, SLOTS * 4, 0x00020000);
        int pend_p = -1, pend_c = 0;     // D4: this lane's listed window waiting for the end of the rows
        int d4_n = 0;                    // D4: windows of this tile listed so far (wave-uniform)
        auto d4_list = [&](int p, int v, int slot) {
            if (slot < kExcPerTile && tile_id < a.piece_w) {
                const long long at = (long long)tile_id * kExcPerTile + slot;
                a.exc_pidx[at] = a0 + p; a.exc_pval[at] = v;
            } else note_exception(a, a0 + p, v);
        };
        // a run found while the queue is full (a tile of very many short repeats): emitted at once by the lane that found it,
        // from global memory (no shuffles inside divergent code)
        auto emit_overflow = [&](int sS, int sT) {
            if (piece) {
                const int r0 = reinterpret_cast<const int32_t *>(a.rep_res_off + r_a)[0], r1 = reinterpret_cast<const int32_t *>(a.rep_res_off + r_a)[2];
                const int slot = atomicAdd(&a.rep_cnt[r_a], 1);
                if (slot >= r1 - r0) { raise_error(a, kErrInternal, r_a); return; }
                const int start = (sS - sm.rtab[0]) * a.reso;
                const long long ix = (long long)r0 + slot;
                a.raw_key[ix] = start; a.raw_s[ix] = start; a.raw_e[ix] = start + (sT - sS) * a.reso;
                return;
            }
            int jo = 0;                          // owner: the last read that begins at or before the run's first slot
            for (int q = 1; q < nr; ++q) if (sm.rtab[q] <= sS) jo = q;
            const int nwin_r = sT - sS;
            if ((long long)nwin_r * a.reso < (long long)a.repeat_length) return;
            const int len = a.read_len[r_a + jo];
            const int r0 = reinterpret_cast<const int32_t *>(a.rep_res_off + r_a + jo)[0], r1 = reinterpret_cast<const int32_t *>(a.rep_res_off + r_a + jo)[2];
            const int slot = atomicAdd(&sm.rtab[64 + jo], 1);
            const int start = (sS - sm.rtab[jo]) * a.reso, end = start + nwin_r * a.reso;
            int s2 = start - a.flank, e2 = end + a.flank;
            if (s2 <= 0) s2 = 0;
            if (e2 >= len) e2 = len;
            if (slot >= r1 - r0) { raise_error(a, kErrInternal, r_a + jo); return; }
            const long long idx = (long long)r0 + slot;
            a.raw_key[idx] = start; a.raw_s[idx] = s2; a.raw_e[idx] = e2;
            lane_rep += end - start;
        };

        const uint64_t *drow = reinterpret_cast<const uint64_t *>(sm.diff);      // dword pairs: 4 slots
        uint64_t dA = drow[lane], dB = drow[64 + lane];
        for (int row = 0; row < rows; ++row) {
            const int base = row * 512;
            const uint64_t cA = dA, cB = dB;
            // the next row, unconditionally (behind the last row: row 0 again, never used)
            const int nrow = row + 1 < rows ? row + 1 : 0;
            dA = drow[nrow * 128 + lane];
            dB = drow[nrow * 128 + 64 + lane];
            constexpr uint64_t kZero2 = ((uint64_t)kZero << 32) | kZero;
            reinterpret_cast<uint64_t *>(sm.diff)[row * 128 + lane] = kZero2;
            reinterpret_cast<uint64_t *>(sm.diff)[row * 128 + 64 + lane] = kZero2;
            // (the bias stays on: 0x8000 in the low half of every dword.  A lane's two dwords of a half-row carry two of them, which
            // cancel modulo 2^16 from the lane's third slot on -- the lanes' totals, the scan and the carry never see them -- and the
            // first two slots lose theirs with the start value: one xor per row instead of four)
#ifdef RAFT_W_NOBIAS
            const unsigned dA0 = (unsigned)cA ^ kZero, dA1 = (unsigned)(cA >> 32) ^ kZero, dB0 = (unsigned)cB ^ kZero, dB1 = (unsigned)(cB >> 32) ^ kZero;
#else
            const unsigned dA0 = (unsigned)cA, dA1 = (unsigned)(cA >> 32), dB0 = (unsigned)cB, dB1 = (unsigned)(cB >> 32);
#endif
            // in-lane prefix of each half-row's four slots (packed, modulo 2^16 per half)
            const unsigned qA0 = dA0 + (dA0 << 16), qB0 = dB0 + (dB0 << 16);
            const unsigned qA1 = pk_add_bcast<true>(dA1 + (dA1 << 16), qA0), qB1 = pk_add_bcast<true>(dB1 + (dB1 << 16), qB0);
            // the lanes' totals of both half-rows through ONE scan: V = totA + 65536 totB as a 32-bit integer (exact: |tot| < 32768)
            const int V = ((int)qA1 >> 16) + (int)(qB1 & 0xffff0000u);
            const int incl = wave_incl_scan_add(V);
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            const int totA = (int)(short)tot, totB = (tot - totA) >> 16;
            // start values: low half = carry + lanes before in half-row A; high half = carry + all of A + lanes before in B
            const unsigned E = (unsigned)(incl - V) + (unsigned)carry * 0x10001u + ((unsigned)totA << 16);
            carry += totA + totB;
#ifdef RAFT_W_NOBIAS
            const unsigned E0 = E;
#else
            const unsigned E0 = E ^ 0x80008000u;
#endif
            const unsigned rA0 = pk_add_bcast<false>(qA0, E0), rA1 = pk_add_bcast<false>(qA1, E);
            const unsigned rB0 = pk_add_bcast<true>(qB0, E0), rB1 = pk_add_bcast<true>(qB1, E);
            const int pA = base + lane * 4, pB = base + 256 + lane * 4;        // this lane's first slot in each half-row
            const unsigned mx = pk_max_u16(pk_max_u16(rA0, rA1), pk_max_u16(rB0, rB1));
            const bool full = base >= off0 && base + 512 <= t_end;              // every slot of the row is a window of the tile
            // ---- store
            // (a full row's stores: the lane's part of the address is a constant -- lane * 4 windows -- and the half-row's a scalar)
            const int hbase = base;
            auto store_half = [&](unsigned r0, unsigned r1, int p0, unsigned d0, unsigned d1, int h) {
                const unsigned c0 = r0 & 0xffffu, c1 = r0 >> 16, c2 = r1 & 0xffffu, c3 = r1 >> 16;
                // (one register for both halves: the half's 256 windows go with the row's scalar.  NOT for the 16-byte stores of int32
                // coverage: a buffer store of more than 8 bytes whose offset comes from a scalar register may still be reading its data
                // registers when the next vector instruction overwrites them -- the compiler pads that hazard only for the form without
                // the scalar offset, and the pass stored the next half-row's packed words into 155 of 444,741 windows of a test set.
                // There the row goes into the lane's offset, one v_lshl_or per row.)
                const int lane_w = OW == 4 ? lane * 4 + hbase + h * 256 : lane * 4;
                const int base = OW == 4 ? 0 : hbase + h * 256;
                if (OW == 4) {
                    if (full) __builtin_amdgcn_raw_buffer_store_b128(v4i{(int)c0, (int)c1, (int)c2, (int)c3}, rsrc, lane_w * 4, base * 4, CovAux<OW>::v);
                    else {      // (a tile's first and last row: the same addressing, element by element where the lane straddles the edge)
                        const unsigned q0 = (unsigned)(p0 - off0), nw = (unsigned)cur.nwin;
                        const bool v0 = q0 < nw, v1 = q0 + 1u < nw, v2 = q0 + 2u < nw, v3 = q0 + 3u < nw;
                        if (v0 && v3) __builtin_amdgcn_raw_buffer_store_b128(v4i{(int)c0, (int)c1, (int)c2, (int)c3}, rsrc, lane_w * 4, base * 4, CovAux<OW>::v);
                        else {
                            if (v0) __builtin_amdgcn_raw_buffer_store_b32((int)c0, rsrc, lane_w * 4, base * 4, CovAux<OW>::v);
                            if (v1) __builtin_amdgcn_raw_buffer_store_b32((int)c1, rsrc, lane_w * 4 + 4, base * 4, CovAux<OW>::v);
                            if (v2) __builtin_amdgcn_raw_buffer_store_b32((int)c2, rsrc, lane_w * 4 + 8, base * 4, CovAux<OW>::v);
                            if (v3) __builtin_amdgcn_raw_buffer_store_b32((int)c3, rsrc, lane_w * 4 + 12, base * 4, CovAux<OW>::v);
                        }
                    }
                } else if (OW == 1 || OW == 2) {
                    // (the windows at or above the limit are listed once per row, below: the common row has none)
                    const unsigned m0 = pk_min_u16(r0, kLimit * 0x10001u), m1 = pk_min_u16(r1, kLimit * 0x10001u);
                    if (full) {
                        if (OW == 1) __builtin_amdgcn_raw_buffer_store_b32((int)__builtin_amdgcn_perm(m1, m0, 0x06040200u), rsrc, lane_w, base, CovAux<OW>::v);
                        else __builtin_amdgcn_raw_buffer_store_b64(v2i{(int)m0, (int)m1}, rsrc, lane_w * 2, base * 2, CovAux<OW>::v);
                    } else {
                        const unsigned q0 = (unsigned)(p0 - off0), nw = (unsigned)cur.nwin;
                        const bool v0 = q0 < nw, v1 = q0 + 1u < nw, v2 = q0 + 2u < nw, v3 = q0 + 3u < nw;
                        // (window records: the edge rows' element stores by the same addressing as the full rows'; with coordinate columns
                        // in, the registers that takes are the ones the kernel does not have: 12 bytes of scratch)
                        if (IN == 0 && OW == 1) {
                            uint8_t *o = reinterpret_cast<uint8_t *>(covp0) + p0;
                            const unsigned pk4 = __builtin_amdgcn_perm(m1, m0, 0x06040200u);
                            if (v0 && v3) __builtin_amdgcn_raw_buffer_store_b32((int)pk4, rsrc, p0, 0, CovAux<OW>::v);
                            else { if (v0) o[0] = (uint8_t)pk4; if (v1) o[1] = (uint8_t)(pk4 >> 8); if (v2) o[2] = (uint8_t)(pk4 >> 16); if (v3) o[3] = (uint8_t)(pk4 >> 24); }
                        } else if (IN == 0) {
                            uint16_t *o = reinterpret_cast<uint16_t *>(covp0) + p0;
                            if (v0 && v3) __builtin_amdgcn_raw_buffer_store_b64(v2i{(int)m0, (int)m1}, rsrc, p0 * 2, 0, CovAux<OW>::v);
                            else { if (v0) o[0] = (uint16_t)m0; if (v1) o[1] = (uint16_t)(m0 >> 16); if (v2) o[2] = (uint16_t)m1; if (v3) o[3] = (uint16_t)(m1 >> 16); }
                        } else if (OW == 1) {
                            const unsigned pk4 = __builtin_amdgcn_perm(m1, m0, 0x06040200u);
                            if (v0 && v3) __builtin_amdgcn_raw_buffer_store_b32((int)pk4, rsrc, lane_w, base, CovAux<OW>::v);
                            else {
                                if (v0) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)pk4, rsrc, lane_w, base, CovAux<OW>::v);
                                if (v1) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(pk4 >> 8), rsrc, lane_w + 1, base, CovAux<OW>::v);
                                if (v2) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(pk4 >> 16), rsrc, lane_w + 2, base, CovAux<OW>::v);
                                if (v3) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(pk4 >> 24), rsrc, lane_w + 3, base, CovAux<OW>::v);
                            }
                        } else {
                            if (v0 && v3) __builtin_amdgcn_raw_buffer_store_b64(v2i{(int)m0, (int)m1}, rsrc, lane_w * 2, base * 2, CovAux<OW>::v);
                            else {
                                if (v0) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)m0, rsrc, lane_w * 2, base * 2, CovAux<OW>::v);
                                if (v1) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(m0 >> 16), rsrc, lane_w * 2 + 2, base * 2, CovAux<OW>::v);
                                if (v2) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)m1, rsrc, lane_w * 2 + 4, base * 2, CovAux<OW>::v);
                                if (v3) __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(m1 >> 16), rsrc, lane_w * 2 + 6, base * 2, CovAux<OW>::v);
                            }
                        }
                    }
                } else {
                    // four-bit steps: a step IS the difference array's value; the lane's four steps are one aligned ushort
                    if (full && hbase > off0) {
                        // the common row (every slot a window of the tile, none of them its first): the four codes by packed
                        // arithmetic -- step + 7 clamped to 15 per half (a step outside [-7, 7] wraps or exceeds: 15), + 1 modulo 16 makes
                        // 1 .. 15 of a step that fits and 0, "listed", of one that does not; two shifts put the nibbles side by side
                        // (d0 / d1 still carry the array's bias in their low halves: taken off with the + 7)
                        const unsigned m0 = pk_min_u16(pk_add_u16(d0, 0x00078007u), 0x000f000fu), m1 = pk_min_u16(pk_add_u16(d1, 0x00078007u), 0x000f000fu);
                        const unsigned k0 = pk_add_u16(m0, 0x00010001u) & 0x000f000fu, k1 = pk_add_u16(m1, 0x00010001u) & 0x000f000fu;
                        const unsigned code = ((k0 | (k0 >> 12)) & 0xffu) | (((k1 | (k1 >> 12)) & 0xffu) << 8);
                        *reinterpret_cast<uint16_t *>(covp0 + ((unsigned)p0 >> 1)) = (uint16_t)code;
                        if ((((unsigned)a0 + (unsigned)p0 + (unsigned)a.d4_shift) & 1023u) == 0u)
                            a.cov_anchor[(a0 + p0 + a.d4_shift) >> 10] = (int)c0 - (int)(short)(d0 ^ kZero);
                        unsigned t = code | (code >> 1);
                        t |= t >> 2;
                        unsigned esc = ~t & 0x1111u;                                  // bit 4q: window q of this lane is listed
                        if (esc && pend_p < 0) {
                            const int q = (__ffs((int)esc) - 1) >> 2;
                            esc &= esc - 1u;
                            pend_p = p0 + q; pend_c = q == 0 ? (int)c0 : q == 1 ? (int)c1 : q == 2 ? (int)c2 : (int)c3;
                        }
                        unsigned long long em = __ballot(esc != 0u);
                        while (em) {
                            if (esc) {
                                const int q = (__ffs((int)esc) - 1) >> 2;
                                esc &= esc - 1u;
                                d4_list(p0 + q, q == 0 ? (int)c0 : q == 1 ? (int)c1 : q == 2 ? (int)c2 : (int)c3, d4_n + (int)__popcll(em & ((1ull << lane) - 1ull)));
                            }
                            d4_n += (int)__popcll(em);
                            em = __ballot(esc != 0u);
                        }
                        return;
                    }
                    const int s0 = (int)(short)(d0 ^ kZero), s1 = (int)d0 >> 16, s2 = (int)(short)(d1 ^ kZero), s3 = (int)d1 >> 16;
                    const unsigned q0 = (unsigned)(p0 - off0), nw = (unsigned)cur.nwin;
                    const bool v0 = q0 < nw, v1 = q0 + 1u < nw, v2 = q0 + 2u < nw, v3 = q0 + 3u < nw;
                    const unsigned valid = full ? 15u : ((v0 ? 1u : 0u) | (v1 ? 2u : 0u) | (v2 ? 4u : 0u) | (v3 ? 8u : 0u));
                    unsigned esc = 0u;
                    if (valid) {
                        const unsigned ux = (unsigned)(s0 + 7), uy = (unsigned)(s1 + 7), uz = (unsigned)(s2 + 7), uw = (unsigned)(s3 + 7);
                        unsigned code = (ux <= 14u ? ux + 1u : 0u) | ((uy <= 14u ? uy + 1u : 0u) << 4) | ((uz <= 14u ? uz + 1u : 0u) << 8) | ((uw <= 14u ? uw + 1u : 0u) << 12);
                        const unsigned f = (unsigned)(off0 - p0);                     // < 4: the tile's first window is this lane's slot f
                        if (f < 4u) code &= ~(0xFu << (4u * f));
                        const unsigned vmask = ((valid & 1u) ? 0xFu : 0u) | ((valid & 2u) ? 0xF0u : 0u) | ((valid & 4u) ? 0xF00u : 0u) | ((valid & 8u) ? 0xF000u : 0u);
                        code &= vmask;
                        unsigned t = code | (code >> 1);
                        t |= t >> 2;
                        esc = ~t & 0x1111u & vmask;                                   // bit 4q: window q of this lane is listed
                        char *const o = covp0 + ((unsigned)p0 >> 1);
                        if (valid == 15u) *reinterpret_cast<uint16_t *>(o) = (uint16_t)code;
                        else {      // the neighbouring tile owns the other nibbles of this ushort: clear mine, then set them
                            const unsigned long long addr = reinterpret_cast<unsigned long long>(o);
                            unsigned *const word = reinterpret_cast<unsigned *>(addr & ~3ull);
                            const unsigned sh = (unsigned)(addr & 2ull) * 8u;
                            atomicAnd(word, ~(vmask << sh));
                            atomicOr(word, (code & vmask) << sh);
                        }
                        // the lane whose first window opens a block of 1024 stores the prefix before it as the anchor
                        if ((valid & 1u) && (((unsigned)a0 + (unsigned)p0 + (unsigned)a.d4_shift) & 1023u) == 0u)
                            a.cov_anchor[(a0 + p0 + a.d4_shift) >> 10] = (int)c0 - s0;
                        // a lane parks its first listed window until the rows are done
                        if (esc && pend_p < 0) {
                            const int q = (__ffs((int)esc) - 1) >> 2;
                            esc &= esc - 1u;
                            pend_p = p0 + q; pend_c = q == 0 ? (int)c0 : q == 1 ? (int)c1 : q == 2 ? (int)c2 : (int)c3;
                        }
                    }
                    // (rare) further listed windows of a lane: placed at once, in the tile's own slots, ranked by ballot
                    unsigned long long em = __ballot(esc != 0u);
                    while (em) {
                        if (esc) {
                            const int q = (__ffs((int)esc) - 1) >> 2;
                            esc &= esc - 1u;
                            d4_list(p0 + q, q == 0 ? (int)c0 : q == 1 ? (int)c1 : q == 2 ? (int)c2 : (int)c3, d4_n + (int)__popcll(em & ((1ull << lane) - 1ull)));
                        }
                        d4_n += (int)__popcll(em);
                        em = __ballot(esc != 0u);
                    }
                }
            };
            if (!(kMode & 8)) {                  // (diagnostic: no coverage stores)
                store_half(rA0, rA1, pA, dA0, dA1, 0);
                store_half(rB0, rB1, pB, dB0, dB1, 1);
            }
            // windows at or above a byte's limit (a 16-bit tile's values are below 32768: the two-byte encoding's 65535 is never
            // reached here): ONE test per row on the packed maximum, c + (0x8000 - limit) has bit 15 set <=> c >= limit
            if (OW == 1) {
                constexpr unsigned klim = ((0x8000u - kLimit) & 0xffffu) * 0x10001u;
                if (__ballot(((mx + klim) & 0x80008000u) != 0u) != 0ull) {
                    auto list_half = [&](unsigned r0, unsigned r1, int p0) {
                        const unsigned c0 = r0 & 0xffffu, c1 = r0 >> 16, c2 = r1 & 0xffffu, c3 = r1 >> 16;
                        const unsigned q0 = (unsigned)(p0 - off0), nw = (unsigned)cur.nwin;
                        if (q0 < nw && c0 >= kLimit) note_exception(a, a0 + p0, (int)c0);
                        if (q0 + 1u < nw && c1 >= kLimit) note_exception(a, a0 + p0 + 1, (int)c1);
                        if (q0 + 2u < nw && c2 >= kLimit) note_exception(a, a0 + p0 + 2, (int)c2);
                        if (q0 + 3u < nw && c3 >= kLimit) note_exception(a, a0 + p0 + 3, (int)c3);
                    };
                    list_half(rA0, rA1, pA);
                    list_half(rB0, rB1, pB);
                }
            }

            // ---- run scan: only rows that hold a high window or inherit an open run
            const unsigned xh = (mx + kthr) & 0x80008000u;
            if (kMode & 2) continue;             // (diagnostic, RAFT_WAVE_MODE: no run scan)
            if (__ballot(xh != 0u) == 0ull && !hp) continue;
            // per half-row: a per-lane scan of four slots
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int hb = base + h * 256;                         // first slot of the half-row
                if (hb > t_end) break;
                const unsigned x0 = (h ? rB0 : rA0) + kthr, x1 = (h ? rB1 : rA1) + kthr;
                const int p0 = hb + lane * 4;
                int hvn = (int)(((x0 >> 15) & 1u) | ((x0 >> 30) & 2u) | ((x1 >> 13) & 4u) | ((x1 >> 28) & 8u));
                // slots outside the tile are not windows
                int inside = 15;
                if (hb < off0 || hb + 256 > t_end) {
                    const int lo_k = min(max(off0 - p0, 0), 4), hi_k = min(max(t_end - p0, 0), 4);
                    inside = ((1 << hi_k) - 1) & ~((1 << lo_k) - 1);
                }
                hvn &= inside;
                const unsigned long long HV = __ballot(hvn != 0);
                if (HV == 0ull && !hp) continue;
                const unsigned long long RS = __ballot((ro_s >> 8) == (hb >> 8) && ro_s >= hb);   // reads that begin in this half-row
                const bool tail = hb + 256 > t_end;
                if (hp && !tail && RS == 0ull && __ballot(hvn == 15) == ~0ull) continue;            // a run passes through the half-row
                int sbm = 0;
                {
                    unsigned long long rs = RS;
                    while (rs) {
                        const int pos = __builtin_amdgcn_readlane(ro_s, (int)__builtin_ctzll(rs)) - hb;
                        rs &= rs - 1ull;
                        if ((pos >> 2) == lane) sbm |= 1 << (pos & 3);
                    }
                }
                const int hk = (hb < off0 + 1 && hb + 256 > off0) ? (off0 - hb) : 0;   // the carried-in bit belongs to the first valid slot
                int prvn = ((hvn << 1) & 15) | wave_shr1(hvn >> 3, 0);
                if (hp && lane == (hk >> 2)) prvn |= 1 << (hk & 3);
                int inside_e = 15;                                   // slots at which a run may END: t_end itself included (the sentinel)
                if (tail) inside_e = (1 << min(max(t_end - p0 + 1, 0), 4)) - 1;
                const int starts = hvn & ((prvn ^ 15) | sbm);
                int ends = prvn & ((hvn ^ 15) | sbm) & inside_e;
                const int ls = starts ? p0 + (31 - __clz(starts)) : -0x40000000;
                const int incl2 = wave_incl_scan_max(ls, -0x40000000);
                const int carried = max(S, wave_shr1(incl2, -0x40000000));
                while (__ballot(ends != 0) != 0ull) {
                    const bool has = ends != 0;
                    const int kq = has ? __builtin_ctz(ends) : 0;
                    const int below = starts & ((1 << kq) - 1);
                    const int best = max(carried, below ? p0 + (31 - __clz(below)) : -0x40000000);
                    const int t = p0 + kq;
                    ends &= ends - 1;
                    const bool keep = has && best >= 0 &&
                                      ((long long)(t - best) * a.reso >= (long long)a.repeat_length || (piece && (best == off0 || t == t_end)));
                    const unsigned long long km = __ballot(keep);
                    if (km) {
                        const int idx = nq + (int)__popcll(km & ((1ull << lane) - 1ull));
                        if (keep) {
                            if (idx < kRunQ) { sm.runq[idx * 2] = best; sm.runq[idx * 2 + 1] = t; }
                            else emit_overflow(best, t);
                        }
                        nq = min(kRunQ, nq + (int)__popcll(km));
                    }
                }
                S = max(S, __builtin_amdgcn_readlane(incl2, 63));
                // what the next half-row inherits: is its predecessor slot high?
                if (!tail) hp = (__builtin_amdgcn_readlane(hvn, 63) & 8) != 0;
                else hp = false;        // (the run that reaches t_end was closed at the sentinel slot)
            }
        }

        if (D4) {                            // the windows the lanes parked: one place in the tile's list each
            const unsigned long long pm = __ballot(pend_p >= 0);
            if (pm) {
                if (pend_p >= 0) d4_list(pend_p, pend_c, d4_n + (int)__popcll(pm & ((1ull << lane) - 1ull)));
                d4_n += (int)__popcll(pm);
            }
            if (d4_n && lane == 0 && tile_id < a.piece_w) a.exc_tile_n[tile_id] = min(d4_n, kExcPerTile);
        }
        // ---- 3. every parked run becomes a raw repeat record, one lane per run
        if (nq > 0) {
            int sS = 0, sT = 0, j = 0;
            if (lane < nq) { sS = sm.runq[lane * 2]; sT = sm.runq[lane * 2 + 1]; }
#pragma unroll 1
            for (int q = 0; q < nq; ++q) {
                const int s0 = __builtin_amdgcn_readlane(sS, q);
                const int jq = __popcll(__ballot(ro_s <= s0)) - 1;
                if (lane == q) j = jq;
            }
            if (piece) {
                const int r0 = __builtin_amdgcn_readlane(rd.rr, 0), r1 = __builtin_amdgcn_readlane(rd.rr, 1);
                if (lane < nq) {
                    const int slot = atomicAdd(&a.rep_cnt[r_a], 1);
                    if (slot >= r1 - r0) raise_error(a, kErrInternal, r_a);
                    else {
                        const int start = (sS - sm.rtab[0]) * a.reso;
                        const long long ix = (long long)r0 + slot;
                        a.raw_key[ix] = start; a.raw_s[ix] = start; a.raw_e[ix] = start + (sT - sS) * a.reso;
                    }
                }
            } else {
                // (shuffles below need every lane: lanes without a run carry an empty one)
                const int jj = lane < nq ? max(j, 0) : 0;
                const int nwin_r = sT - sS;
                const bool live = lane < nq && (long long)nwin_r * a.reso >= (long long)a.repeat_length;
                const int len = __shfl(rd.rl, jj), r0 = __shfl(rd.rr, jj), r1 = __shfl(rd.rr, jj + 1);
                if (live) {
                    const int off = sm.rtab[jj];
                    const int slot = atomicAdd(&sm.rtab[64 + jj], 1);
                    const int start = (sS - off) * a.reso, end = start + nwin_r * a.reso;
                    int s = start - a.flank, e = end + a.flank;
                    if (s <= 0) s = 0;
                    if (e >= len) e = len;
                    if (slot >= r1 - r0) raise_error(a, kErrInternal, r_a + jj);
                    else {
                        const long long idx = (long long)r0 + slot;
                        a.raw_key[idx] = start; a.raw_s[idx] = s; a.raw_e[idx] = e;
                        lane_rep += end - start;
                    }
                }
            }
            if (!piece && lane < nr) {
                const int c = sm.rtab[64 + lane];
                if (c) a.rep_cnt[r_a + lane] = c;
            }
        }

        // ---- hand over to the next tile
        if (!have_next) break;
        if (reuse) {
            const int src = lane + shift;
            rd.cv = __shfl(rd.cv, src); rd.rr = __shfl(rd.rr, src); rd.rl = __shfl(rd.rl, src);
            if (IN == 1) {
#pragma unroll
                for (int s = 0; s < NSEG; ++s) rd.so[s] = __shfl(rd.so[s], src);
            }
            rd.n -= shift;
        } else rd = rdn;
        ts = nts; g = gn;
    }
    {
        const long long cs = wave_reduce_add64(lane_cov), rs = wave_reduce_add64(lane_rep);
        if (lane == 0) { a.block_sums[2 * (long long)wave_id] = cs; a.block_sums[2 * (long long)wave_id + 1] = rs; }
    }
}

// WPB waves per workgroup, each with a tile stream and an LDS slice of its own (no barrier anywhere); WPS waves per SIMD asked
// of the register allocator.
template <int SLOTS, int NSEG, int U, int OW, int IN, int WPB, int WPS>
__global__ __launch_bounds__(64 * WPB, WPS) void pileup_wave_kernel(const TileCut *__restrict__ cuts, PileupArgs a)
{
    __shared__ __attribute__((aligned(16))) WaveSmem<SLOTS> sm[WPB];
    const int wid = WPB == 1 ? 0 : uni((int)(threadIdx.x >> 6));
    wave_tile_loop<SLOTS, NSEG, U, OW, IN>(sm[wid], cuts, a, (int)blockIdx.x * WPB + wid, (int)gridDim.x * WPB);
}

} // namespace raft
