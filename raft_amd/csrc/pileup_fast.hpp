// pileup_fast.hpp -- the lean pileup kernel for tiles of whole reads that fit one LDS window (all tiles of a
// HiFi-like read set).  Same algorithm as pileup.hpp (which keeps handling the other tiles: reads longer than the
// LDS window, tiles with very many reads); what differs is the bookkeeping around the rows, which in the general
// kernel costs four times the vector instructions of the rows themselves (rocprofv3 SQ_INSTS_VALU) and five
// workgroup barriers per tile.  The kernel's time follows its instruction count, not its bytes.
//   * tiles are handed out in batches from a device counter: with a fixed stride the persistent workgroups finished
//     up to a third apart (tiles inside repeats cost more; CUs are not equally fast);
//   * a tile is described by two adjacent 32-byte TileCut records (first read, first interval per segment, first
//     window); the pair for the tile after next travels as one VGPR, lane l holding dword l (scalar loads held in
//     ten SGPRs across a tile were spilled on arrival and cost three exposed waits per tile);
//   * prefetch slot u of a lane is interval (u / NSEG) * 256 + lane-id of segment u % NSEG: a per-tile scalar base
//     plus the lane's constant offset, no per-record index arithmetic; intervals beyond the prefetched slots (tiles
//     inside repeats) are fetched synchronously afterwards;
//   * three barriers per tile: pass B zeroes each row of the difference array right after reading it, and the
//     per-read tables are double-buffered -- those of tile i+1 are written from the prefetched registers just
//     before pass B of tile i, the repeat counts of tile i-1 are published at the same point -- so no clearing
//     phase, no table phase and no publish phase stand between barriers;
//   * four ballots per row tell whether it holds a high window at all; the run scan of the rows that do is done per
//     lane on the lane's four slots as four bits (one wave shift for "is the slot before mine high", one max-scan for
//     "where did the run that reaches me begin"); read boundaries come from registers, not from an LDS bit array;
//   * coverage totals accumulate per lane in registers and are reduced once per workgroup, not once per tile.
// Reference semantics: repeat.hpp:28-79 (profileCoverage), repeat.hpp:111-168 (run scan); see pileup.hpp.
#pragma once
#include "pileup.hpp"

namespace raft {

template <int U>
struct FastRegs {
    int rid[U], st[U], en[U];
};
struct FastReadRegs {           // per read of a tile (thread j <-> read r_a + j, j <= nr)
    int cv, rr, rl;             // low dwords of cov_off / rep_res_off, read length
    int so[kMaxSeg];            // IN = 1: low dword of the read's first record in each run (GroupedOff::off)
};

struct FastTile {               // scalars of one tile
    int r_a, nr, nwin, fast, more;   // more: a segment holds intervals beyond the prefetched slots
    int piece;                       // the tile is a piece of one read longer than the LDS window (kCutPiece)
    int cnt[kMaxSeg];                // intervals per segment
    long long g_lo;
};

constexpr int kFastMaxReads = 86;    // reads per fast tile (their tables live in LDS, double-buffered)

template <int CAP, int NSO = 0>
struct FastSmem {
    static constexpr int NW = 4;
    static constexpr int SLOTS = CAP + 256;  // window slots: 3 alignment + CAP + 1 sentinel, rounded to rows
    static constexpr int TAB = kFastMaxReads + 2;
    static_assert(SLOTS / 256 <= 32, "pass B keeps one bit per row in a 32-bit mask");
    int32_t diff[SLOTS];                     // zero whenever no tile is between its interval phase and its pass B
    int32_t roff[2][TAB];                    // first slot of read r_a+j relative to a0 (j <= nr)
    int32_t rlen[2][TAB];
    int32_t rcnt[2][TAB];                    // raw repeats emitted for the read
    int32_t rres[2][TAB];                    // its first reserved raw-repeat slot (rep_res_off, < 2^31 checked by the host)
    int32_t soff[2][NSO > 0 ? NSO : 1][NSO > 0 ? TAB : 2];   // IN = 1: first record of read r_a+j in run s (low dword of the caller's offset)
    unsigned long long acc_cov, acc_rep;
    __attribute__((aligned(16))) int32_t wsum[NW];
    int32_t next_tile;                       // tile index wave 0 drew for the workgroup (dynamic tile hand-out)
    int32_t exc_n;                           // OW = 8: windows of the current tile listed so far (PileupArgs::exc_pidx)
    int32_t wst[NW * 8];                     // per wave: rows, pclose, sfinal, hpfinal, hpin
    unsigned long long stamps[16];           // diagnostic build
    int32_t runq[NW * 2 * kRunQ];            // per wave: closed runs parked for emission (slots relative to a0)
};

struct FastTables {             // the table set of one tile, in the shape emit_run_of()/owner_slot() expect
    int32_t *roff, *rlen, *rcnt, *rres;
    unsigned long long &acc_rep;
};

// Two adjacent cuts (16 dwords) travel as ONE VGPR -- lane l holds dword l -- from the load one tile ahead to the
// readlanes that unpack them: ten scalar registers held across a whole tile would be spilled to VGPR lanes right
// after an s_load (measured: three exposed scalar-load waits per tile).
template <int NSEG, int ITER>
__device__ __forceinline__ void cut_unpack(int raw, FastTile &t, int (&lo)[NSEG], int (&n)[NSEG])
{
    auto d = [&](int i) -> int { return __builtin_amdgcn_readlane(raw, i); };
    auto q = [&](int i) -> long long { return (long long)(((unsigned long long)(unsigned)d(i + 1) << 32) | (unsigned)d(i)); };
    const int r0 = d(0);
    t.r_a = r0; t.nr = d(8) - r0; t.fast = d(1) & kCutFast; t.piece = d(1) & kCutPiece; t.g_lo = q(6);
    t.nwin = d(14) - d(6);                        // low dwords suffice: a fast tile has at most CAP windows
    t.more = 0;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        lo[s] = d(2 + s); n[s] = d(10 + s) - lo[s];
        t.cnt[s] = n[s];
        if (n[s] > ITER * 256) t.more = 1;
    }
}

// Loads of one tile: three per read (threads 0..nr) and three per interval slot.  Every address is a per-tile scalar
// base plus this lane's constant byte offset (global_load saddr form).  The offsets pass through an empty asm so
// that the compiler cannot fold them into six loop-invariant 64-bit per-lane pointers (12 VGPRs held for the whole
// kernel); `at` does the typed load.
template <class T>
__device__ __forceinline__ T at(const T *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

template <int NSEG, int U, int IN>
__device__ __forceinline__ void fast_issue(const PileupArgs &a, unsigned tid, const FastTile &t, const int (&lo)[NSEG],
                                           const int (&n)[NSEG], FastRegs<U> &g, FastReadRegs &rd)
{
    unsigned b4 = tid * 4u, b8 = tid * 8u;
    asm volatile("" : "+v"(b4), "+v"(b8));
    rd.cv = 0; rd.rr = 0; rd.rl = 0;
    // (IN = 1: the zeroes are pinned ahead of the loads.  Left to itself the compiler sank "rl = 0" for the lanes without a read
    // BEHIND the other lanes' load of the same register and put a wait for every outstanding load -- and store -- before it:
    // 2.7 k of a tile's 16 k cycles in the issue phase, tools/stamp_probe.py)
    if (IN == 1) asm volatile("" : "+v"(rd.cv), "+v"(rd.rr), "+v"(rd.rl));
    if ((int)tid <= t.nr) {
        rd.cv = at(reinterpret_cast<const int32_t *>(a.cov_off + t.r_a), b8);
        rd.rr = at(reinterpret_cast<const int32_t *>(a.rep_res_off + t.r_a), b8);
        if (IN == 1) {
            // where the read's records begin in each run: the low dword as it is -- relative to the tile's first read when it is
            // used (32-bit wrap-around keeps that difference exact).  (No arithmetic on the loaded value here: the first version
            // rebased it on the spot, which put a wait for the load into the issue phase -- 2.7 k of a tile's 16 k cycles.)
#pragma unroll
            for (int s = 0; s < NSEG; ++s) rd.so[s] = at(reinterpret_cast<const int32_t *>(a.grp.off + s * a.grp.stride + t.r_a), b8);
        }
    }
    if ((int)tid < t.nr) rd.rl = at(a.read_len + t.r_a, b4);
    // (Measured and dropped, round 3: segment s taken by the threads rotated by s waves, so that not always wave 0 gets the
    // partly filled last slot of EVERY segment -- no difference, hg002 2.58 / 2.59 ms, ultralong 2.53 / 2.53.)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int s = u % NSEG, first = (u / NSEG) * 256;
        if (IN == 1) {
            g.st[u] = 0;                                  // an empty slot is an empty interval (windows [0, 0))
            if ((int)tid < n[s] - first) g.st[u] = (int)at(a.iv_w + ((long long)lo[s] + first), b4);
        } else {
            g.rid[u] = t.r_a; g.st[u] = 0; g.en[u] = 0;   // an empty slot is an empty interval of the tile's first read: no effect
            if ((int)tid < n[s] - first) {
                const long long base = (long long)lo[s] + first;
                g.rid[u] = at(a.iv_rid + base, b4);
                g.st[u] = at(a.iv_s + base, b4);
                g.en[u] = at(a.iv_e + base, b4);
            }
        }
    }
}

// A run of high windows inside a piece of a long read (slots [sS, sT) of the piece's LDS window): recorded as it is --
// unflanked [start, end) in bases of the read -- because it may continue in the neighbouring piece; finalize_count_kernel
// joins the pieces' runs, applies the length test, the flanks and the clamp (repeat.hpp:125-140), and adds the read's
// repeat bases to the total.  Pieces of one read are processed by different workgroups: slots come from a global counter.
__device__ __forceinline__ void emit_piece_run(const PileupArgs &a, const FastTables &tb, int r, int sS, int sT)
{
    const int r0 = tb.rres[0], r1 = tb.rres[1];
    const int slot = atomicAdd(&a.rep_cnt[r], 1);
    if (slot >= r1 - r0) { raise_error(a, kErrInternal, r); return; }
    const int start = (sS - tb.roff[0]) * a.reso;    // (roff[0]: the read's first window, before the piece's slots)
    const long long idx = (long long)r0 + slot;
    a.raw_key[idx] = start;
    a.raw_s[idx] = start;
    a.raw_e[idx] = start + (sT - sS) * a.reso;
}

// EXTRA = false: the regular tiles (bounded by adjacent cuts).  EXTRA = true: the same kernel over the extra tiles that
// tile_desc_kernel cut out of what does not fit (explicit cut pairs; intervals clipped to the tile; pieces of long reads):
// an instantiation of its own, so that the regular tiles' code carries none of that.
// OW = bytes per window of the coverage written: 4 = cov[] as int32; 1 / 2 = its transfer encoding (PileupArgs::covp) --
// the consumer of cov[] is a text formatter on the host, four fifths of the kernel's HBM traffic is this array, and real
// coverage fits a byte.
// (LS: the lane-serial rows of round 3 -- equal on the byte path, 7-9 % slower with int32 output -- were removed in round 4; the
// parameter stays false.)
//
// IN = 1 ("window records", round 3): a record is ONE 32-bit word -- its first window and one past its last, 16 bits each,
// cut from the coordinates by the tokeniser (raft_host_pack_windows) -- and carries no read id: the caller's offsets say
// where each read's records begin (GroupedOff), the tile's slice of them sits in LDS beside the other per-read tables, and
// a wave finds the reads of its 64 consecutive records with two ballots and a short loop over the boundaries between.
// 4 bytes per record cross the link and are read by this kernel instead of 12.
template <int CAP, int NSEG, int U, bool DIAG, bool EXTRA, int OW, bool LS, int IN>
__device__ __forceinline__ void fast_tile_loop(FastSmem<CAP, IN ? NSEG : 0> &sm, const TileCut *__restrict__ cuts, const PileupArgs &a)
{
    constexpr int THREADS = 256, NW = 4, ITER = U / NSEG;
    static_assert(U % NSEG == 0 && ITER >= 1, "slots are split evenly over the segments");
    static_assert(!LS, "the lane-serial rows were removed in round 4");
    using Smem = FastSmem<CAP, IN ? NSEG : 0>;
    const unsigned tid = threadIdx.x;
    const int lane = (int)(tid & 63u);
    const int wid = uni((int)(tid >> 6));
    // tile indices fit 32 bits (the host checks n_tiles * 18 < 2^31): scalar compares, half the registers
    const int nb = (int)gridDim.x;
    // regular tiles 0 .. n_reg - 1 are bounded by adjacent cuts; extra tiles (tile_desc_kernel re-cut what does not fit:
    // PileupArgs::n_extra) follow as explicit (begin, end) pairs behind the closing boundary, numbered from 0 here
    const int n_reg = (int)a.n_tiles;
    // (an overflowed list -- kErrExtra -- counts tiles that were never written: nothing to do, the engine runs the pass again;
    // kErrStop: the pass was sized by a window count, or cut by offsets, that the device found wrong, see pileup.hpp)
    if (uni(*(volatile int32_t *)a.err_flags) & (EXTRA ? (kErrExtra | kErrStop) : kErrStop)) return;
    const int n_tiles = EXTRA ? uni(*a.n_extra) : n_reg;
    const int last_cut = n_tiles - 1;            // (tile index: the word index below maps it)
    auto cut_of = [&](int t) -> int { return EXTRA ? n_reg + 1 + 2 * t : t; };   // index of tile t's first cut
    // window of base n without a branch: n / reso == ((n & win_m1) | mulhi(n, div_magic)) >> win_sh  (div_magic is 0 and
    // win_m1 all ones when reso == 1, see win_of)
    const unsigned win_m1 = a.div_shift < 0 ? ~0u : 0u;
    const int win_sh = a.div_shift < 0 ? 0 : a.div_shift;

    // LDS starts clean: the difference array is zero between tiles
    for (int i = (int)tid * 4; i < Smem::SLOTS; i += THREADS * 4) *reinterpret_cast<int4 *>(&sm.diff[i]) = make_int4(0, 0, 0, 0);
    if (tid == 0) { sm.acc_cov = 0ull; sm.acc_rep = 0ull; sm.exc_n = 0; }
    lds_barrier();
    long long lane_cov = 0;                      // this lane's share of the coverage total (reduced once, at the end)

    // per-read tables of a tile, from the registers its loads filled
    auto stage_reads = [&](int set, const FastTile &t, const FastReadRegs &rd) {
        if ((int)tid <= t.nr) {
            sm.roff[set][tid] = rd.cv - (int)(t.g_lo & ~3LL);       // 32-bit wrap-around is exact
            sm.rlen[set][tid] = rd.rl; sm.rres[set][tid] = rd.rr; sm.rcnt[set][tid] = 0;
            if (IN == 1) {
#pragma unroll
                for (int s = 0; s < NSEG; ++s) sm.soff[set][s][tid] = rd.so[s];
            }
        }
    };
    // repeat counts of a finished tile (rep_cnt[] was zeroed by the host)
    auto publish_counts = [&](int set, int r_a, int nr) {
        if ((int)tid < nr) {
            const int c = sm.rcnt[set][tid];
            if (c) a.rep_cnt[r_a + tid] = c;
        }
    };

    const int32_t *cut_words = reinterpret_cast<const int32_t *>(cuts);
    auto cut_word = [&](int kc) -> int {   // dword `lane` of cuts[kc], cuts[kc + 1]; n_tiles * 8 < 2^31 (host check)
        const unsigned idx = (unsigned)kc * 8u + (unsigned)lane;
        return lane < 16 ? cut_words[idx] : 0;
    };
    // Tiles are handed out dynamically: a workgroup's first two tiles are blockIdx.x and blockIdx.x + gridDim.x, every
    // later one comes from batches of kBatch consecutive tiles drawn from a device counter.  (With a fixed stride the
    // workgroups finished between 2.2 and 3.0 ms -- tiles inside repeats cost more, and CUs are not equally fast --
    // and the kernel lasted as long as the slowest.  One draw per tile is no option either: 3e5 returning atomics on
    // one word take longer than the kernel.)  The draw for the batch after the current one is one returning atomic
    // issued by thread 0 together with the prefetch loads; it lands before pass B like every other load and crosses to
    // the other waves through LDS behind the last barrier of that iteration.
    // (batches of 2 .. 16 measured alike at human scale; a set with few tiles per workgroup gets them one by one, or the
    // last batches -- eight tiles in a row on one workgroup -- are the kernel's tail: 50 k reads took 171 us instead of 60)
    const int kBatch = a.tile_batch;
    int k = (int)blockIdx.x;                     // current tile
    int kn = k + nb;                             // next tile (its cuts are in raw_n)
    int knn = n_tiles;                           // the tile after next
    int bn = 0, be = 0;                          // rest of the current batch: tiles [bn, be)
    int next_base = 0;                           // first tile of the batch drawn last
    bool want_draw = true;                       // a draw is due (issued at the top of the next iteration)
    bool drew = false;                           // a draw was issued in this iteration
    int drawn = 0;                               // thread 0: result of the draw in flight
    FastTile cur{}, nxt{};
    FastRegs<U> g{}, gn{};
    FastReadRegs rdn{};
    int pub_r_a = 0, pub_nr = 0;                 // tile whose repeat counts are still in LDS (set 1 - p)
    int p = 0;                                   // table / start-bit set of the current tile
    int raw_n = 0, raw_nn = 0;                   // cuts of the next tile (landed) / of the tile after next (in flight)
    // (the result is not touched before the loads of the iteration are waited for anyway: used right away, wave 0 would sit
    // out the round trip AND the previous tile's coverage stores -- vmcnt counts both -- with three waves waiting at the next
    // barrier.  The counter's address passes through an empty asm: to the compiler it may differ by lane, which keeps its
    // wave-aggregation of atomics off this one -- that reads the result back at once, and one lane is all there is)
    typedef __attribute__((address_space(1))) int32_t *global_i32_ptr;   // (stays a global_ -- not a flat_ -- atomic)
    global_i32_ptr draw_from = (global_i32_ptr)a.tile_counter;
    asm volatile("" : "+v"(draw_from));
    auto draw = [&]() { if (tid == 0) drawn = __hip_atomic_fetch_add(draw_from, kBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto hand_out = [&]() -> int {               // next tile of this workgroup; switches to the drawn batch when needed
        if (bn == be) { bn = next_base; be = bn + kBatch; want_draw = true; }
        return bn++;
    };
    if (k < n_tiles) {
        int lo[NSEG], n[NSEG];
        cut_unpack<NSEG, ITER>(cut_word(cut_of(k)), cur, lo, n);
        if (cur.fast) {
            FastReadRegs rd;
            fast_issue<NSEG, U, IN>(a, tid, cur, lo, n, g, rd);
            wait_all_loads();
            stage_reads(0, cur, rd);
        }
        raw_n = cut_word(cut_of(min(kn, last_cut)));
        draw();
    }
    wait_all_loads();
    if (tid == 0) sm.next_tile = 2 * nb + drawn;
    lds_barrier();
    next_base = uni(sm.next_tile);
    knn = hand_out();                            // (this also asks for the next draw)
    while (k < n_tiles) {
        if (DIAG && tid == 0 && a.dbg) { sm.stamps[0] = __builtin_amdgcn_s_memtime(); sm.stamps[9] = __builtin_amdgcn_s_memrealtime(); }
        {
            // raw_n always holds real cuts (the index is clamped), so the unpacking needs no guard
            int lo[NSEG], n[NSEG];
            cut_unpack<NSEG, ITER>(raw_n, nxt, lo, n);
            if (kn >= n_tiles) nxt.fast = 0;
            raw_nn = cut_word(cut_of(min(knn, last_cut)));
            drew = want_draw;
            if (want_draw) { draw(); want_draw = false; }
            if (nxt.fast) fast_issue<NSEG, U, IN>(a, tid, nxt, lo, n, gn, rdn);
        }
        if (DIAG && tid == 0 && a.dbg) { sm.stamps[8] = (unsigned long long)cur.nwin; sm.stamps[12] = (unsigned long long)cur.more; sm.stamps[14] = blockIdx.x; }
        RAFT_STAMP(1);
        if (cur.fast) {
            // ---- geometry of the LDS window: slots are windows relative to a0 (16-byte aligned in cov[])
            const int nr = cur.nr, r_a = cur.r_a;
            const long long a0 = cur.g_lo & ~3LL;
            const int off0 = (int)(cur.g_lo - a0);       // first valid slot
            const int t_end = off0 + cur.nwin;           // one past the last valid slot (the sentinel slot)
            const int rows = (t_end + 1 + 255) >> 8;
            int32_t *const cov0 = OW == 4 ? a.cov + a0 : nullptr;
            constexpr bool D4 = OW == 8;           // the four-bit step encoding (pack.hpp kCovDelta4): slot p -> nibble a0 + p
            char *const covp0 = OW == 4 ? nullptr : reinterpret_cast<char *>(a.covp) + (D4 ? a0 / 2 : a0 * OW);   // slot p -> covp0 + p * OW (D4: + p / 2)
            // D4: a step is the difference array's own value -- except at the tile's first window, whose predecessor another
            // workgroup holds: that window is always listed with its value.  A lane's four steps are one aligned ushort; the
            // ushort a tile shares with its neighbour is updated with an and / or pair of atomics on its own nibbles.
            int pend_p = -1, pend_c = 0;           // D4: this lane's listed window waiting for the end of the rows (slot, value)
            auto d4_list = [&](int p, int v, int slot) {      // slot: the window's place among the tile's listed ones
                if (slot < kExcPerTile) {
                    const long long at = (long long)(EXTRA ? n_reg + k : k) * kExcPerTile + slot;
                    a.exc_pidx[at] = a0 + p; a.exc_pval[at] = v;
                } else note_exception(a, a0 + p, v);
            };
            auto d4_codes = [&](const int4 d, int c0, int c1, int c2, int c3, int p0, unsigned valid) -> unsigned {
                // four steps -> four nibbles, step + 8 where it lies within [-7, 7], else 0; the tile's first window and the
                // slots outside the tile are cleared as well.  Straight-line: half of all rows have ONE lane with a listed
                // window (a large step at a read boundary), and whatever that lane does the whole wave walks through -- so
                // the lane only finds its zero nibble and parks the window in two registers until the rows are done (one LDS
                // atomic per wave and tile then places all of them in the tile's own kExcPerTile slots, plain stores,
                // gathered by compact_exceptions_kernel).  History: a returning atomic per window on the one shared counter
                // took 12 ms for the 3.4e5 of a 36 ms chunk; an LDS atomic with its wait inside the rows 0.15 ms of the
                // kernel; a second path that recomputed the steps for the wave whenever one lane listed a window 0.2 ms.
                const unsigned ux = (unsigned)(d.x + 7), uy = (unsigned)(d.y + 7), uz = (unsigned)(d.z + 7), uw = (unsigned)(d.w + 7);
                unsigned code = (ux <= 14u ? ux + 1u : 0u) | ((uy <= 14u ? uy + 1u : 0u) << 4) | ((uz <= 14u ? uz + 1u : 0u) << 8) |
                                ((uw <= 14u ? uw + 1u : 0u) << 12);
                const unsigned f = (unsigned)(off0 - p0);                     // < 4: the tile's first window is this lane's slot f
                if (f < 4u) code &= ~(0xFu << (4u * f));
                const unsigned vmask = valid == 15u ? 0xFFFFu : (((valid & 1u) ? 0xFu : 0u) | ((valid & 2u) ? 0xF0u : 0u) | ((valid & 4u) ? 0xF00u : 0u) | ((valid & 8u) ? 0xF000u : 0u));
                code &= vmask;
                unsigned t = code | (code >> 1);
                t |= t >> 2;                                                  // bit 4q: nibble q is not zero
                unsigned esc = ~t & 0x1111u & vmask;                          // bit 4q: window q of this lane is listed
                if (esc) {
                    if (pend_p < 0 && (esc & (esc - 1u)) == 0u) {
                        const int q = (__ffs((int)esc) - 1) >> 2;
                        pend_p = p0 + q; pend_c = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
                    } else {
                        int slot = __hip_atomic_fetch_add(&sm.exc_n, (int)__popc(esc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        while (esc) {
                            const int q = (__ffs((int)esc) - 1) >> 2;
                            esc &= esc - 1u;
                            d4_list(p0 + q, q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3, slot++);
                        }
                    }
                }
                return code;
            };
            constexpr unsigned kLimit = OW == 1 ? 255u : 65535u;
            FastTables tb{sm.roff[p], sm.rlen[p], sm.rcnt[p], sm.rres[p], sm.acc_rep};

            // 1. intervals -> +1 / -1 (profileCoverage, closed form).  (Tried and measured slower: issuing the table
            //    look-ups of all slots first and predicating only the two LDS adds -- more live state, more spills.)
            int covsum = 0;
            bool bad_any = false, bad_order = false;
            if constexpr (IN == 0) {
                auto win = [&](unsigned n) -> int { return (int)(((n & win_m1) | __umulhi(n, a.div_magic)) >> win_sh); };
                auto one = [&](int rid, int st, int en) {
                    const unsigned jr = (unsigned)(rid - r_a);
                    const unsigned j = min(jr, (unsigned)nr);                         // (an empty slot would read entry nr)
                    const int b0 = tb.roff[j], nb_r = tb.roff[j + 1] - b0;
                    const int first = win((unsigned)st);
                    const int last1 = win((unsigned)(en - 1)) + 1;                    // meaningful for en >= 1
                    // valid: a record of one of the tile's reads.  Anything else refutes the sampled guess of the sorted runs
                    // this pass may be built on (engine.hip run_pass): it is flagged, must not reach the tables, and the pass
                    // is run again
                    const bool valid = jr < (unsigned)nr, sign_ok = (st | en) >= 0, pos = en > 0;
                    const bool over = last1 > first && last1 > nb_r;                   // repeat.hpp:69-72 writes past its vector
                    // (clipped to the tile's slots: a no-op for tiles of whole reads, the cut for a piece of a long read)
                    const int pf = EXTRA ? max(b0 + first, off0) : b0 + first, pl1 = EXTRA ? min(b0 + min(last1, nb_r), t_end) : b0 + min(last1, nb_r);
                    bad_any |= valid && (!sign_ok || (pos && over));
                    bad_order |= !valid;             // a record of a read outside this tile (see kErrOrder in pileup.hpp)
                    if (valid && sign_ok && pos && pf < pl1) {
                        __hip_atomic_fetch_add(&sm.diff[pf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&sm.diff[pl1], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        covsum += pl1 - pf;          // sum of coverage over the window == windows touched by its intervals
                    }
                };
                // (slots behind the first of a segment are mostly empty for three waves in four -- a segment of a HiFi tile
                // holds ~520 intervals, the third slot's lanes begin at 512 -- and an empty slot still costs its ~30 vector
                // instructions: a wave skips the slots none of its lanes holds a record in)
    #pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u < NSEG || cur.cnt[u % NSEG] > (u / NSEG) * 256 + wid * 64) one(g.rid[u], g.st[u], g.en[u]);
                }
                if (cur.more) {                      // intervals beyond the prefetched slots (dense tiles): synchronous loads
                    const TileCut d0 = cuts[cut_of(k)], d1 = cuts[cut_of(k) + 1];
    #pragma unroll
                    for (int s = 0; s < NSEG; ++s) {
                        const int n_s = d1.iv_lo[s] - d0.iv_lo[s];
                        const long long base = (long long)d0.iv_lo[s];
                        for (int i = ITER * 256 + (int)tid; i < n_s; i += 256)
                            one((a.iv_rid + base)[i], (a.iv_s + base)[i], (a.iv_e + base)[i]);
                    }
                }
                if (__ballot(bad_order) != 0ull && lane == 0) atomicOr(a.err_flags, kErrOrder);   // (every wave for itself)
                if (__ballot(bad_any) != 0ull) {     // rare: find the offending records again and report the first index
                    const TileCut d0 = cuts[cut_of(k)], d1 = cuts[cut_of(k) + 1];
                    auto is_bad = [&](int rid, int st, int en) -> bool {
                        if ((unsigned)(rid - r_a) >= (unsigned)nr) return false;
                        const int j = rid - r_a;
                        const int nb_r = tb.roff[j + 1] - tb.roff[j];
                        const int first = (int)win_of(a, (unsigned)st), last1 = (int)win_of(a, (unsigned)(en - 1)) + 1;
                        return (st | en) < 0 || (en > 0 && last1 > first && last1 > nb_r);
                    };
    #pragma unroll
                    for (int s = 0; s < NSEG; ++s) {
                        const int n_s = d1.iv_lo[s] - d0.iv_lo[s];
                        const long long base = (long long)d0.iv_lo[s];
                        for (int i = (int)tid; i < n_s; i += 256)
                            if (is_bad((a.iv_rid + base)[i], (a.iv_s + base)[i], (a.iv_e + base)[i])) raise_error(a, kErrCoord, base + i);
                    }
                }
            } else {
                // ---- window records: (first window, one past the last) in one word, the read from the caller's offsets
                const int32_t *const so_base = &sm.soff[p][0][0];
                constexpr int TABW = Smem::TAB;
                // generic look-up (tiles of more than 64 reads, records beyond the prefetched slots, the error path): the
                // last read of the tile whose records begin at or before record i of run s
                auto find_j = [&](int s, int i) -> int {
                    const int o0 = so_base[s * TABW];     // (the tile's first record in run s is the first of its first read)
                    int x = 1, y = nr;                     // first t in [1, nr) with soff[t] - soff[0] > i
                    while (x < y) {
                        const int m = (x + y) >> 1;
                        if (so_base[s * TABW + m] - o0 <= i) x = m + 1; else y = m;
                    }
                    return x - 1;
                };
                auto one_w = [&](int j, unsigned w) {
                    const int b0 = tb.roff[j], nb_r = tb.roff[j + 1] - b0;
                    const int first = (int)(w & 0xffffu), last1 = (int)(w >> 16);
                    const bool over = last1 > first && last1 > nb_r;                  // repeat.hpp:69-72 writes past its vector
                    const int pf = EXTRA ? max(b0 + first, off0) : b0 + first, pl1 = EXTRA ? min(b0 + min(last1, nb_r), t_end) : b0 + min(last1, nb_r);
                    bad_any |= over;
                    if (pf < pl1) {
                        __hip_atomic_fetch_add(&sm.diff[pf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&sm.diff[pl1], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        covsum += pl1 - pf;
                    }
                };
                // boundary lane + 1 of every run: where the records of read r_a + lane + 1 begin (entry nr closes the run)
                int bnd[NSEG];
#pragma unroll
                for (int s = 0; s < NSEG; ++s) bnd[s] = (lane < nr) ? so_base[s * TABW + lane + 1] - so_base[s * TABW] : 0x7fffffff;
                if (nr > 64) {                       // more reads than a wave has lanes for their boundaries: the table
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int s = u % NSEG, i0 = (u / NSEG) * 256 + wid * 64;
                        if (cur.cnt[s] > i0) one_w(find_j(s, min(i0 + lane, cur.cnt[s] - 1)), (unsigned)g.st[u]);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int s = u % NSEG;
                        const int i0 = (u / NSEG) * 256 + wid * 64;   // the wave's first record of this slot (in run s, from the tile's first)
                        if (cur.cnt[s] > i0) {
                            // reads of the wave's first and last record: boundaries at or before them, counted by ballot; the
                            // boundaries between decide the lanes (the first two without a loop: 64 records rarely span more than
                            // three reads.  One ballot and a loop that ends at the first boundary past the wave's last record
                            // instead of the second ballot: measured the same)
                            const int i_last = min(i0 + 63, cur.cnt[s] - 1);
                            const unsigned jf = (unsigned)__popcll(__ballot(bnd[s] <= i0)), jl = (unsigned)__popcll(__ballot(bnd[s] <= i_last));
                            const int b1 = __builtin_amdgcn_readlane(bnd[s], (int)min(jf, 63u)), b2 = __builtin_amdgcn_readlane(bnd[s], (int)min(jf + 1u, 63u));
                            int j = (int)jf + ((jf < jl && i0 + lane >= b1) ? 1 : 0) + ((jf + 1u < jl && i0 + lane >= b2) ? 1 : 0);
                            for (unsigned t = jf + 2u; t < jl; ++t) j += (i0 + lane >= __builtin_amdgcn_readlane(bnd[s], (int)t)) ? 1 : 0;
                            one_w(j, (unsigned)g.st[u]);
                        }
                    }
                }
                if (cur.more) {                      // records beyond the prefetched slots (dense tiles): synchronous loads
                    const TileCut d0 = cuts[cut_of(k)], d1 = cuts[cut_of(k) + 1];
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) {
                        const int n_s = d1.iv_lo[s] - d0.iv_lo[s];
                        const long long base = (long long)d0.iv_lo[s];
                        for (int i = ITER * 256 + (int)tid; i < n_s; i += 256) one_w(find_j(s, i), (a.iv_w + base)[i]);
                    }
                }
                if (__ballot(bad_any) != 0ull) {     // rare: find the offending records again and report the first index
                    const TileCut d0 = cuts[cut_of(k)], d1 = cuts[cut_of(k) + 1];
#pragma unroll
                    for (int s = 0; s < NSEG; ++s) {
                        const int n_s = d1.iv_lo[s] - d0.iv_lo[s];
                        const long long base = (long long)d0.iv_lo[s];
                        for (int i = (int)tid; i < n_s; i += 256) {
                            const int j = find_j(s, i);
                            const unsigned w = (a.iv_w + base)[i];
                            const int first = (int)(w & 0xffffu), last1 = (int)(w >> 16);
                            if (last1 > first && last1 > tb.roff[j + 1] - tb.roff[j]) raise_error(a, kErrCoord, base + i);
                        }
                    }
                }
            }
            lane_cov += covsum;
            RAFT_STAMP(2);
            lds_barrier();
            RAFT_STAMP(3);

            // 2. pass A: per-wave sums of the difference array (each wave owns rpw contiguous rows)
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
            if (drew && tid == 0) sm.next_tile = 2 * nb + drawn;   // read by every wave behind barrier C
            // The other table set is free now (its tile's runs were emitted before the last barrier pair): publish
            // that tile's repeat counts, then stage the next tile's reads in it.
            publish_counts(1 - p, pub_r_a, pub_nr);
            if (nxt.fast) stage_reads(1 - p, nxt, rdn);

            // 3. pass B: prefix sum, store, run detection; each row is zeroed for the next tile once it is read
            int carry;
            {
                const int4 ws = *reinterpret_cast<const int4 *>(&sm.wsum[0]);   // one LDS read, the same in every lane
                const int w0 = uni(ws.x), w1 = uni(ws.y), w2 = uni(ws.z);
                carry = (wid > 0 ? w0 : 0) + (wid > 1 ? w1 : 0) + (wid > 2 ? w2 : 0);
            }
            bool hp = (wid > 0) && (row_b < rows) && (carry >= a.high_cov); // window before this wave's first slot is high
            const bool hp_in = hp;
            int S = hp ? kOpen : kNone;  // start slot of the run currently open
            int pclose = -1;             // slot at which the run inherited from before this wave closed
            int nq = 0;                  // runs this wave has parked for emission
            const int full_b = (off0 + 255) >> 8, full_e = t_end >> 8;   // rows [full_b, full_e) hold valid slots only
            const uint32_t partial_rows = ((1u << full_b) - 1u) | ~((1u << full_e) - 1u);   // rows 0 .. 31, full_e <= 31
            // first slots of the tile's reads, two per lane (a run never continues across a read boundary,
            // repeat.hpp:111-112); also used to find the owner of a parked run
            const int ro0 = (lane < nr) ? tb.roff[lane] : 0x7fffffff;
            const int ro1 = (lane + 64 < nr) ? tb.roff[lane + 64] : 0x7fffffff;
            const int piece = EXTRA ? cur.piece : 0;
            auto park = [&](int sS, int sT) {    // wave-uniform arguments
                // repeat.hpp:125,150 -- except that a run touching an edge of a PIECE of a long read may continue in the
                // neighbouring piece: it is kept whatever its length and judged when finalize has joined the pieces
                if ((long long)(sT - sS) * a.reso < (long long)a.repeat_length && !(piece && (sS == off0 || sT == t_end))) return;
                if (nq < kRunQ) {
                    if (lane == 0) { sm.runq[(wid * kRunQ + nq) * 2] = sS; sm.runq[(wid * kRunQ + nq) * 2 + 1] = sT; }
                    ++nq;
                } else if (lane == 0) { if (piece) emit_piece_run(a, tb, r_a, sS, sT); else emit_run(a, tb, nr, sS, sT); }
            };

            int4 dn = make_int4(0, 0, 0, 0);
            if (row_b < row_e) dn = *reinterpret_cast<const int4 *>(&sm.diff[row_b * 256 + lane * 4]);
#ifdef RAFT_ABLATE_ROWS
            for (int row = row_b; row < row_e; ++row) *reinterpret_cast<int4 *>(&sm.diff[row * 256 + lane * 4]) = make_int4(0, 0, 0, 0);
            if (0)
#endif
            for (int row = row_b; row < row_e; ++row) {
                const int base = row * 256, p0 = base + lane * 4;
                const int4 d = dn;
                // the next row, unconditionally: past the wave's last row this reads a row another wave owns (or the
                // tables behind the array) and the value is never used
                dn = *reinterpret_cast<const int4 *>(&sm.diff[p0 + 256]);
                *reinterpret_cast<int4 *>(&sm.diff[p0]) = make_int4(0, 0, 0, 0);
                // prefix sum of the row's 256 slots
                const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
                const int incl = wave_incl_scan_add(w);
                const int excl = incl - w + carry;
                carry += __builtin_amdgcn_readlane(incl, 63);
                const int c0 = excl + x, c1 = excl + y, c2 = excl + z, c3 = excl + w;
                unsigned long long M0, M1, M2, M3;
                // (coverage is never negative -- every interval's +1 / -1 balance out -- so one unsigned compare of the OR finds
                // any value at or above the limit)
                const unsigned k0 = min((unsigned)c0, kLimit), k1 = min((unsigned)c1, kLimit), k2 = min((unsigned)c2, kLimit), k3 = min((unsigned)c3, kLimit);
                const bool big = OW != 4 && ((unsigned)c0 | (unsigned)c1 | (unsigned)c2 | (unsigned)c3) >= kLimit;
                if (((partial_rows >> row) & 1u) == 0u) {
                    // scalar base + this lane's 32-bit byte offset (the LDS address of the row): no 64-bit address arithmetic
                    if (OW == 4) *reinterpret_cast<int4 *>(reinterpret_cast<char *>(cov0) + (unsigned)p0 * 4u) = make_int4(c0, c1, c2, c3);
                    else if (OW == 1) *reinterpret_cast<unsigned *>(covp0 + (unsigned)p0) = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24);
                    else if (D4) {
                        *reinterpret_cast<uint16_t *>(covp0 + ((unsigned)p0 >> 1)) = (uint16_t)d4_codes(d, c0, c1, c2, c3, p0, 15u);
                        if ((((unsigned)a0 + (unsigned)p0 + (unsigned)a.d4_shift) & 1023u) == 0u) a.cov_anchor[(a0 + p0 + a.d4_shift) >> 10] = excl;
                    }
                    else *reinterpret_cast<uint2 *>(covp0 + (unsigned)p0 * 2u) = make_uint2(k0 | (k1 << 16), k2 | (k3 << 16));
                    if (big && !D4) {
                        if ((unsigned)c0 >= kLimit) note_exception(a, a0 + p0, c0);
                        if ((unsigned)c1 >= kLimit) note_exception(a, a0 + p0 + 1, c1);
                        if ((unsigned)c2 >= kLimit) note_exception(a, a0 + p0 + 2, c2);
                        if ((unsigned)c3 >= kLimit) note_exception(a, a0 + p0 + 3, c3);
                    }
                    M0 = __ballot(c0 >= a.high_cov); M1 = __ballot(c1 >= a.high_cov);
                    M2 = __ballot(c2 >= a.high_cov); M3 = __ballot(c3 >= a.high_cov);
                } else {                         // first / last row of the tile: some slots lie outside [off0, t_end)
                    const unsigned q0 = (unsigned)(p0 - off0), nw_u = (unsigned)cur.nwin;
                    const bool v0 = q0 < nw_u, v1 = q0 + 1u < nw_u, v2 = q0 + 2u < nw_u, v3 = q0 + 3u < nw_u;
                    if (OW == 4) {
                        if (v0 && v3) *reinterpret_cast<int4 *>(&cov0[p0]) = make_int4(c0, c1, c2, c3);
                        else {
                            if (v0) cov0[p0 + 0] = c0;
                            if (v1) cov0[p0 + 1] = c1;
                            if (v2) cov0[p0 + 2] = c2;
                            if (v3) cov0[p0 + 3] = c3;
                        }
                    } else if (D4) {
                        const unsigned valid = (v0 ? 1u : 0u) | (v1 ? 2u : 0u) | (v2 ? 4u : 0u) | (v3 ? 8u : 0u);
                        if (valid) {
                            const unsigned code = d4_codes(d, c0, c1, c2, c3, p0, valid);
                            char *const o = covp0 + ((unsigned)p0 >> 1);
                            if (valid == 15u) *reinterpret_cast<uint16_t *>(o) = (uint16_t)code;
                            else {      // the neighbouring tile owns the other nibbles of this ushort: clear mine, then set them
                                const unsigned mask = (v0 ? 0xFu : 0u) | (v1 ? 0xF0u : 0u) | (v2 ? 0xF00u : 0u) | (v3 ? 0xF000u : 0u);
                                const unsigned long long addr = reinterpret_cast<unsigned long long>(o);
                                unsigned *const word = reinterpret_cast<unsigned *>(addr & ~3ull);
                                const unsigned sh = (unsigned)(addr & 2ull) * 8u;
                                atomicAnd(word, ~(mask << sh));
                                atomicOr(word, (code & mask) << sh);
                            }
                            if (v0 && (((unsigned)a0 + (unsigned)p0 + (unsigned)a.d4_shift) & 1023u) == 0u) a.cov_anchor[(a0 + p0 + a.d4_shift) >> 10] = excl;
                        }
                    } else if (OW == 1) {
                        uint8_t *const o = reinterpret_cast<uint8_t *>(covp0) + p0;
                        if (v0 && v3) *reinterpret_cast<unsigned *>(o) = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24);
                        else {          // (the neighbouring tile owns the other bytes of this dword)
                            if (v0) o[0] = (uint8_t)k0;
                            if (v1) o[1] = (uint8_t)k1;
                            if (v2) o[2] = (uint8_t)k2;
                            if (v3) o[3] = (uint8_t)k3;
                        }
                    } else {
                        uint16_t *const o = reinterpret_cast<uint16_t *>(covp0) + p0;
                        if (v0 && v3) *reinterpret_cast<uint2 *>(o) = make_uint2(k0 | (k1 << 16), k2 | (k3 << 16));
                        else {
                            if (v0) o[0] = (uint16_t)k0;
                            if (v1) o[1] = (uint16_t)k1;
                            if (v2) o[2] = (uint16_t)k2;
                            if (v3) o[3] = (uint16_t)k3;
                        }
                    }
                    if (big && !D4) {
                        if (v0 && (unsigned)c0 >= kLimit) note_exception(a, a0 + p0, c0);
                        if (v1 && (unsigned)c1 >= kLimit) note_exception(a, a0 + p0 + 1, c1);
                        if (v2 && (unsigned)c2 >= kLimit) note_exception(a, a0 + p0 + 2, c2);
                        if (v3 && (unsigned)c3 >= kLimit) note_exception(a, a0 + p0 + 3, c3);
                    }
                    M0 = __ballot(v0 && c0 >= a.high_cov); M1 = __ballot(v1 && c1 >= a.high_cov);
                    M2 = __ballot(v2 && c2 >= a.high_cov); M3 = __ballot(v3 && c3 >= a.high_cov);
                }
#ifdef RAFT_ABLATE_SCAN
                continue;
#endif
                if ((M0 | M1 | M2 | M3) == 0ull && !hp) continue;   // the common case: no high window in the row
                const unsigned long long RS0 = __ballot((ro0 >> 8) == row), RS1 = __ballot((ro1 >> 8) == row); // reads starting here
                const bool tail = base + 256 > t_end;               // the row holds slots past the end of the tile
                const unsigned long long A = M0 & M1 & M2 & M3;     // lanes whose four slots are all high
                // a run passes through the whole row (all high, no read begins, no tile end)
                if (hp && !tail && (RS0 | RS1) == 0ull && A == ~0ull) continue;

                // ---- reads that begin in the row: bit k of sbm <=> a read begins at this lane's slot k (a run never continues
                // across a read boundary, repeat.hpp:111-112)
                int sbm = 0;
                auto mark_starts = [&](unsigned long long rs, int ro) {
                    while (rs) {
                        const int pos = __builtin_amdgcn_readlane(ro, (int)__builtin_ctzll(rs)) & 255;
                        rs &= rs - 1ull;
                        if ((pos >> 2) == lane) sbm |= 1 << (pos & 3);
                    }
                };
                mark_starts(RS0, ro0);
                mark_starts(RS1, ro1);

                // ---- run scan of the row, per lane: every lane knows which of its four slots is high, begins a read, or lies
                // inside the tile; what it lacks -- is the slot before mine high, and where did the run that reaches me begin --
                // comes from one wave shift and one max-scan.  (The first version did this as scalar bit logic on twenty 64-bit
                // masks -- 250-300 scalar instructions per row under the kernel's worst register pressure -- behind a filter
                // that recognised by whole lanes the rows in which no run long enough to be kept could end; with the scan at
                // ~60 vector instructions the filter costs more than it saves.)
                {
                    // the lane's four slots as four bits: high, read begins, previous slot high, run starts, run ends before
                    const int hk = (row == 0) ? off0 : 0;     // the carried-in "previous slot is high" belongs to the first valid slot
                    const int hvn = (int)((M0 >> lane) & 1ull) | ((int)((M1 >> lane) & 1ull) << 1) | ((int)((M2 >> lane) & 1ull) << 2) |
                                    ((int)((M3 >> lane) & 1ull) << 3);
                    int prvn = ((hvn << 1) & 15) | wave_shr1(hvn >> 3, 0);
                    if (hp && lane == 0) prvn |= 1 << hk;
                    int inside = 15;
                    if (tail) inside = (1 << min(max(t_end - p0, 0), 4)) - 1;
                    const int starts = hvn & ((prvn ^ 15) | sbm);
                    int ends = prvn & ((hvn ^ 15) | sbm) & inside;
                    const int ls = starts ? p0 + (31 - __clz(starts)) : -0x40000000;      // last slot of this lane at which a run starts
                    const int incl = wave_incl_scan_max(ls, -0x40000000);
                    const int carried = max(S, wave_shr1(incl, -0x40000000));      // start of the run that reaches this lane's first slot
                    // ends of runs: every lane takes its first, then (rarely) its second
                    while (__ballot(ends != 0) != 0ull) {
                        const bool has = ends != 0;
                        const int k = has ? __builtin_ctz(ends) : 0;
                        const int below = starts & ((1 << k) - 1);                 // starts in this lane before slot k
                        const int best = max(carried, below ? p0 + (31 - __clz(below)) : -0x40000000);
                        const int t = p0 + k;
                        ends &= ends - 1;
                        const unsigned long long inh = __ballot(has && best == kOpen);
                        if (inh) pclose = __builtin_amdgcn_readlane(t, (int)__builtin_ctzll(inh));   // the inherited run: its start is known after the barrier
                        // repeat.hpp:125,150 -- except at the edges of a PIECE of a long read (see park())
                        const bool keep = has && best >= 0 &&
                                          ((long long)(t - best) * a.reso >= (long long)a.repeat_length || (piece && (best == off0 || t == t_end)));
                        const unsigned long long km = __ballot(keep);
                        if (km) {
                            const int idx = nq + (int)__popcll(km & ((1ull << lane) - 1ull));
                            if (keep) {
                                if (idx < kRunQ) { sm.runq[(wid * kRunQ + idx) * 2] = best; sm.runq[(wid * kRunQ + idx) * 2 + 1] = t; }
                                else if (piece) emit_piece_run(a, tb, r_a, best, t);
                                else emit_run(a, tb, nr, best, t);
                            }
                            nq = min(kRunQ, nq + (int)__popcll(km));
                        }
                    }
                    S = max(S, __builtin_amdgcn_readlane(incl, 63));
                    if (!tail) hp = (M3 >> 63) != 0ull;
                    else if (t_end > base) {
                        const int tl = t_end - 1 - base; // last valid slot of the row
                        const unsigned long long Mk = (tl & 3) == 0 ? M0 : (tl & 3) == 1 ? M1 : (tl & 3) == 2 ? M2 : M3;
                        hp = ((Mk >> (tl >> 2)) & 1ull) != 0ull;
                    }
                }
            }

            if (D4) {                            // the windows the lanes parked: one place in the tile's list each
                const unsigned long long pm = __ballot(pend_p >= 0);
                if (pm) {
                    int base = 0;
                    if (lane == 0) base = __hip_atomic_fetch_add(&sm.exc_n, (int)__popcll(pm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (pend_p >= 0) d4_list(pend_p, pend_c, base + (int)__popcll(pm & ((1ull << lane) - 1ull)));
                }
            }
            // 4. publish the wave's seam state
            if (lane == 0) {
                *reinterpret_cast<int4 *>(&sm.wst[wid * 8]) = make_int4(row_e > row_b ? 1 : 0, pclose, S, hp ? 1 : 0);
                sm.wst[wid * 8 + 4] = hp_in ? 1 : 0;
            }
            RAFT_STAMP(5);
            lds_barrier();
            RAFT_STAMP(6);
            if (D4 && tid == 0) {                    // (the next tile lists its windows two barriers from here)
                const int n_listed = sm.exc_n;
                if (n_listed) { a.exc_tile_n[EXTRA ? n_reg + k : k] = min(n_listed, kExcPerTile); sm.exc_n = 0; }
            }

            // 5. seams, resolved by every wave for itself: a run inherited from earlier waves starts at the run start of
            //    the nearest earlier wave that saw one; the wave holding the last valid slot closes the run that
            //    reaches the end of the tile (the end of a read closes a run, repeat.hpp:150)
            {
                const int v = (lane < NW * 8) ? sm.wst[lane] : 0;
                auto run_start_before = [&](int w) -> int { // start slot of the run open at the end of wave w
#pragma unroll
                    for (int y = NW - 1; y >= 0; --y) {
                        if (y > w) continue;
                        if (!__builtin_amdgcn_readlane(v, y * 8 + 0)) continue;
                        const int sf = __builtin_amdgcn_readlane(v, y * 8 + 2);
                        if (sf != kOpen) return sf;
                    }
                    return -1;
                };
                if (row_e > row_b) {
                    if (hp_in && pclose >= 0) { const int b = run_start_before(wid - 1); if (b >= 0) park(b, pclose); }
                    if (row_e == rows && hp) {           // this wave holds the last valid slot
                        const int b = (S != kOpen) ? S : (wid == 0 ? -1 : run_start_before(wid - 1));
                        if (b >= 0) park(b, t_end);
                    }
                }
            }
            RAFT_STAMP(11);

            // 6. every run this wave parked becomes a repeat record, one lane per run
            if (nq > 0) {
                int sS = 0, sT = 0, j = 0;
                if (lane < nq) { sS = sm.runq[(wid * kRunQ + lane) * 2]; sT = sm.runq[(wid * kRunQ + lane) * 2 + 1]; }
                // owner of each run: the number of reads that begin at or before its first slot (one compare + ballot)
#pragma unroll 1
                for (int q = 0; q < nq; ++q) {
                    const int s0 = __builtin_amdgcn_readlane(sS, q);
                    const int jq = __popcll(__ballot(ro0 <= s0)) + __popcll(__ballot(ro1 <= s0)) - 1;
                    if (lane == q) j = jq;
                }
                // (LS finds runs in the concatenated windows: one that spans a read boundary is two runs, repeat.hpp:111-112)
                if (lane < nq) {
                    if (piece) emit_piece_run(a, tb, r_a, sS, sT);
                    else
                        while (sS < sT) {
                            const int e = min(sT, tb.roff[j + 1]);
                            if (e > sS) emit_run_of(a, tb, j, sS, e);
                            sS = max(sS, e); ++j;
                        }
                }
            }
            RAFT_STAMP(13);
            pub_r_a = r_a; pub_nr = nr;          // counts are final once every wave is past its next barrier
        } else {
            // no work in this tile: keep the hand-over of the table sets going
            lds_barrier();                       // the previous tile's runs are all emitted
            wait_all_loads();
            if (drew && tid == 0) sm.next_tile = 2 * nb + drawn;
            publish_counts(1 - p, pub_r_a, pub_nr);
            if (nxt.fast) stage_reads(1 - p, nxt, rdn);
            lds_barrier();
            pub_nr = 0;
        }
        RAFT_STAMP(7);
        if (DIAG && tid == 0 && a.dbg) {
            sm.stamps[10] = __builtin_amdgcn_s_memrealtime();
            if (!EXTRA) for (int i = 0; i < 16; ++i) a.dbg[(long long)k * 16 + i] = sm.stamps[i];
        }
        if (drew) next_base = uni(sm.next_tile);     // (written before this iteration's last barrier)
        k = kn; kn = knn; knn = hand_out();
        cur = nxt; g = gn; p = 1 - p; raw_n = raw_nn;
    }
    lds_barrier();                               // the last tile's runs are all emitted
    publish_counts(1 - p, pub_r_a, pub_nr);
    {
        const long long cs = wave_reduce_add64(lane_cov);
        if (lane == 0 && cs) atomicAdd(&sm.acc_cov, (unsigned long long)cs);
    }
    lds_barrier();
    if (tid == 0) {
        a.block_sums[2 * (long long)blockIdx.x] = (long long)sm.acc_cov;
        a.block_sums[2 * (long long)blockIdx.x + 1] = (long long)sm.acc_rep;
    }
}

// EXTRA: 0 = the regular tiles; 1 = the extra tiles (a launch of its own, beside the regular one on a second stream:
// rounds 1-2, kept for A/B and for the diagnostic build); 2 = both from ONE persistent grid -- a workgroup that finds no
// regular tile left goes on with the extra ones.  (Two launches on two streams cost the pass an event hand-over at either
// end, ~25 us; on a long-read set, where a fifth of the windows sit in extra tiles, those used to start only when regular
// workgroups retired.)
template <int CAP, int NSEG, int U, int MINW, bool DIAG, int EXTRA, int OW = 4, bool LS = false, int IN = 0>
__global__ __launch_bounds__(256, MINW) void pileup_fast_kernel(const TileCut *__restrict__ cuts, PileupArgs a)
{
    using Smem = FastSmem<CAP, IN ? NSEG : 0>;
    // Measured on MI355X (tools/occ_probe.hip, tools/stamp_probe.py): five 31,744-byte workgroups share a CU, five
    // 31,856-byte ones do not -- a fifth of the persistent grid then starts only when the first workgroups retire and
    // the kernel takes a third longer.  Stay at or below the footprint that is known to fit.
    static_assert(sizeof(Smem) * MINW <= 31328 * 5, "LDS footprint does not allow MINW workgroups per CU");
    __shared__ __attribute__((aligned(16))) Smem sm;
    if (EXTRA != 1) fast_tile_loop<CAP, NSEG, U, DIAG, false, OW, LS, IN>(sm, cuts, a);
    if (EXTRA == 1) fast_tile_loop<CAP, NSEG, U, DIAG, true, OW, LS, IN>(sm, cuts, a);
    if (EXTRA == 2) {
        PileupArgs b = a;                            // the extra tiles: their own hand-out counter, one at a time; sums behind the regular ones'
        b.tile_counter = a.slow_counter; b.tile_batch = 1; b.block_sums = a.block_sums + 2 * (long long)gridDim.x;
        lds_barrier();
        fast_tile_loop<CAP, NSEG, U, DIAG, true, OW, LS, IN>(sm, cuts, b);
    }
}

} // namespace raft
