// pileup_wave.hpp -- barrier-free form of the dominant kernel: ONE WAVE owns one tile.
//
// Measured on MI355X (tools/stamp_probe.py, profiles/r01_*): with a 4-wave workgroup per
// tile the tile's lifetime was ~35 k cycles whatever the tile size -- descriptor and
// interval load latency, seven workgroup barriers, a serial stitch of runs across wave
// seams -- and only ~30 % of it issued coverage stores.  Here a wavefront is the unit:
//   * persistent waves (grid = CUs x resident workgroups) walk tiles k = wave, wave+W, ...
//   * the tile descriptor (one 72-byte record: reads, windows, interval ranges) of the
//     NEXT tile is fetched with scalar loads, and the next tile's read offsets and first
//     256 intervals are prefetched into registers, while the current tile is processed;
//   * the wave's private LDS slice holds the difference array; LDS operations of one wave
//     execute in order, so no barrier is needed anywhere: clear -> +1/-1 -> row scan;
//   * rows of 256 windows are prefix-summed (DPP) with a scalar carry and stored as
//     aligned 1 KiB wave-stores; run detection uses the row's four 64-bit ballots with
//     the run state (open run start, previous-window-high) carried in scalars, so there
//     are no seams to stitch.
// Semantics are those of pileup.hpp (same closed forms, same error reporting).
#pragma once
#include "pileup.hpp"

namespace raft {

struct SegStarts { long long start[kMaxSeg + 1]; int32_t n_seg; };

struct TileDesc {              // written by tile_desc_kernel, read with scalar loads
    int32_t r_lo, r_hi;        // reads [r_lo, r_hi) start in this tile
    int32_t n_iv[kMaxSeg];     // intervals of those reads in segment s
    long long g_lo, g_hi;      // their windows [g_lo, g_hi) in cov[]
    long long iv_lo[kMaxSeg];  // first interval in segment s
};

template <int CAPW>
struct WaveSmem {
    static constexpr int SLOTS = CAPW + 256;
    static constexpr int SBW = SLOTS / 32;
    static constexpr int MAXR = 126;
    static_assert(SLOTS <= 2048, "sbits are cleared by one instruction per wave");
    int32_t diff[SLOTS];
    uint32_t sbits[SBW];
    int32_t roff[MAXR + 2];
};

struct WaveAcc {               // per-lane partial sums, reduced once per wave at kernel end
    long long cov;
    long long rep;
};

// Compiler-level ordering point between phases that communicate through LDS across lanes of
// the same wave (the hardware executes one wave's LDS operations in order).
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void emit_run_wave(const PileupArgs &a, WaveAcc &acc, int r_a, int r_b, bool single_read,
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
    acc.rep += end - start;                        // repeat.hpp:127,152
}

struct IvRegs { int rid[4], st[4], en[4]; long long cv; };

// v-th interval of a window whose segments are (seg_lo[s], seg_cum[s]..seg_cum[s+1])
__device__ __forceinline__ long long iv_index_of(const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1], int v)
{
    long long idx = seg_lo[0] + v;
#pragma unroll
    for (int s = 1; s < kMaxSeg; ++s)
        if (v >= seg_cum[s]) idx = seg_lo[s] + (v - seg_cum[s]);
    return idx;
}

// Issues the loads a window needs first: lane j's read offset and intervals v = lane + 64u.
__device__ __forceinline__ void issue_window_loads(const PileupArgs &a, int lane, int r_a, int nr,
                                                   const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1],
                                                   IvRegs &g)
{
    g.cv = (lane <= nr) ? a.cov_off[r_a + lane] : 0;
    const int n_iv = seg_cum[kMaxSeg];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int v = lane + u * 64;
        const bool ok = v < n_iv;
        const long long i = ok ? iv_index_of(seg_lo, seg_cum, v) : 0;
        g.rid[u] = ok ? a.iv_rid[i] : -1;
        g.st[u] = ok ? a.iv_s[i] : 0;
        g.en[u] = ok ? a.iv_e[i] : 0;
    }
}

// Run state carried by the wave across rows and chunks (uniform values).
struct RunState {
    long long open;  // global window index where the currently open run began (valid when hp)
    bool hp;         // the window just before the next slot is high and belongs to the same read
};

// One LDS window of one wave: global windows [w_lo, w_hi) (<= CAPW) of reads [r_a, r_b).
template <int CAPW>
__device__ void wave_window(const PileupArgs &a, WaveSmem<CAPW> &ws, WaveAcc &acc, RunState &rs, int lane,
                            int r_a, int r_b, long long w_lo, long long w_hi, bool single_read, bool first_chunk,
                            bool last_chunk, const long long (&seg_lo)[kMaxSeg], const int (&seg_cum)[kMaxSeg + 1],
                            IvRegs &g)
{
    using Smem = WaveSmem<CAPW>;
    const long long a0 = w_lo & ~3LL;
    const int off0 = (int)(w_lo - a0);
    const int t_end = off0 + (int)(w_hi - w_lo);
    const int rows = (t_end + 1 + 255) >> 8;
    const int nr = r_b - r_a;
    const bool use_tab = nr <= Smem::MAXR;
    const int n_iv = seg_cum[kMaxSeg];

    // 1. clear, stage read offsets
    for (int i = lane * 4; i < rows * 256; i += 256)
        *reinterpret_cast<int4 *>(&ws.diff[i]) = make_int4(0, 0, 0, 0);
    if (lane < rows * 8) ws.sbits[lane] = 0u;       // rows*8 <= 64 because SLOTS <= 2048
    if (use_tab) {
        if (lane <= nr) ws.roff[lane] = (int)(g.cv - a0);
        for (int j = lane + 64; j <= nr; j += 64) ws.roff[j] = (int)(a.cov_off[r_a + j] - a0);
    }
    wave_lds_order();

    // 2. read-start bits
    if (single_read) {
        if (first_chunk && lane == 0) ws.sbits[0] = 1u << off0;
    } else {
        for (int j = lane; j < nr; j += 64) {
            const int p = use_tab ? ws.roff[j] : (int)(a.cov_off[r_a + j] - a0);
            atomicOr(&ws.sbits[p >> 5], 1u << (p & 31));
        }
    }

    // 3. intervals -> +1 / -1
    for (int v0 = lane; v0 < n_iv; v0 += 256) {
        if (v0 != lane) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int v = v0 + u * 64;
                const bool ok = v < n_iv;
                const long long i = ok ? iv_index_of(seg_lo, seg_cum, v) : 0;
                g.rid[u] = ok ? a.iv_rid[i] : -1;
                g.st[u] = ok ? a.iv_s[i] : 0;
                g.en[u] = ok ? a.iv_e[i] : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (g.rid[u] < 0) continue;
            if ((g.st[u] | g.en[u]) < 0) { raise_error(a, kErrCoord, iv_index_of(seg_lo, seg_cum, v0 + u * 64)); continue; }
            const int first = (int)win_of(a, (unsigned)g.st[u]);
            int last = (g.en[u] > 0) ? (int)win_of(a, (unsigned)(g.en[u] - 1)) : -1;
            if (last < first) continue;
            int b0, nb_r;
            if (use_tab) { const int j = g.rid[u] - r_a; b0 = ws.roff[j]; nb_r = ws.roff[j + 1] - b0; }
            else {
                const long long c0 = a.cov_off[g.rid[u]];
                b0 = (int)(c0 - a0); nb_r = (int)(a.cov_off[g.rid[u] + 1] - c0);
            }
            if (last >= nb_r) {             // reference writes past its vector here (repeat.hpp:69-72)
                raise_error(a, kErrCoord, iv_index_of(seg_lo, seg_cum, v0 + u * 64));
                last = nb_r - 1;
                if (last < first) continue;
            }
            int pf = b0 + first, pl1 = b0 + last + 1;
            if (single_read) {
                pf = max(pf, off0);
                pl1 = min(pl1, t_end);
                if (pf >= pl1) continue;
            }
            __hip_atomic_fetch_add(&ws.diff[pf], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&ws.diff[pl1], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    wave_lds_order();

    // 4. rows: prefix sum, store, run detection (all state carried in scalars)
    int carry = 0;
    bool hp = (single_read && !first_chunk) ? rs.hp : false;
    long long open = rs.open;
    long long covsum = 0;
    for (int row = 0; row < rows; ++row) {
        const int base = row * 256;
        const int p0 = base + lane * 4;
        const int4 d = *reinterpret_cast<const int4 *>(&ws.diff[p0]);
        const int x = d.x, y = x + d.y, z = y + d.z, w = z + d.w;
        const int incl = wave_incl_scan_add(w);
        const int excl = incl - w + carry;
        carry += __builtin_amdgcn_readlane(incl, 63);
        const int c0 = excl + x, c1 = excl + y, c2 = excl + z, c3 = excl + w;
        // validity of the lane's four slots: off0 <= p0+k < t_end, as one unsigned compare each; the ballots land
        // in SGPR pairs and are combined with scalar ANDs (no control flow between their definition and their use)
        const bool full = (base >= off0) && (base + 256 <= t_end);
        const unsigned q0 = (unsigned)(p0 - off0), nbw_u = (unsigned)(t_end - off0);
        const unsigned long long VA0 = __ballot(q0 + 0u < nbw_u), VA1 = __ballot(q0 + 1u < nbw_u),
                                 VA2 = __ballot(q0 + 2u < nbw_u), VA3 = __ballot(q0 + 3u < nbw_u);
        const unsigned long long M0 = __ballot(c0 >= a.high_cov) & VA0, M1 = __ballot(c1 >= a.high_cov) & VA1,
                                 M2 = __ballot(c2 >= a.high_cov) & VA2, M3 = __ballot(c3 >= a.high_cov) & VA3;
        if (full) {
            *reinterpret_cast<int4 *>(&a.cov[a0 + p0]) = make_int4(c0, c1, c2, c3);
            covsum += (long long)(c0 + c1 + c2 + c3);
        } else {
            if ((p0 + 0 >= off0) && (p0 + 0 < t_end)) { a.cov[a0 + p0 + 0] = c0; covsum += c0; }
            if ((p0 + 1 >= off0) && (p0 + 1 < t_end)) { a.cov[a0 + p0 + 1] = c1; covsum += c1; }
            if ((p0 + 2 >= off0) && (p0 + 2 < t_end)) { a.cov[a0 + p0 + 2] = c2; covsum += c2; }
            if ((p0 + 3 >= off0) && (p0 + 3 < t_end)) { a.cov[a0 + p0 + 3] = c3; covsum += c3; }
        }
        if ((M0 | M1 | M2 | M3) == 0ull && !hp) continue;

        const unsigned long long VE0 = __ballot(p0 + 0 < t_end), VE1 = __ballot(p0 + 1 < t_end),
                                 VE2 = __ballot(p0 + 2 < t_end), VE3 = __ballot(p0 + 3 < t_end);
        const uint32_t word = ws.sbits[p0 >> 5];
        const uint32_t nib = (word >> (p0 & 31)) & 0xFu;
        const unsigned long long SB0 = __ballot(nib & 1u), SB1 = __ballot(nib & 2u),
                                 SB2 = __ballot(nib & 4u), SB3 = __ballot(nib & 8u);
        const unsigned long long hb = hp ? 1ull : 0ull;
        const int hk = (row == 0) ? off0 : 0;
        const unsigned long long P0 = (M3 << 1) | (hk == 0 ? hb : 0ull), P1 = M0 | (hk == 1 ? hb : 0ull),
                                 P2 = M1 | (hk == 2 ? hb : 0ull), P3 = M2 | (hk == 3 ? hb : 0ull);
        const unsigned long long CL0 = P0 & (~M0 | SB0) & VE0, CL1 = P1 & (~M1 | SB1) & VE1,
                                 CL2 = P2 & (~M2 | SB2) & VE2, CL3 = P3 & (~M3 | SB3) & VE3;
        const unsigned long long CA0 = M0 & (~P0 | SB0), CA1 = M1 & (~P1 | SB1),
                                 CA2 = M2 & (~P2 | SB2), CA3 = M3 & (~P3 | SB3);
        if ((CL0 | CL1 | CL2 | CL3) != 0ull) {
            const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
            unsigned cl4 = (unsigned)((CL0 >> lane) & 1ull) | (unsigned)(((CL1 >> lane) & 1ull) << 1) |
                           (unsigned)(((CL2 >> lane) & 1ull) << 2) | (unsigned)(((CL3 >> lane) & 1ull) << 3);
#pragma unroll 1
            while (cl4) {
                const int k = __builtin_ctz(cl4);
                cl4 &= cl4 - 1u;
                long long best = open;           // latest run start before (lane, k): carried, or in this row
                int bs = -1;
                unsigned long long m;
                m = CA0 & (0 < k ? le : lt); if (m) bs = max(bs, 4 * top_bit(m) + 0);
                m = CA1 & (1 < k ? le : lt); if (m) bs = max(bs, 4 * top_bit(m) + 1);
                m = CA2 & (2 < k ? le : lt); if (m) bs = max(bs, 4 * top_bit(m) + 2);
                m = CA3 & lt;                if (m) bs = max(bs, 4 * top_bit(m) + 3);
                if (bs >= 0) best = a0 + base + bs;
                emit_run_wave(a, acc, r_a, r_b, single_read, best, a0 + p0 + k);
            }
        }
        int ns = -1;
        if (CA0) ns = max(ns, 4 * top_bit(CA0) + 0);
        if (CA1) ns = max(ns, 4 * top_bit(CA1) + 1);
        if (CA2) ns = max(ns, 4 * top_bit(CA2) + 2);
        if (CA3) ns = max(ns, 4 * top_bit(CA3) + 3);
        if (ns >= 0) open = a0 + base + ns;
        if (base + 256 <= t_end) hp = (M3 >> 63) != 0ull;
        else if (t_end > base) {
            const int tl = t_end - 1 - base;
            const unsigned long long Mk = (tl & 3) == 0 ? M0 : (tl & 3) == 1 ? M1 : (tl & 3) == 2 ? M2 : M3;
            hp = ((Mk >> (tl >> 2)) & 1ull) != 0ull;
        }
    }
    acc.cov += covsum;
    // 5. end of window: the end of the read closes the run (repeat.hpp:150); a chunk seam carries it
    if (hp && last_chunk) {
        if (lane == 0) emit_run_wave(a, acc, r_a, r_b, single_read, open, w_hi);
        hp = false;
    }
    rs.hp = hp;
    rs.open = open;
    wave_lds_order();
}

// scalar (uniform) load of one tile descriptor into registers
__device__ __forceinline__ void load_desc(const TileDesc *td, long long k, int &r_lo, int &r_hi, long long &g_lo,
                                          long long &g_hi, long long (&seg_lo)[kMaxSeg], int (&seg_cum)[kMaxSeg + 1])
{
    const TileDesc &d = td[k];
    r_lo = uni(d.r_lo); r_hi = uni(d.r_hi); g_lo = uni(d.g_lo); g_hi = uni(d.g_hi);
    seg_cum[0] = 0;
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) { seg_lo[s] = uni(d.iv_lo[s]); seg_cum[s + 1] = seg_cum[s] + uni(d.n_iv[s]); }
}

template <int WPB, int CAPW, int MINW>
__global__ __launch_bounds__(WPB * 64, MINW) void pileup_wave_kernel(PileupArgs a, const TileDesc *td, long long n_tiles,
                                                                      long long *wave_sums)
{
    using Smem = WaveSmem<CAPW>;
    __shared__ __attribute__((aligned(16))) Smem smem[WPB];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Smem &ws = smem[wid];
    const long long gw = (long long)blockIdx.x * WPB + wid;
    const long long nw = (long long)gridDim.x * WPB;
    WaveAcc acc{0, 0};
    RunState rs{-1, false};

    long long k = gw;
    int r_lo = 0, r_hi = 0;
    long long g_lo = 0, g_hi = 0;
    long long seg_lo[kMaxSeg], tseg_lo[kMaxSeg];
    int seg_cum[kMaxSeg + 1], tseg_n[kMaxSeg];
    IvRegs g;
    bool simple = false;
    if (k < n_tiles) {
        load_desc(td, k, r_lo, r_hi, g_lo, g_hi, seg_lo, seg_cum);
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) { tseg_lo[s] = seg_lo[s]; tseg_n[s] = seg_cum[s + 1] - seg_cum[s]; }
        simple = (r_hi > r_lo) && (g_hi - g_lo <= CAPW) && (g_hi > g_lo);
        if (simple) issue_window_loads(a, lane, r_lo, r_hi - r_lo, seg_lo, seg_cum, g);
    }
    while (k < n_tiles) {
        // next tile: descriptor (scalar loads) and, when it is a plain one-window tile, its first loads
        const long long kn = k + nw;
        int nr_lo = 0, nr_hi = 0;
        long long ng_lo = 0, ng_hi = 0;
        long long nseg_lo[kMaxSeg];
        int nseg_cum[kMaxSeg + 1];
        IvRegs gn;
        bool nsimple = false;
        if (kn < n_tiles) {
            load_desc(td, kn, nr_lo, nr_hi, ng_lo, ng_hi, nseg_lo, nseg_cum);
            nsimple = (nr_hi > nr_lo) && (ng_hi - ng_lo <= CAPW) && (ng_hi > ng_lo);
            if (nsimple) issue_window_loads(a, lane, nr_lo, nr_hi - nr_lo, nseg_lo, nseg_cum, gn);
        }

        if (r_hi > r_lo) {
            // A plain tile is one window over all its reads (loads already in flight).  A tile holding a read
            // longer than the LDS window is split on the fly: sub-batches of whole reads, long reads in chunks.
            int r = r_lo;
            long long chunk_pos = -1, g_first = 0, g_end = 0;
            bool pre = simple;
            for (;;) {
                int r_a, r_b;
                long long w_lo, w_hi;
                bool single, first, last;
                if (chunk_pos < 0) {
                    if (r >= r_hi) break;
                    const long long gl = (r == r_lo) ? g_lo : uni(a.cov_off[r]);
                    int r2;
                    if (g_hi - gl <= CAPW) r2 = r_hi;
                    else {
                        int lo = r, hi = r_hi;
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (uni(a.cov_off[mid]) - gl <= CAPW) lo = mid; else hi = mid;
                        }
                        r2 = lo;
                    }
                    if (r2 > r) {
                        r_a = r; r_b = r2; w_lo = gl; w_hi = (r2 == r_hi) ? g_hi : uni(a.cov_off[r2]);
                        single = false; first = true; last = true;
                        r = r2;
                        if (w_hi == w_lo) continue;
                    } else {
                        g_first = gl; g_end = uni(a.cov_off[r + 1]); chunk_pos = gl;
                    }
                }
                if (chunk_pos >= 0) {
                    r_a = r; r_b = r + 1; w_lo = chunk_pos;
                    w_hi = (chunk_pos + CAPW < g_end) ? chunk_pos + CAPW : g_end;
                    single = true; first = (chunk_pos == g_first); last = (w_hi == g_end);
                    if (last) { chunk_pos = -1; r = r + 1; } else chunk_pos = w_hi;
                }
                if (!pre) {       // narrow the tile's interval ranges to reads [r_a, r_b) and load synchronously
                    int cum = 0;
#pragma unroll
                    for (int s = 0; s < kMaxSeg; ++s) {
                        long long lo = tseg_lo[s], hi = tseg_lo[s] + tseg_n[s];
                        if (s < a.n_seg && !(r_a == r_lo && r_b == r_hi)) {
                            const long long l2 = lower_bound_rid_uni(a.iv_rid, lo, hi, r_a);
                            hi = lower_bound_rid_uni(a.iv_rid, lo, hi, r_b);
                            lo = l2;
                        }
                        seg_lo[s] = lo;
                        seg_cum[s] = cum;
                        cum += (int)(hi - lo);
                    }
                    seg_cum[kMaxSeg] = cum;
                    issue_window_loads(a, lane, r_a, r_b - r_a, seg_lo, seg_cum, g);
                }
                pre = false;
                if (first) { rs.hp = false; rs.open = -1; }
                wave_window<CAPW>(a, ws, acc, rs, lane, r_a, r_b, w_lo, w_hi, single, first, last, seg_lo, seg_cum, g);
            }
        }

        k = kn;
        r_lo = nr_lo; r_hi = nr_hi; g_lo = ng_lo; g_hi = ng_hi; simple = nsimple;
#pragma unroll
        for (int s = 0; s < kMaxSeg; ++s) {
            seg_lo[s] = nseg_lo[s]; seg_cum[s + 1] = nseg_cum[s + 1];
            tseg_lo[s] = nseg_lo[s]; tseg_n[s] = nseg_cum[s + 1] - nseg_cum[s];
        }
        seg_cum[0] = 0;
        g = gn;
    }
    const long long c = wave_reduce_add64(acc.cov), rp = wave_reduce_add64(acc.rep);
    if (lane == 0) { wave_sums[2 * gw] = c; wave_sums[2 * gw + 1] = rp; }
}

// One thread per tile: the 72-byte descriptor the pileup waves read with scalar loads.
__global__ __launch_bounds__(256) void tile_desc_kernel(long long n_tiles, SegStarts sb, const long long *seg_end_dev,
                                                        const int32_t *iv_rid, const int32_t *tile_first,
                                                        const long long *cov_off, TileDesc *td)
{
    const int n_seg = sb.n_seg;
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tiles) return;
    TileDesc d;
    d.r_lo = tile_first[k]; d.r_hi = tile_first[k + 1];
    d.g_lo = cov_off[d.r_lo]; d.g_hi = cov_off[d.r_hi];
#pragma unroll
    for (int s = 0; s < kMaxSeg; ++s) {
        long long lo = 0, hi = 0;
        if (s < n_seg) {
            long long b = sb.start[s], e = sb.start[s + 1];
            if (seg_end_dev) e = *seg_end_dev;
            lo = lower_bound_rid(iv_rid, b, e, d.r_lo);
            hi = lower_bound_rid(iv_rid, b, e, d.r_hi);
        }
        d.iv_lo[s] = lo;
        d.n_iv[s] = (int)(hi - lo);
    }
    td[k] = d;
}

} // namespace raft
