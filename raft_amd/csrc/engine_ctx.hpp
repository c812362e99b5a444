// engine_ctx.hpp -- what the engine's translation units share: the context (one device + its streams and grow-only buffers), the
// device buffers with their placement (virtual ranges over pooled 32 MiB chunks), the control block, error plumbing.
//   engine.hip            context life cycle, one pass (run_pass), finish / fetch / encodings -- and every kernel header
//   engine_pipeline.hip   the host-to-host entry points: chunked upload / pass / download, several contexts, routed streams
//   engine_exchange.hip   pre-split PAF: symmetric flag across ranks, grouped sides, the exchange (RCCL / peer copies), the pre-split job
//   engine_placement.hip  where buffers lie: pool, trim, policy, the callers' input buffers, page-locking
#pragma once
#include "../../include/raft_hip.h"
#include "raft_types.hpp"
#include "wave_launch.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <utility>
#include <vector>

namespace raft {

// The windows of a wave tile (pileup_wave.hpp): its LDS array minus three alignment slots and the sentinel.  A read with more
// windows is piled up in pieces of that many (joined by finalize_count_kernel).
constexpr int kTileCap = kWaveSlots - 4;
constexpr int kWaveCounters = 32;      // tile hand-out counters of the wave kernel, 256 bytes apart (pileup_wave.hpp next_range)
constexpr int kWinMaxRuns = 2;         // runs the window-record instantiations (pileup_wave.hpp IN = 1) take; more: unpacked to coordinate columns first
// coverage arrays a context's placement trial compares (run_pass): off unless asked for -- RAFT_PLACEMENT_TRIALS=<k>, k >= 2, or
// raft_hip_set_placement_trial
inline int default_trial_candidates()
{
    static const int v = [] { const char *e = getenv("RAFT_PLACEMENT_TRIALS"); return e ? std::max(0, std::min(8, atoi(e))) : 0; }();
    return v;
}

struct Ctrl {                         // device control block, cleared every pass
    int32_t err_flags;
    int32_t pad_slow;
    long long err_index;              // (the first 16 bytes are what the pass's host wait reads back)
    int32_t next_tile;                // (unused since round 6: the wave kernel's hand-out counters are raft_hip_ctx::wave_ctr)
    int32_t slow_next;                // delta4: the counter tile ids are drawn from (PileupArgs::slow_counter)
    int32_t pad_extra[2];
    unsigned long long totals[4];     // coverage, repeat bp, read length
    InspectOut insp;
    long long out_totals[4];          // rep / cut / frag totals land here via the scan
    GuessOut guess;                   // sorted runs as seen from samples
    unsigned long long n_exc;         // windows at or above the limit of the encoding a pass wrote directly (PileupArgs::n_exc)
    int32_t n_deep, pad_deep;         // tiles pileup_wave_kernel listed for pileup_deep_kernel (may exceed the list: kErrDeep)
};

// Device buffers.  The large ones the pass streams through (coverage, repeats, cut points, fragments, the pipeline's staging
// columns: `big`) are virtual ranges over 32 MiB physical chunks (hipMemCreate / hipMemMap) that are SPREAD over a wide
// physical span: for a buffer of 1 GiB or more eight times the chunks are created and every eighth is mapped, in a
// shuffled order.
// Why (tools/membench, profiles/r04_membench_placement.txt; tools/placement_probe2.py): what a stream gets from this part
// is a property of where its buffer lies.  Plain stores into an 8 GB hipMalloc block: 5.65 TB/s, with few exceptions; into
// chunks taken one after the other: 5.6 .. 7.1 TB/s from buffer to buffer; into every eighth chunk of a 64 GB span:
// 7.0 .. 7.1 TB/s, every time.  The pileup kernel followed its coverage array -- contexts of ONE process ran at 2.14 or at
// 2.6 ms, and swapping their `cov` buffers swapped their times.
// Two rules the mapping calls turned out to need on this stack (tools/gpu_tmp.py's sequence: one-byte coverage, then two-byte
// coverage in the same context -- tests/test_gpu_windows.py):
//   * a virtual range is reserved once and never given back (hipMemAddressFree) while the process lives.  A range that was
//     unmapped, freed and handed out again by the next hipMemAddressReserve was served from STALE translations: writes and reads
//     of the new buffer went to the chunks the old buffer had been mapped to, deterministically from the second chunk on.
//     (Address space is not scarce: 47 bits.  A range never mapped -- a failed attempt -- may go back.)
//   * chunks are not handed back to the driver either: the spare ones, and the ones of a buffer that is released or outgrown,
//     go to a per-device pool that later buffers draw from (random picks: spread again) -- no create / release storm when a
//     buffer grows, and nothing depends on when the driver wipes released memory.  Cost: the pool keeps up to seven times the
//     largest spread buffer (56 GB for the bench set's coverage array, of 288).
// Any failure falls back (fewer spare chunks, then hipMalloc); RAFT_NO_VMM=1 switches the mapping off.  Buffers other devices
// write into (the exchange's receive side) stay with hipMalloc.
struct ChunkPool {                    // per device; handles of 32 MiB physical chunks nobody maps at the moment
    std::mutex mu;
    std::vector<hipMemGenericAllocationHandle_t> free_chunks;
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    int live_ctx = 0;                 // contexts of this device: the last one to go hands the pool back to the driver
    static ChunkPool &of(int dev) { static ChunkPool pools[64]; return pools[dev & 63]; }
    unsigned long long next() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return rng >> 33; }
    // most chunks the pool keeps (RAFT_VMM_POOL_GB, default 64): what comes back beyond that goes to the driver
    static size_t cap_chunks()
    {
        static const size_t v = [] {
            const char *e = getenv("RAFT_VMM_POOL_GB");
            const double gb = e ? std::max(0.0, atof(e)) : 64.0;
            return (size_t)(gb * 32.0);                    // 32 chunks of 32 MiB per GiB
        }();
        return v;
    }
    // (mu held) a chunk nobody maps: kept for later buffers while there is room, else released
    void put(hipMemGenericAllocationHandle_t h)
    {
        if (free_chunks.size() < cap_chunks()) free_chunks.push_back(h);
        else (void)hipMemRelease(h);
    }
    // hands all but `keep` chunks back to the driver; returns how many went
    size_t trim(size_t keep)
    {
        std::lock_guard<std::mutex> lk(mu);
        size_t n = 0;
        while (free_chunks.size() > keep) { (void)hipMemRelease(free_chunks.back()); free_chunks.pop_back(); ++n; }
        if (free_chunks.empty()) free_chunks.shrink_to_fit();
        return n;
    }
};

// The streams whose work may still use a buffer this thread is about to release or re-map (the context's own and its side
// stream): release() waits for those instead of the whole device -- other contexts' passes go on.  None named: the device.
struct SyncScope {
    static inline thread_local hipStream_t streams[2] = {nullptr, nullptr};
    static inline thread_local int n = 0;
    int saved_n; hipStream_t saved[2];
    SyncScope(hipStream_t a, hipStream_t b) { saved_n = n; saved[0] = streams[0]; saved[1] = streams[1]; streams[0] = a; streams[1] = b; n = 2; }
    ~SyncScope() { n = saved_n; streams[0] = saved[0]; streams[1] = saved[1]; }
    static void wait()
    {
        if (n == 0) { (void)hipDeviceSynchronize(); return; }
        for (int i = 0; i < n; ++i) (void)hipStreamSynchronize(streams[i]);
    }
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool big = false;                 // may be backed by pooled chunks
    int dev = 0;                      // device of the chunks
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    std::vector<size_t> map_order;    // chunk mapped at the i-th 32 MiB of the range
    size_t va_bytes = 0;
    static constexpr size_t kChunk = 32u << 20, kVmmMin = 64u << 20, kSpreadMin = size_t(1) << 30;
    // the placement policy of buffers made from now on (process-wide): 0 = plain hipMalloc, k >= 1 = chunks, k times as many made
    // as used for buffers of 1 GiB or more.  RAFT_NO_VMM=1 / RAFT_VMM_SPREAD=<k> set the start value; raft_hip_set_placement changes it.
    // set once the policy was chosen by hand (RAFT_NO_VMM / RAFT_VMM_SPREAD / raft_hip_set_placement): no placement trial then
    static std::atomic<bool> &policy_explicit()
    {
        static std::atomic<bool> e{getenv("RAFT_NO_VMM") != nullptr || getenv("RAFT_VMM_SPREAD") != nullptr};
        return e;
    }
    static std::atomic<int> &policy()
    {
        static std::atomic<int> p{getenv("RAFT_NO_VMM") ? 0 : (getenv("RAFT_VMM_SPREAD") ? std::max(1, atoi(getenv("RAFT_VMM_SPREAD"))) : 8)};
        return p;
    }
    bool map_chunks(size_t want)
    {
        const int pol = policy().load();
        if (pol <= 0) return false;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0 || kChunk % gran) {
            (void)hipGetLastError(); return false;
        }
        const size_t n = (want + kChunk - 1) / kChunk;
        const size_t kSpread = (size_t)pol;
        hipDeviceptr_t va = nullptr;
        if (hipMemAddressReserve(&va, n * kChunk, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        ChunkPool &pool = ChunkPool::of(dev);
        chunks.clear();
        {
            std::lock_guard<std::mutex> lk(pool.mu);
            // (1) from the pool: random picks -- its chunks lie all over the spans earlier buffers were spread over
            auto draw = [&]() {
                while (chunks.size() < n && !pool.free_chunks.empty()) {
                    const size_t j = (size_t)(pool.next() % pool.free_chunks.size());
                    chunks.push_back(pool.free_chunks[j]);
                    pool.free_chunks[j] = pool.free_chunks.back();
                    pool.free_chunks.pop_back();
                }
            };
            draw();
            // (2) the rest fresh from the driver: k times as many, every k-th for this buffer, the others into the pool; when the
            // device cannot give that many, what was made goes to the pool, serves first, and the rest is tried with fewer spares
            for (size_t k = want >= kSpreadMin ? kSpread : 1; chunks.size() < n; k /= 2) {
                const size_t need = n - chunks.size();
                if (k > 1) {
                    // spares only while the pool has room for them and the device keeps an eighth of its memory (8 GiB at least)
                    // free behind them: plain hipMalloc buffers of this pass, RCCL, torch and other processes live there
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; total_b = 0; }
                    const size_t reserve = std::max<size_t>(size_t(8) << 30, total_b / 8);
                    const size_t room_mem = free_b > reserve + need * kChunk ? (free_b - reserve - need * kChunk) / kChunk : 0;
                    const size_t room_pool = pool.free_chunks.size() < ChunkPool::cap_chunks() ? ChunkPool::cap_chunks() - pool.free_chunks.size() : 0;
                    const size_t spares = std::min(room_mem, room_pool);
                    while (k > 1 && need * (k - 1) > spares) k /= 2;
                }
                std::vector<hipMemGenericAllocationHandle_t> all(need * k);
                size_t made = 0;
                bool ok = true;
                for (; ok && made < need * k; ++made) ok = hipMemCreate(&all[made], kChunk, &prop, 0) == hipSuccess;
                if (!ok) { --made; (void)hipGetLastError(); }
                for (size_t i = 0; i < made; ++i) {
                    if (ok && i % k == 0) chunks.push_back(all[i]);
                    else if (ok) pool.put(all[i]);                 // spares: kept while the pool has room, else handed back
                    else pool.free_chunks.push_back(all[i]);       // (the device ran out: drawn from again right below, whatever the cap)
                }
                if (!ok) { draw(); if (k == 1) break; }
            }
            if (chunks.size() != n) {
                for (auto &h : chunks) pool.put(h);
                chunks.clear();
            }
        }
        if (chunks.size() != n) { (void)hipMemAddressFree(va, n * kChunk); return false; }
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = i;
        unsigned long long x = 0x9E3779B97F4A7C15ull;                     // (Fisher-Yates with a fixed generator)
        for (size_t i = n; i > 1; --i) { x = x * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i - 1], order[(size_t)((x >> 33) % i)]); }
        size_t mapped = 0;
        bool ok = true;
        for (; ok && mapped < n; ++mapped) ok = hipMemMap((hipDeviceptr_t)((char *)va + mapped * kChunk), kChunk, 0, chunks[order[mapped]], 0) == hipSuccess;
        if (!ok) --mapped;
        if (ok) {
            hipMemAccessDesc acc{};
            acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(va, n * kChunk, &acc, 1) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((hipDeviceptr_t)((char *)va + i * kChunk), kChunk);
            { std::lock_guard<std::mutex> lk(pool.mu); for (auto &h : chunks) pool.put(h); }
            chunks.clear();
            return false;                                  // (a range that was mapped, even in part, stays reserved)
        }
        map_order = order;
        p = va; cap = n * kChunk; va_bytes = n * kChunk;
        return true;
    }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        release();
        size_t want = (bytes + 255) & ~size_t(255);
        if (want == 0) want = 256;
        if (big && want >= kVmmMin && map_chunks(want)) return hipSuccess;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            // out of memory with chunks parked in the pool: they go back to the driver and the request is made once more
            (void)hipGetLastError();
            int d = 0;
            if (hipGetDevice(&d) == hipSuccess && ChunkPool::of(d).trim(0) > 0) {
                if (big && want >= kVmmMin && map_chunks(want)) return hipSuccess;
                e = hipMalloc(&p, want);
            }
        }
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release()
    {
        if (p && va_bytes) {
            SyncScope::wait();                             // (nothing in flight may still use the range)
            // chunk by chunk, as it was mapped; a chunk whose mapping did not go away is neither pooled nor released
            ChunkPool &pool = ChunkPool::of(dev);
            std::lock_guard<std::mutex> lk(pool.mu);
            for (size_t i = 0; i < chunks.size(); ++i) {
                if (hipMemUnmap((hipDeviceptr_t)((char *)p + i * kChunk), kChunk) == hipSuccess) pool.put(chunks[i < map_order.size() ? map_order[i] : i]);
                else (void)hipGetLastError();
            }
            // (the range stays reserved: see above)
            chunks.clear(); map_order.clear();
        } else if (p) (void)hipFree(p);
        p = nullptr; cap = 0; va_bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};


} // namespace raft

using namespace raft;

struct raft_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t side_stream = nullptr;   // the general pileup kernel runs beside the fast one
    bool counted = false;                // this context is one of ChunkPool::live_ctx
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_ifork = nullptr;
    raft_hip_params prm{};
    int32_t high_cov = 0, div = 0, minbins = 1;
    int32_t tile_q = 0;               // 0 = variant default
    int32_t force_bucket = 0;
    bool no_bucket_win = false;       // general bucketing: a side's windows did not fit 16 bits once (kErrWide): coordinate pairs from then on
    size_t cov_trial_cap = 0;         // capacity of `cov` the placement trial has been run for (run_pass)
    double trial_ms[2] = {0.0, 0.0};  // that trial: the pileup kernel into `cov` as first placed / into the best of the other candidates (ms)
    int32_t trial_kept = 0;           // 0: the first placement stayed, 1: a plain hipMalloc block was kept, 2: another chunk mapping
    int32_t trial_candidates = default_trial_candidates();   // coverage arrays the placement trial compares; < 2: no trial (the default)
    std::string last_error;

    // device buffers
    DevBuf deep_list;                 // tiles too deep for 16-bit coverage (pileup_deep.hpp)
    long long deep_cap = 1024;        // its entries; grows when a pass lists more (raft_hip_finish)
    DevBuf tail_buf;                  // the fused tail's sums (finalize.hpp FinalizeArgs::tail_part ...)
    DevBuf wave_ctr, ctrl, scan_tmp, cov_off, rep_res_off, tile_first, tile_cuts, block_sums;
    DevBuf cov, rep_cnt, raw_key, raw_s, raw_e, cut_cnt, frag_cnt, rep_off, cut_off, frag_off;
    DevBuf rep_s, rep_e, cuts, frag_read, frag_begin, frag_end;
    DevBuf b_cnt, b_off, b_rid, b_s, b_e;
    DevBuf gs_rid, gs_s, gs_e, gs_off, gs_err;  // raft_hip_group_sides: the slice it hands back (+ its error word)
    std::vector<long long> gs_off_host;
    DevBuf rs_k0, rs_k1, rs_v0, rs_v1, gaps;   // general streams, large inputs: (read id, start | end << 32) per side, before and after the radix sort; long runs of reads without intervals
    DevBuf samples;                   // up to kSamples + 2 read ids at evenly spaced records (guess_runs_kernel): coarse index
    DevBuf in_len, in_col[6];         // staging for raft_hip_run_host
    DevBuf cov8, exc_idx, exc_val, exc_cnt;   // transfer encoding of cov[] (raft_hip_fetch_packed)
    int packed_width = 0;             // width (bytes per window) of the encoding the buffers hold, 0 = none
    long long n_exc = 0, exc_cap = 0;
    int out_width = 4;                // raft_hip_set_output_width: 1 / 2 = the pass writes the encoding, cov[] only on request
    int pass_width = 4;               // what the last pass wrote
    bool cov_valid = false;           // c->cov holds the int32 array of the last pass
    void *pinned = nullptr;           // small pinned scratch for readbacks
    long long *pinned_dev = nullptr;  // the same block as the device addresses it
    hipEvent_t ev_gjoin = nullptr;
    hipEvent_t ev_pass0 = nullptr, ev_pass1 = nullptr, ev_pile0 = nullptr, ev_pile1 = nullptr;

    // chunked host pipeline (raft_hip_run_pipelined): sub-contexts on the same device, one upload stream
    std::vector<raft_hip_ctx *> lanes;
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    std::vector<hipEvent_t> lane_up_ev, lane_down_ev;

    // a pass that verifies in its kernels (see run_pass), and the arguments to run it again if a kernel objects
    bool spec = false;
    bool assume_sym = true;            // what a detecting context's verified pass assumes (the last detection's answer)
    // grp_*: the grouped form (raft_hip_run_device_grouped): per-run record offsets instead of searches; hint_bins >= 0: the
    // caller's window count, which sizes the pass without a host wait
    struct PassArgs {
        int32_t n_reads; const int32_t *len; int64_t n_rec; const int32_t *col[6];
        int32_t n_runs; const long long *rec_off; long long adj[kMaxSeg]; long long hint_bins;
        const uint32_t *win;           // window records instead of col[1..2] (raft_hip_run_device_windows); grouped only
    } args{};
    bool grouped = false;              // the last pass was built on the caller's offsets (verified in its kernels)
    bool no_wait = false;              // ... and sized by the caller's window count: nothing was read back on the way
    DevBuf exp_qid, in_off;            // grouped input without a query column: the ids rebuilt from the offsets; staged offsets
    DevBuf m_off;                      // grouped input of more than kMaxSeg runs: offsets of the merged run
    DevBuf u_s, u_e;                   // window records unpacked for the passes that need coordinate columns
    DevBuf cov_anchor, abs_bits;       // delta4 encoding of cov[] (pack.hpp): block anchors; escape flags of the device-side decoder
    DevBuf exc_idx2, exc_val2, sort_tmp;   // the exception list in ascending order (sort_exceptions)
    DevBuf exc_pidx, exc_pval, exc_tile_n; // delta4: the windows each tile lists, kExcPerTile slots per tile (compact_exceptions_kernel)
    bool exc_sorted = false;
    long long sizes_seq = 0;           // number of the last sizes hand-over of run_pass (publish_sizes_kernel)
    long long pass_seq = 0;            // number of the pass whose closing kernel is queued (written behind the control block when it is through)
    bool seq_armed = false;
    int d4_shift = 0;                  // delta4 on a chunk of a larger array (the host pipelines' lanes): windows of the block its first window lies in that precede it
    DevBuf x_qs, x_qe, x_off, x_raw, x_send_off, x_cnt;   // pre-split exchange (raft_hip_exchange*): what this rank received / staged

    // state of the last pass
    bool ran = false, finished = false;
    int pending_err = RAFT_HIP_OK;
    long long pending_err_index = -1;
    raft_hip_summary sum{};
    long long cap_rep = 0, cap_cut = 0;
    FinalizeArgs fa{};                // of the last pass (the cut points are materialised on demand)
    bool cuts_ready = false;
    bool is_lane = false;              // a sub-context of a host pipeline (prepare_lanes)
    void *h_stage = nullptr;           // page-locked staging of a lane: what the host derives from a chunk's columns (window records, offsets)
    size_t h_stage_cap = 0;
    std::vector<DevBuf *> user_bufs;   // raft_hip_device_alloc
    bool emit_cuts = true;             // the pass writes the cut points (final_stars) itself; false: on demand (raft_hip_set_emit_cuts)
    // what the context's last pass over plain columns found out on the way (run_pass: `speculate`): a pass over a stream of the same
    // shape is built on it without the host wait and verifies it on the device
    struct Shape {
        bool valid = false;
        int32_t n_reads = 0, reso = 0, minbins = 0, interval_length = 0, symmetric_mode = 0, variant = 0, tile_q = 0;
        int64_t n_rec = 0;
        const void *len = nullptr, *qid = nullptr;
        long long B = 0, RU = 0, CU = 0;
        int n_desc = 0;
        long long desc[kMaxSeg] = {};
        bool had_deep = true;          // its tiles of 2^15 intervals or more (raft_hip_finish): none -> a speculative pass does not launch pileup_deep_kernel
        unsigned long long geom_id = 0;   // which writing of the per-read geometry (cov_off, rep_res_off, tile_first, len_seen) that pass ran on
    } shape;
    // The per-read geometry of a pass depends on the read lengths and the parameters alone.  A speculative pass over the same reads
    // keeps what the context holds (geom_id says that nobody has written the arrays since) and only compares the lengths with the copy
    // the scan left (len_seen): one kernel over 8 bytes per read where the scan's two halves ran over the reads twice.
    DevBuf len_seen;
    unsigned long long geom_id = 0;
    bool speculated = false;           // the pass in flight was built on `shape`
    bool deep_skipped = false;         // ... and without a launch of pileup_deep_kernel (a deep tile then refutes it: kErrDeep)
    hipStream_t clean_stream = nullptr;
    bool ctrl_clean = false;           // the control block and the hand-out counters were cleared by the last pass's closing kernel, on clean_stream
};


namespace raft {

inline int fail_hip(raft_hip_ctx *c, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    c->last_error = buf;
    return e == hipErrorOutOfMemory ? RAFT_HIP_ERR_NOMEM : RAFT_HIP_ERR_DEVICE;
}

#define HIP_TRY(c, expr)                                         \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return fail_hip((c), e_, #expr);   \
    } while (0)

inline int check_params(const raft_hip_params *p)
{
    if (!p) return RAFT_HIP_ERR_PARAM;
    if (p->reso <= 0 || p->est_cov <= 0 || p->repeat_length <= 0 || p->interval_length <= 0) return RAFT_HIP_ERR_PARAM;
    if (p->read_length / p->interval_length <= 0) return RAFT_HIP_ERR_PARAM; // div == 0: SIGFPE at chop.hpp:270
    if (p->symmetric_mode < -1 || p->symmetric_mode > 1) return RAFT_HIP_ERR_PARAM;
    return RAFT_HIP_OK;
}

inline void apply_params(raft_hip_ctx *c, const raft_hip_params *p)
{
    c->prm = *p;
    c->high_cov = (int32_t)(p->est_cov * p->cov_mul);            // repeat.hpp:89-90 (int * double, truncated)
    c->div = p->read_length / p->interval_length;                // chop.hpp:248
    c->minbins = (p->repeat_length + p->reso - 1) / p->reso;     // windows a run needs to reach repeat_length
    if (c->minbins < 1) c->minbins = 1;
}

inline int code_from_flags(int flags)
{
    if (flags & (kErrLen | kErrGroup)) return RAFT_HIP_ERR_PARAM;
    if (flags & kErrReadId) return RAFT_HIP_ERR_READ_ID;
    if (flags & kErrCoord) return RAFT_HIP_ERR_COORD;
    if (flags & kErrFragment) return RAFT_HIP_ERR_FRAGMENT;
    if (flags & (kErrInternal | kErrOrder | kErrExtra | kErrHint | kErrDeep)) return RAFT_HIP_ERR_DEVICE;   // (kErrOrder / kErrHint never outlive raft_hip_finish's second run)
    return RAFT_HIP_OK;
}


// ---- what engine.hip provides to the other translation units
void launch_rebase_ids(hipStream_t st, int32_t *ids, long long n, int32_t base);                 // ids[i] -= base (pack.hpp rebase_ids_kernel)
void launch_add_base(hipStream_t st, long long *a, long long n, long long base);                 // a[i] += base (pack.hpp add_base_kernel)

} // namespace raft

// (C linkage only because their definitions sit among the ABI's entry points, inside engine.hip's extern "C" block; none of them is
// exported: raft_amd/csrc/exports.map)
extern "C" {
int run_pass(raft_hip_ctx *c, const raft_hip_ctx::PassArgs &in, bool verify_in_kernels);
raft::Ctrl host_ctrl(const raft_hip_ctx *c);       // the control block as the last pass's closing kernel handed it over
int sort_sides(raft_hip_ctx *c, hipStream_t st, long long n_rec, int32_t n_reads, int symmetric, const int32_t *d_qid, const int32_t *d_qs,
               const int32_t *d_qe, const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te, long long cap_iv, int32_t *o_rid, int32_t *o_s,
               int32_t *o_e, long long *off, int32_t *err_flags, long long *err_index);
int run_grouped(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec, int32_t n_runs, const int64_t *d_rec_offset,
                const long long *adj, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe, int64_t n_bins,
                const uint32_t *d_win = nullptr);
int pack_coverage(raft_hip_ctx *c, int width);
int sort_exceptions(raft_hip_ctx *c);
int fetch_packed_impl(raft_hip_ctx *c, int32_t width, int64_t *cov_offset, void *cov_packed, int32_t *cov_anchor, int64_t exc_cap, int64_t *exc_index,
                      int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                      int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end);
}
