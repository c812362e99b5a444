// engine.hip -- context, launch sequence and C ABI (include/raft_hip.h) of the
// MI355X engine.  One context = one device + one stream + grow-only device
// buffers; one pass = the kernels listed in DESIGN.md §Kernels, in order.
#include "../../include/raft_hip.h"

#include "bucket.hpp"
#include "sort_pairs.hpp"
#include "device_scan.hpp"
#include "finalize.hpp"
#include "pack.hpp"
#include "pileup.hpp"
#include "pileup_wave.hpp"
#include "wave_launch.hpp"
#include "pileup_deep.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

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

using namespace raft;

namespace {

// The windows of a wave tile (pileup_wave.hpp): its LDS array minus three alignment slots and the sentinel.  A read with more
// windows is piled up in pieces of that many (joined by finalize_count_kernel).
constexpr int kTileCap = kWaveSlots - 4;
// coverage arrays a context's placement trial compares (run_pass): off unless asked for -- RAFT_PLACEMENT_TRIALS=<k>, k >= 2, or
// raft_hip_set_placement_trial
int default_trial_candidates()
{
    static const int v = [] { const char *e = getenv("RAFT_PLACEMENT_TRIALS"); return e ? std::max(0, std::min(8, atoi(e))) : 0; }();
    return v;
}
constexpr int kWinMaxRuns = 2;    // runs the window-record instantiations (pileup_wave.hpp IN = 1) take; more: unpacked to coordinate columns first

struct Ctrl {                         // device control block, cleared every pass
    int32_t err_flags;
    int32_t pad_slow;
    long long err_index;              // (the first 16 bytes are what the pass's host wait reads back)
    int32_t next_tile;                // pileup_fast_kernel's tile hand-out counter
    int32_t slow_next;                // the general kernel's item hand-out counter (list mode)
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
    static thread_local hipStream_t streams[2];
    static thread_local int n;
    int saved_n; hipStream_t saved[2];
    SyncScope(hipStream_t a, hipStream_t b) { saved_n = n; saved[0] = streams[0]; saved[1] = streams[1]; streams[0] = a; streams[1] = b; n = 2; }
    ~SyncScope() { n = saved_n; streams[0] = saved[0]; streams[1] = saved[1]; }
    static void wait()
    {
        if (n == 0) { (void)hipDeviceSynchronize(); return; }
        for (int i = 0; i < n; ++i) (void)hipStreamSynchronize(streams[i]);
    }
};
thread_local hipStream_t SyncScope::streams[2] = {nullptr, nullptr};
thread_local int SyncScope::n = 0;

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

struct ReadPrepLoader {               // per read: windows, reserved repeat slots, marker capacity
    const int32_t *len;
    int32_t reso, minbins, L;
    int32_t long_windows, piece_w;    // reads longer than long_windows are piled up in pieces of piece_w windows
    int32_t *err_flags;
    long long *err_index;
    FastDiv by_reso, by_mb1, by_L;    // reso, minbins + 1, L as divisors (three hardware divisions per read, one of them 64 bits wide,
                                      // twice per pass, were most of what the two scan kernels executed)
    __device__ void operator()(long long i, long long (&v)[3]) const
    {
        int l = len[i];
        if (l < 0) {
            atomicOr(err_flags, kErrLen);
            atomicMin((unsigned long long *)err_index, (unsigned long long)i);
            l = 0;
        }
        const int q = fdiv(by_reso, l);
        const long long nb = (long long)q + ((l - q * reso) ? 1 : 0);   // repeat.hpp:32-37
        v[0] = nb;
        // most runs of >= minbins windows a read can hold
        v[1] = nb + 1 < (1LL << 31) && minbins < INT32_MAX ? (long long)fdiv(by_mb1, (int)(nb + 1)) : (nb + 1) / ((long long)minbins + 1);
        if (nb > long_windows) v[1] += 2 * ((nb + piece_w - 1) / piece_w);   // + two runs per piece that touch its edges
        v[2] = fdiv(by_L, l) + 2;                                // chop.hpp:209-223
    }
};

template <int K> struct CountLoader {
    const int32_t *c[K];
    __device__ void operator()(long long i, long long (&v)[K]) const
    {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = c[k][i];
    }
};

// What the host reads back goes straight into its page-locked block (device-visible host memory): a copy command per
// few bytes cost ~25 us each on the device timeline (three of them ahead of the pass's host wait).  Stamped lines (finalize.hpp):
// run_pass looks for them itself instead of sleeping in the runtime's wait (as raft_hip_finish does for the pass's end).
constexpr int kInspWords = (int)(sizeof(InspectOut) / 8), kGuessWords = (int)(sizeof(GuessOut) / 8);
constexpr int kSizesWords = 3 + 2 + kInspWords + kGuessWords;      // scan totals; err_flags | pad, err_index; InspectOut; GuessOut
static_assert(kSizesWords <= 48 && stamped_lines(kSizesWords) * 8 <= 96, "the sizes block outgrew its place");
constexpr int kPackCountWord = 100;   // (raft_hip_pack's count of listed windows: a word of the block outside the stamped lines)
__global__ void publish_sizes_kernel(const long long *scan_totals, const Ctrl *ctrl, long long *host, long long seq)
{
    const long long *c8 = reinterpret_cast<const long long *>(ctrl);
    const long long *in = reinterpret_cast<const long long *>(&ctrl->insp), *gu = reinterpret_cast<const long long *>(&ctrl->guess);
    publish_stamped(host, [&](int i) { return i < 3 ? scan_totals[i] : i < 5 ? c8[i - 3] : i < 5 + kInspWords ? in[i - 5] : gu[i - 5 - kInspWords]; },
                    kSizesWords, seq, (int)threadIdx.x);
    __threadfence_system();
}

constexpr int kWaveCounters = 32;      // tile hand-out counters of the wave kernel, 256 bytes apart (pileup_wave.hpp next_range)
// The pass's last kernel: one wave copies the control block -- everything raft_hip_finish reports -- into the context's page-locked
// block, stamped with the pass's number (raft_hip_finish looks for it itself instead of sleeping in the runtime's wait, whose
// wake-up is 20-30 us of a pass that may take 200), and then clears the block and the hand-out counters for the NEXT pass: the
// one-wave launch that did that at the head of every pass (clear_ctrl_kernel) is only needed for a context's first pass now.
__global__ __launch_bounds__(64) void publish_and_clear_kernel(TailPublish tp, long long *ctrl_words, int32_t *wave_ctr, int word_err_index, int word_insp_err_index)
{
    const int t = (int)threadIdx.x;
    publish_stamped(tp.host_block, [&](int i) { return reinterpret_cast<const volatile long long *>(tp.ctrl_words)[i]; }, tp.n_ctrl_words, tp.pass_seq, t);
    __builtin_amdgcn_s_waitcnt(0x0F70);           // (every lane has its words: nothing below can overtake the reads)
    __threadfence_system();
    if (t < tp.n_ctrl_words) ctrl_words[t] = (t == word_err_index || t == word_insp_err_index) ? -1LL : 0LL;
    if (t < kWaveCounters) wave_ctr[t * kCtrStride] = 0;
}
__global__ void clear_ctrl_kernel(Ctrl *ctrl, int32_t *wave_ctr)
{
    constexpr int kWords = (int)(sizeof(Ctrl) / 8);
    if ((int)threadIdx.x < kWords) reinterpret_cast<long long *>(ctrl)[threadIdx.x] = 0;
    if ((int)threadIdx.x < kWaveCounters) wave_ctr[threadIdx.x * kCtrStride] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { ctrl->err_index = -1; ctrl->insp.err_index = -1; }
}

__global__ void selftest_kernel(const int *in, int *out_dpp, int *out_shfl, unsigned long long *ballots)
{
    const int v = in[threadIdx.x];
    out_dpp[threadIdx.x] = wave_incl_scan_add(v);
    out_shfl[threadIdx.x] = wave_incl_scan_add_shfl(v);
    const unsigned long long b = __ballot(v & 1);
    if ((threadIdx.x & 63) == 0) ballots[threadIdx.x >> 6] = b;
}

} // namespace

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
    DevBuf wave_ctr, ctrl, scan_tmp, cov_off, rep_res_off, cutcap_off, tile_first, tile_cuts, block_sums;
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
    int pass_width = 4;               // what the last pass wrote (4 where the general kernel had to take part)
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
    long long pass_seq = 0;            // number of the pass whose totals_kernel is queued (written behind the control block when it is through)
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
    } shape;
    bool speculated = false;           // the pass in flight was built on `shape`
    hipStream_t clean_stream = nullptr;
    bool ctrl_clean = false;           // the control block and the hand-out counters were cleared by the last pass's closing kernel, on clean_stream
};

namespace {

int fail_hip(raft_hip_ctx *c, hipError_t e, const char *what)
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

int check_params(const raft_hip_params *p)
{
    if (!p) return RAFT_HIP_ERR_PARAM;
    if (p->reso <= 0 || p->est_cov <= 0 || p->repeat_length <= 0 || p->interval_length <= 0) return RAFT_HIP_ERR_PARAM;
    if (p->read_length / p->interval_length <= 0) return RAFT_HIP_ERR_PARAM; // div == 0: SIGFPE at chop.hpp:270
    if (p->symmetric_mode < -1 || p->symmetric_mode > 1) return RAFT_HIP_ERR_PARAM;
    return RAFT_HIP_OK;
}

void apply_params(raft_hip_ctx *c, const raft_hip_params *p)
{
    c->prm = *p;
    c->high_cov = (int32_t)(p->est_cov * p->cov_mul);            // repeat.hpp:89-90 (int * double, truncated)
    c->div = p->read_length / p->interval_length;                // chop.hpp:248
    c->minbins = (p->repeat_length + p->reso - 1) / p->reso;     // windows a run needs to reach repeat_length
    if (c->minbins < 1) c->minbins = 1;
}

int code_from_flags(int flags)
{
    if (flags & (kErrLen | kErrGroup)) return RAFT_HIP_ERR_PARAM;
    if (flags & kErrReadId) return RAFT_HIP_ERR_READ_ID;
    if (flags & kErrCoord) return RAFT_HIP_ERR_COORD;
    if (flags & kErrFragment) return RAFT_HIP_ERR_FRAGMENT;
    if (flags & (kErrInternal | kErrOrder | kErrExtra | kErrHint | kErrDeep)) return RAFT_HIP_ERR_DEVICE;   // (kErrOrder / kErrHint never outlive raft_hip_finish's second run)
    return RAFT_HIP_OK;
}

} // namespace

extern "C" {

int raft_hip_abi_version(void) { return RAFT_HIP_ABI_VERSION; }

const char *raft_hip_strerror(int code)
{
    switch (code) {
    case RAFT_HIP_OK: return "ok";
    case RAFT_HIP_ERR_PARAM: return "invalid parameter (reso/est_cov/repeat_length/interval_length <= 0, read_length < interval_length, or negative read length)";
    case RAFT_HIP_ERR_READ_ID: return "PAF record names a read id outside [0, n_reads)";
    case RAFT_HIP_ERR_COORD: return "PAF coordinate negative or beyond the last coverage window of its read";
    case RAFT_HIP_ERR_FRAGMENT: return "fragment would start before base 0 (overlap_length larger than its first cut point)";
    case RAFT_HIP_ERR_NOMEM: return "out of memory";
    case RAFT_HIP_ERR_DEVICE: return "HIP device/runtime error";
    case RAFT_HIP_ERR_STATE: return "call order violated";
    case RAFT_HIP_ERR_TOO_LARGE: return "input too large for 32-bit per-read quantities";
    default: return "unknown error";
    }
}

const char *raft_hip_last_error(const raft_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int raft_hip_create(int device_id, const raft_hip_params *params, raft_hip_ctx **out)
{
    if (!out) return RAFT_HIP_ERR_PARAM;
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return RAFT_HIP_ERR_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return RAFT_HIP_ERR_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RAFT_HIP_ERR_DEVICE; // kernels are built for gfx950 only
    raft_hip_ctx *c = new (std::nothrow) raft_hip_ctx();
    if (!c) return RAFT_HIP_ERR_NOMEM;
    c->device = device_id;
    for (DevBuf *b : {&c->cov, &c->cov8, &c->cuts, &c->frag_read, &c->frag_begin, &c->frag_end, &c->raw_key, &c->raw_s, &c->raw_e, &c->rep_s, &c->rep_e,
                      &c->in_col[0], &c->in_col[1], &c->in_col[2], &c->in_col[3], &c->in_col[4], &c->in_col[5], &c->u_s, &c->u_e, &c->exp_qid,
                      &c->b_rid, &c->b_s, &c->b_e, &c->rs_k0, &c->rs_k1, &c->rs_v0, &c->rs_v1, &c->gs_rid, &c->gs_s, &c->gs_e})
        b->big = true;                                         // (what a pass streams through: see DevBuf)
    apply_params(c, params);
    if (const char *w = getenv("RAFT_COV_WIDTH")) {           // (test sweeps: every context of the process in that width)
        const int v = atoi(w);
        if (v == 1 || v == 2 || v == 4 || v == kCovDelta4) c->out_width = v;
    }
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ifork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&c->pinned_dev), c->pinned, 0) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_gjoin, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev_pass0) != hipSuccess || hipEventCreate(&c->ev_pass1) != hipSuccess ||
        hipEventCreate(&c->ev_pile0) != hipSuccess || hipEventCreate(&c->ev_pile1) != hipSuccess) {
        raft_hip_destroy(c);
        return RAFT_HIP_ERR_DEVICE;
    }
    c->stream = c->own_stream;
    memset(c->pinned, 0, 4096);                            // (the pass numbers raft_hip_finish looks for start at 1)
    { ChunkPool &pool = ChunkPool::of(device_id); std::lock_guard<std::mutex> lk(pool.mu); ++pool.live_ctx; }
    c->counted = true;
    *out = c;
    return RAFT_HIP_OK;
}

void raft_hip_destroy(raft_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (raft_hip_ctx *l : c->lanes) raft_hip_destroy(l);
    if (c->up_stream) (void)hipStreamSynchronize(c->up_stream);
    if (c->down_stream) (void)hipStreamSynchronize(c->down_stream);
    SyncScope scope(c->stream, c->side_stream);            // (the buffers' releases wait for this context's streams, not the device)
    c->lanes.clear();
    for (hipEvent_t e : c->lane_up_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->lane_down_ev) (void)hipEventDestroy(e);
    c->lane_up_ev.clear(); c->lane_down_ev.clear();
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
    DevBuf *all[] = {&c->deep_list, &c->tail_buf, &c->wave_ctr, &c->ctrl, &c->scan_tmp, &c->cov_off, &c->rep_res_off, &c->cutcap_off, &c->tile_first, &c->tile_cuts,
                     &c->block_sums, &c->cov, &c->rep_cnt, &c->raw_key, &c->raw_s, &c->raw_e, &c->cut_cnt, &c->frag_cnt,
                     &c->rep_off, &c->cut_off, &c->frag_off, &c->rep_s, &c->rep_e, &c->cuts, &c->frag_read,
                     &c->frag_begin, &c->frag_end, &c->b_cnt, &c->b_off, &c->b_rid, &c->b_s, &c->b_e, &c->gs_rid, &c->gs_s, &c->gs_e, &c->gs_off, &c->gs_err, &c->rs_k0, &c->rs_k1, &c->rs_v0, &c->rs_v1, &c->gaps, &c->in_len,
                     &c->samples, &c->exp_qid, &c->in_off, &c->m_off, &c->u_s, &c->u_e, &c->cov_anchor, &c->abs_bits, &c->exc_idx2, &c->exc_val2, &c->sort_tmp, &c->exc_pidx, &c->exc_pval, &c->exc_tile_n, &c->x_qs, &c->x_qe, &c->x_off, &c->x_raw, &c->x_send_off, &c->x_cnt, &c->cov8, &c->exc_idx, &c->exc_val, &c->exc_cnt, &c->in_col[0], &c->in_col[1], &c->in_col[2], &c->in_col[3], &c->in_col[4], &c->in_col[5]};
    for (DevBuf *b : all) b->release();
    for (DevBuf *b : c->user_bufs) { b->release(); delete b; }
    c->user_bufs.clear();
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->ev_pass0) (void)hipEventDestroy(c->ev_pass0);
    if (c->ev_pass1) (void)hipEventDestroy(c->ev_pass1);
    if (c->ev_pile0) (void)hipEventDestroy(c->ev_pile0);
    if (c->ev_pile1) (void)hipEventDestroy(c->ev_pile1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_ifork) (void)hipEventDestroy(c->ev_ifork);
    if (c->ev_gjoin) (void)hipEventDestroy(c->ev_gjoin);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->counted) {                                      // the device's last context hands the pooled chunks back to the driver
        ChunkPool &pool = ChunkPool::of(c->device);
        bool last = false;
        { std::lock_guard<std::mutex> lk(pool.mu); last = --pool.live_ctx == 0; }
        if (last && getenv("RAFT_VMM_KEEP_POOL") == nullptr) (void)pool.trim(0);
    }
    delete c;
}

int64_t raft_hip_trim(int device_id, int64_t keep_bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev || keep_bytes < 0) return -(int64_t)RAFT_HIP_ERR_PARAM;
    // (hipMemRelease needs no current device: the caller's stays as it is)
    return (int64_t)ChunkPool::of(device_id).trim((size_t)(keep_bytes / (int64_t)DevBuf::kChunk)) * (int64_t)DevBuf::kChunk;
}

int32_t raft_hip_set_placement(int32_t spread)
{
    DevBuf::policy_explicit().store(true);
    return (int32_t)DevBuf::policy().exchange(spread < 0 ? 0 : std::min(spread, 64));
}

int raft_hip_placement_trial(raft_hip_ctx *c, double *first_ms, double *best_other_ms, int32_t *kept)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (first_ms) *first_ms = c->trial_ms[0];
    if (best_other_ms) *best_other_ms = c->trial_ms[1];
    if (kept) *kept = c->trial_kept;
    return c->trial_ms[0] > 0.0 ? RAFT_HIP_OK : RAFT_HIP_ERR_STATE;
}

int raft_hip_set_placement_trial(raft_hip_ctx *c, int32_t candidates)
{
    if (!c || candidates < 0 || candidates > 8) return RAFT_HIP_ERR_PARAM;
    c->trial_candidates = candidates;
    return RAFT_HIP_OK;
}

int64_t raft_hip_pool_bytes(int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev || device_id >= 64) return -(int64_t)RAFT_HIP_ERR_PARAM;
    ChunkPool &pool = ChunkPool::of(device_id);
    std::lock_guard<std::mutex> lk(pool.mu);
    return (int64_t)pool.free_chunks.size() * (int64_t)DevBuf::kChunk;
}

int raft_hip_set_params(raft_hip_ctx *c, const raft_hip_params *params)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    int rc = check_params(params);
    if (rc) return rc;
    apply_params(c, params);
    return RAFT_HIP_OK;
}

int raft_hip_set_stream(raft_hip_ctx *c, void *stream)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    c->stream = (hipStream_t)stream;
    return RAFT_HIP_OK;
}

int raft_hip_use_own_stream(raft_hip_ctx *c)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    c->stream = c->own_stream;
    return RAFT_HIP_OK;
}

void *raft_hip_get_stream(raft_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int raft_hip_set_tuning(raft_hip_ctx *c, int32_t tile_bins, int32_t force_bucket_path, int32_t variant)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    // (variant: rounds 1-5 kept several pileup kernels; -1 and 5 name the one there is -- pileup_wave.hpp -- and nothing else is accepted)
    if (variant != -1 && variant != 5) return RAFT_HIP_ERR_PARAM;
    if (tile_bins < 0 || tile_bins > (1 << 20)) return RAFT_HIP_ERR_PARAM;   // (the quantum is a worker's share of windows per draw, not a tile)
    c->tile_q = tile_bins;
    c->force_bucket = force_bucket_path ? 1 : 0;
    return RAFT_HIP_OK;
}

// One pass.  `verify_in_kernels` (the default): no full look at the record
// stream at all (inspect_kernel: ids in range, sorted runs -- one read of the qid column, 0.22-0.24 ms at human scale,
// all of it ahead of the pass's host wait).  A one-workgroup-per-CU kernel samples the stream and names the sorted runs;
// the pass is built on that, and what makes it safe is that tile_desc_kernel and the pileup kernels enforce what they
// rely on: tile ranges tile every run exactly, and every record is checked against the reads of the tile (sub-batch,
// chunk) that processes it.  A record that refutes the guess -- an id out of range, a dip in the order between two
// samples -- raises kErrOrder, and raft_hip_finish() then runs the pass again from the same arguments, this time after
// inspect_kernel has looked at every record (which also reports errors exactly as before).  A detecting context
// (symmetric_mode = -1) assumes the symmetric PAF hifiasm writes and has tile_desc_kernel search for the mirror of
// record 0 where sorted runs keep it (among the records of record 0's target: pileup.hpp MirrorArgs); none found sends
// the pass to the second form too, and the context then stops assuming until a pass of its own detects a symmetric PAF.
// (Measured and dropped: starting on the guess and running inspect_kernel BESIDE the pileup kernels on a low-priority
// stream -- it costs the pileup what it would cost alone, 0.15-0.2 ms; the pass did not get shorter.)
//
// The grouped form (raft_hip_run_device_grouped; `in.rec_off`): the caller says where every read's records begin in every
// run, so nothing is guessed or searched -- the runs are what the offsets say, tile cuts are look-ups -- and, as in a
// verified pass, every record is still checked against the reads of the tile that processes it (with a query column
// at hand; without one the ids ARE the offsets, expanded on the device).  A record that does not sit where the offsets
// say sends the pass to the plain form above.  With the caller's window count (`in.hint_bins`) the host sizes everything
// without waiting for the device: the pass is one uninterrupted sequence of launches.
// The sides of a record stream in any order, sorted by read (bucket.hpp): o_rid / o_s / o_e hold every read's intervals
// together, reads in index order; off[r] says where read r's begin, off[n_reads] how many there are.
static int sort_sides(raft_hip_ctx *c, hipStream_t st, long long n_rec, int32_t n_reads, int symmetric, const int32_t *d_qid, const int32_t *d_qs,
                      const int32_t *d_qe, const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te, long long cap_iv, int32_t *o_rid, int32_t *o_s,
                      int32_t *o_e, long long *off, int32_t *err_flags, long long *err_index)
{
    HIP_TRY(c, c->rs_k0.ensure((size_t)cap_iv * 4)); HIP_TRY(c, c->rs_k1.ensure((size_t)cap_iv * 4));
    HIP_TRY(c, c->rs_v0.ensure((size_t)cap_iv * 8)); HIP_TRY(c, c->rs_v1.ensure((size_t)cap_iv * 8));
    HIP_TRY(c, c->gaps.ensure(sizeof(GapList)));
    HIP_TRY(c, hipMemsetAsync(c->gaps.p, 0, 8, st));
    const unsigned g1 = (unsigned)std::max<long long>(1, std::min<long long>((n_rec + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(expand_sides_kernel, dim3(g1), dim3(256), 0, st, n_rec, n_reads, symmetric, d_qid, d_qs, d_qe, d_tid, d_ts, d_te,
                       c->rs_k0.as<uint32_t>(), c->rs_v0.as<unsigned long long>(), err_flags, err_index);
    int bits = 1;
    while (bits < 32 && (1LL << bits) <= (long long)n_reads) ++bits;               // keys 0 .. n_reads (the sides that do not exist)
    uint32_t *k_sorted = c->rs_k1.as<uint32_t>();
    unsigned long long *v_sorted = c->rs_v1.as<unsigned long long>();
    {   // sort_pairs.hpp: LSD radix sort, eight bits per pass, every store part of a run (hand-written since round 5: no library call on this path)
        HIP_TRY(c, c->sort_tmp.ensure(rs_tmp_bytes<unsigned long long>(cap_iv)));
        bool in_b = false;
        HIP_TRY(c, radix_sort_by_key<unsigned long long>(st, c->rs_k0.as<uint32_t>(), c->rs_v0.as<unsigned long long>(), c->rs_k1.as<uint32_t>(),
                                                         c->rs_v1.as<unsigned long long>(), cap_iv, bits, c->sort_tmp.p, &in_b));
        if (!in_b) { k_sorted = c->rs_k0.as<uint32_t>(); v_sorted = c->rs_v0.as<unsigned long long>(); }
    }
    const unsigned g2 = (unsigned)std::max<long long>(1, std::min<long long>((cap_iv + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(unzip_sorted_kernel, dim3(g2), dim3(256), 0, st, cap_iv, n_reads, k_sorted, v_sorted,
                       o_rid, o_s, o_e, off, c->gaps.as<GapList>());
    hipLaunchKernelGGL(fill_gaps_kernel, dim3(64), dim3(256), 0, st, c->gaps.as<GapList>(), off);
    HIP_TRY(c, hipGetLastError());
    return RAFT_HIP_OK;
}

// ... the same as window records (bucket.hpp, round 5): o_win holds every read's records together, one word each (first window | one
// past the last << 16), off[] where every read's begin -- the pileup kernel's window-record input with one run.  8 bytes per side
// through the sort instead of 12.  A side whose windows need more than 16 bits raises kErrWide (raft_hip_finish runs the pass again
// with the coordinate route).
static int sort_sides_win(raft_hip_ctx *c, hipStream_t st, long long n_rec, int32_t n_reads, int symmetric, const int32_t *d_qid, const int32_t *d_qs,
                          const int32_t *d_qe, const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te, long long cap_iv, uint32_t *o_win,
                          long long *off, int32_t *err_flags, long long *err_index)
{
    HIP_TRY(c, c->rs_v0.ensure((size_t)cap_iv * 8)); HIP_TRY(c, c->rs_v1.ensure((size_t)cap_iv * 8));
    HIP_TRY(c, c->gaps.ensure(sizeof(GapList)));
    HIP_TRY(c, hipMemsetAsync(c->gaps.p, 0, 8, st));
    const unsigned g1 = (unsigned)std::max<long long>(1, std::min<long long>((n_rec + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(expand_sides_win_kernel, dim3(g1), dim3(256), 0, st, n_rec, n_reads, symmetric, c->prm.reso, d_qid, d_qs, d_qe, d_tid, d_ts, d_te,
                       c->rs_v0.as<unsigned long long>(), err_flags, err_index);
    int bits = 1;
    while (bits < 32 && (1LL << bits) <= (long long)n_reads) ++bits;               // keys 0 .. n_reads (the sides that do not exist)
    HIP_TRY(c, c->sort_tmp.ensure(rs_items_tmp_bytes(cap_iv)));
    bool in_b = false;
    HIP_TRY(c, radix_sort_items(st, c->rs_v0.as<unsigned long long>(), c->rs_v1.as<unsigned long long>(), cap_iv, bits, c->sort_tmp.p, &in_b));
    const unsigned long long *sorted = in_b ? c->rs_v1.as<unsigned long long>() : c->rs_v0.as<unsigned long long>();
    const unsigned g2 = (unsigned)std::max<long long>(1, std::min<long long>((cap_iv + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(unzip_items_kernel, dim3(g2), dim3(256), 0, st, cap_iv, n_reads, sorted, o_win, off, c->gaps.as<GapList>());
    hipLaunchKernelGGL(fill_gaps_kernel, dim3(64), dim3(256), 0, st, c->gaps.as<GapList>(), off);
    HIP_TRY(c, hipGetLastError());
    return RAFT_HIP_OK;
}

static int run_pass(raft_hip_ctx *c, const raft_hip_ctx::PassArgs &in, bool verify_in_kernels)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    const int32_t n_reads = in.n_reads;
    const int64_t n_rec = in.n_rec;
    const int32_t *d_len = in.len, *d_qid = in.col[0], *d_qs = in.col[1], *d_qe = in.col[2], *d_tid = in.col[3], *d_ts = in.col[4],
                  *d_te = in.col[5];
    const bool grouped = in.rec_off != nullptr;
    const uint32_t *d_win = in.win;
    if (n_reads < 0 || n_rec < 0) return RAFT_HIP_ERR_PARAM;
    if (n_reads > 0 && !d_len) return RAFT_HIP_ERR_PARAM;
    if (grouped && (in.n_runs < 1 || in.n_runs > kMaxRuns || c->prm.symmetric_mode != 1)) return RAFT_HIP_ERR_PARAM;
    if (d_win && (!grouped || c->prm.reso > 32767)) return RAFT_HIP_ERR_PARAM;   // (65535 windows * reso stays inside int32 where they are unpacked)
    // more runs than the pileup kernels take: merged into one on the device first (bucket.hpp merge_runs_kernel)
    const bool merge = grouped && in.n_runs > kMaxSeg && n_rec > 0;
    const int32_t eff_runs = grouped ? (in.n_runs > kMaxSeg ? 1 : in.n_runs) : 0;
    if (n_rec > 0 && ((!d_qid && !grouped) || ((!d_qs || !d_qe) && !d_win))) return RAFT_HIP_ERR_PARAM;
    if (n_reads == INT32_MAX) return RAFT_HIP_ERR_TOO_LARGE;
    if (n_rec >= (1LL << 29)) return RAFT_HIP_ERR_TOO_LARGE;   // interval byte offsets are 32-bit (2 sides per record at most)
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    SyncScope scope(c->stream, c->side_stream);            // (a buffer that grows waits for this context's streams only)
    bool expand = false;
    // window records go to the fast kernel's own instantiation (pileup_fast.hpp IN = 1) where every tile is the fast kernel's
    // in its default configuration and the runs are few; anything else gets coordinate columns that fall into the same
    // windows (bucket.hpp unpack_windows_kernel) and takes the paths those have
    const bool lean = d_win && n_rec > 0 && !merge && eff_runs <= kWinMaxRuns && !c->force_bucket && getenv("RAFT_NO_WINDOW_KERNEL") == nullptr;
    if (d_win && !lean && n_rec > 0) {
        HIP_TRY(c, c->u_s.ensure((size_t)n_rec * 4));
        HIP_TRY(c, c->u_e.ensure((size_t)n_rec * 4));
        d_qs = c->u_s.as<int32_t>(); d_qe = c->u_e.as<int32_t>();
    }
    if (merge) {
        HIP_TRY(c, c->b_rid.ensure((size_t)n_rec * 4));
        HIP_TRY(c, c->b_s.ensure((size_t)n_rec * 4));
        HIP_TRY(c, c->b_e.ensure((size_t)n_rec * 4));
        HIP_TRY(c, c->m_off.ensure((size_t)(n_reads + 1LL) * 8));
    } else if (n_rec > 0 && grouped && !d_qid && !lean) {      // no query column: the ids are rebuilt from the offsets
        HIP_TRY(c, c->exp_qid.ensure((size_t)n_rec * 4));
        d_qid = c->exp_qid.as<int32_t>();
        expand = true;
    }
    if (n_rec > 0 && (!d_tid || !d_ts || !d_te)) {
        // symmetric_mode = 1: the target columns are never read (query sides only, no detection) and may be omitted
        if (c->prm.symmetric_mode != 1) return RAFT_HIP_ERR_PARAM;
        d_tid = d_qid; d_ts = d_qs; d_te = d_qe;
    }
    c->ran = false; c->finished = false; c->pending_err = RAFT_HIP_OK; c->pending_err_index = -1; c->packed_width = 0; c->seq_armed = false;
    c->cov_valid = false; c->pass_width = 4; c->n_exc = 0; c->exc_sorted = false;
    c->args = in;
    const bool no_verify_env = getenv("RAFT_ALWAYS_INSPECT") != nullptr;   // (A/B measurements; bench.py times both forms)
    // (a detecting context assumes a symmetric PAF -- hifiasm's shape -- until a pass of its own has found otherwise)
    const bool spec = !grouped && verify_in_kernels && !no_verify_env && n_rec > 1 && !c->force_bucket &&
                      (c->prm.symmetric_mode == 1 || (c->prm.symmetric_mode < 0 && c->assume_sym));
    c->spec = spec;
    c->grouped = grouped;
    memset(&c->sum, 0, sizeof c->sum);
    c->sum.n_reads = n_reads; c->sum.n_records = n_rec; c->sum.high_cov = c->high_cov; c->sum.error_index = -1;
    const long long N = n_reads;

    // the pileup kernel writes cov[] in the width the context asked for (int32, or its transfer encodings: pileup_wave.hpp OW)
    const int ow = c->out_width;
    // a grouped pass whose caller announced the window count needs nothing back from the device on the way
    const bool no_wait = grouped && in.hint_bins >= 0 && getenv("RAFT_NO_HINT") == nullptr;
    c->no_wait = no_wait;
    // (RAFT_HOST_CLOCK=1: where the host is, us after entering, when it has issued what -- a speculative pass over an eighth of the
    // bench set is issued in 26 us, 2-3 us a launch: profiles/r06_host_clock.txt)
    static const bool host_clock = getenv("RAFT_HOST_CLOCK") != nullptr;
    const auto hc_t0 = std::chrono::steady_clock::now();
    auto hc_mark = [&](const char *what) { if (host_clock) fprintf(stderr, "[host] %-18s %7.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hc_t0).count()); };
    HIP_TRY(c, hipEventRecord(c->ev_pass0, st));
    hc_mark("ev_pass0");
    HIP_TRY(c, c->ctrl.ensure(sizeof(Ctrl)));
    Ctrl *ctrl = c->ctrl.as<Ctrl>();
    HIP_TRY(c, c->wave_ctr.ensure((size_t)kWaveCounters * kCtrStride * 4));
    // (the last pass's closing kernel has cleared the block behind its hand-over -- unless this is the context's first pass, the
    // last one did not get that far, or the stream is another)
    if (!c->ctrl_clean || c->clean_stream != st || getenv("RAFT_ALWAYS_CLEAR") != nullptr)
        hipLaunchKernelGGL(clear_ctrl_kernel, dim3(1), dim3(64), 0, st, ctrl, c->wave_ctr.as<int32_t>());      // (three fill commands before: ~5 us each on the device)
    c->ctrl_clean = false;
    if (d_win && !lean && n_rec > 0)
        hipLaunchKernelGGL(unpack_windows_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((n_rec + 255) / 256, 256 * 16))), dim3(256), 0, st,
                           (long long)n_rec, d_win, c->prm.reso, c->u_s.as<int32_t>(), c->u_e.as<int32_t>());
    const long long *eff_off = in.rec_off;
    if (merge) {
        const long long stride = (long long)n_reads + 1;
        hipLaunchKernelGGL(check_offsets_kernel, dim3((unsigned)((n_reads + 1LL + 255) / 256)), dim3(256), 0, st, n_reads, in.n_runs, in.rec_off, stride,
                           (long long)n_rec, &ctrl->err_flags, &ctrl->err_index);
        hipLaunchKernelGGL(merge_runs_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(((n_reads + 64LL) / 64 + 3) / 4, 256 * 16))), dim3(256), 0, st,
                           n_reads, in.n_runs, in.rec_off, stride, d_qs, d_qe, c->m_off.as<long long>(), c->b_rid.as<int32_t>(), c->b_s.as<int32_t>(),
                           c->b_e.as<int32_t>(), &ctrl->err_flags);
        d_qid = d_tid = c->b_rid.as<int32_t>(); d_qs = d_ts = c->b_s.as<int32_t>(); d_qe = d_te = c->b_e.as<int32_t>();
        eff_off = c->m_off.as<long long>();
    }

    // ---- two things have to be known before the host can size and launch the rest, and they run side by side:
    //  (main stream) what the record stream looks like -- sorted runs sampled by guess_runs_kernel and, unless the pass
    //      verifies in its kernels, every record by inspect_kernel: ids in range? the runs as sampled? mirror of record 0?
    //  (side stream) the per-read geometry: windows, reserved repeat slots, marker capacity (one scan, three sums).
    // (a grouped pass has nothing to find out about the stream: the scan runs on the main stream, nothing beside it)
    long long h[kSizesWords] = {};                    // the sizes hand-over, taken out of its stamped lines (below)
    InspectOut *hi = reinterpret_cast<InspectOut *>(h + 5);
    GuessOut *hg = reinterpret_cast<GuessOut *>(h + 5 + kInspWords);
    const unsigned igrid = (unsigned)std::max<long long>(1, std::min<long long>((n_rec / 4 + 255) / 256, 256 * 8));
    const int nb_scan = std::max(scan_blocks(N), 1);
    HIP_TRY(c, c->scan_tmp.ensure(((size_t)nb_scan * 3 + 8) * sizeof(long long)));
    HIP_TRY(c, c->cov_off.ensure((size_t)(N + 1) * 8));
    HIP_TRY(c, c->rep_res_off.ensure((size_t)(N + 1) * 8));
    HIP_TRY(c, c->cutcap_off.ensure((size_t)(N + 1) * 8));
    const bool want_guess = !grouped && n_rec > 1 && c->prm.symmetric_mode != 0 && !c->force_bucket;   // (the sorted-segment path is possible)
    if (want_guess) HIP_TRY(c, c->samples.ensure((size_t)(kSamples + 2) * 4));
    GroupedOff grp{};
    if (grouped) {
        grp.off = eff_off; grp.stride = N + 1;
        for (int s2 = 0; s2 < kMaxSeg; ++s2) grp.adj[s2] = merge ? 0 : in.adj[s2];
    }
    long long *scan_totals = nullptr;
    ReadPrepLoader prep_ld{d_len, c->prm.reso, c->minbins, c->prm.interval_length, kTileCap, kTileCap,
                           &ctrl->err_flags, &ctrl->err_index, make_fast_div(c->prm.reso),
                           make_fast_div(c->minbins < INT32_MAX ? c->minbins + 1 : 1), make_fast_div(c->prm.interval_length)};
    ScanOut<3> prep_so{{c->cov_off.as<long long>(), c->rep_res_off.as<long long>(), c->cutcap_off.as<long long>()}};
    // ---- A pass whose sizes the host knows before anything has run needs no wait on the way, and its head is THREE launches
    // (round 6): [geometry scan, first half | run guess] -> [geometry scan, second half + the per-read work of tile_first_kernel +
    // the check of what was assumed] -> tile_desc_kernel.  Two ways to know:
    //  * the caller of a grouped pass announced its window count (no_wait, since round 3);
    //  * SPECULATION: the context's last pass over plain columns went the sorted-run way, and this one has the same shape -- reads,
    //    records, column addresses, parameters.  It is built on what that pass found (windows, reserved slots, where the runs end)
    //    and every kernel that relies on it checks it: the scan's totals against the assumed ones, the sampled run ends against
    //    the assumed ones (kErrHint: the later kernels return at once and raft_hip_finish runs the pass again the long way).
    //    A streaming caller that hands over batch after batch through the same buffers gets the long way once.
    const bool shape_fits = c->shape.valid && c->shape.n_reads == n_reads && c->shape.n_rec == n_rec && c->shape.len == (const void *)d_len &&
                            c->shape.qid == (const void *)d_qid && c->shape.reso == c->prm.reso && c->shape.minbins == c->minbins &&
                            c->shape.interval_length == c->prm.interval_length && c->shape.symmetric_mode == c->prm.symmetric_mode &&
                            c->shape.tile_q == c->tile_q;
    const bool speculate = spec && shape_fits && N > 0 && !c->is_lane && getenv("RAFT_NO_SPECULATE") == nullptr;
    const bool known = N > 0 && (speculate || (no_wait && getenv("RAFT_NO_FUSED_HEAD") == nullptr));
    c->speculated = speculate;
    if (speculate) c->sum.flags |= RAFT_HIP_SUM_SPECULATED;
    if (!known) {
        hipStream_t gst = grouped ? st : c->side_stream;
        if (!grouped) {
            HIP_TRY(c, hipEventRecord(c->ev_ifork, st));                    // (the control block is clear)
            HIP_TRY(c, hipStreamWaitEvent(gst, c->ev_ifork, 0));
        }
        exclusive_scan<ReadPrepLoader, 3>(gst, prep_ld, N, c->scan_tmp.as<long long>(), prep_so, &scan_totals);
        if (!grouped) HIP_TRY(c, hipEventRecord(c->ev_gjoin, gst));
        if (n_rec > 0 && !grouped) {
            if (want_guess)
                hipLaunchKernelGGL(guess_runs_kernel, dim3(kGuessBlocks), dim3(256), 0, st, (long long)n_rec, d_qid, &ctrl->guess,
                                   c->samples.as<int32_t>());
            if (!spec)
                hipLaunchKernelGGL(inspect_kernel, dim3(igrid), dim3(256), 0, st, (long long)n_rec, n_reads,
                                   c->prm.symmetric_mode < 0 ? 1 : 0, d_qid, d_qs, d_qe, d_tid, d_ts, d_te, &ctrl->insp);
        }
        if (!grouped) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_gjoin, 0));
    }
    long long B, RU, CU;
    if (speculate) {
        B = c->shape.B; RU = c->shape.RU; CU = c->shape.CU;
        hg->n_desc = c->shape.n_desc;
        for (int i = 0; i < kMaxSeg; ++i) hg->desc_pos[i] = c->shape.desc[i];
    } else if (no_wait) {
        // (the caller's count is compared with the scan's on the device: kErrHint stops the pass there, and raft_hip_finish runs
        // it again with the host wait)
        // sizes from the caller's window count: B as announced (checked on the device, kErrHint); bounds for the rest --
        // reserved raw-repeat slots sum_r ((w_r + 1) / (minbins + 1) + two per piece of a long read), markers sum_r (len_r / L + 2)
        B = in.hint_bins;
        RU = (B + N) / ((long long)c->minbins + 1) + 4 * (B / kTileCap) + 4;
        CU = B / std::max(1, c->prm.interval_length / c->prm.reso) + 2 * N + 2;
        if (c->prm.interval_length < c->prm.reso) CU = B * ((long long)c->prm.reso / c->prm.interval_length + 1) + 2 * N + 2;
    } else {
        hipLaunchKernelGGL(publish_sizes_kernel, dim3(1), dim3(64), 0, st, scan_totals, ctrl, c->pinned_dev, ++c->sizes_seq);
        // the pass's only host wait: sizes + path choice
        bool seen = false;
        const volatile long long *lines = reinterpret_cast<const volatile long long *>(c->pinned);
        if (!c->is_lane && getenv("RAFT_NO_SPIN") == nullptr) {
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
            for (int it = 0; !(seen = stamped_seen(lines, kSizesWords, c->sizes_seq)); ++it)
                if ((it & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(c, hipStreamSynchronize(st));
        unstamp(lines, kSizesWords, h);
        B = h[0]; RU = h[1]; CU = h[2];
        const int32_t flags = reinterpret_cast<int32_t *>(h + 3)[0];
        if (flags) {
            c->pending_err = code_from_flags(flags);
            c->pending_err_index = h[4];
            c->ran = true;
            HIP_TRY(c, hipEventRecord(c->ev_pile0, st)); HIP_TRY(c, hipEventRecord(c->ev_pile1, st));
            HIP_TRY(c, hipEventRecord(c->ev_pass1, st));
            return RAFT_HIP_OK;
        }
    }
    c->sum.n_bins = B; c->sum.total_windows = B;
    c->cap_rep = RU; c->cap_cut = CU;
    if (RU >= (1LL << 31)) return RAFT_HIP_ERR_TOO_LARGE;   // reserved raw-repeat slots are indexed with 32 bits in LDS
    // The quantum: boundaries at which a worker of the pileup kernel may begin (it cuts its tiles itself; tile_desc_kernel finds each
    // boundary's first read, records and window).  Three tiles' worth (four until round 5: the kernel likes short ranges -- 2.42 / 2.45 /
    // 2.49 / 2.56 ms at two / three / four / eight tiles' worth in one context -- and tile_desc_kernel long ones; the pass is shortest
    // at three, profiles/r05_quantum_sweep.txt), but never so few ranges that workers stay without one (a 50 k-read set) or that the
    // last draws of the kernel are a fifth of its duration (an eighth of the human-scale set: two tiles' worth)
    const long long q3 = 3LL * (kTileCap / 128) * 128, q1 = (kTileCap / 128) * 128;
    const int Q = c->tile_q ? std::max(256, c->tile_q) : (int)std::max(q1, std::min(q3, (B / (8LL * wave_grid_waves(true))) / 128 * 128));
    const long long n_tiles = B / Q + 1;

    if (ow == 4) HIP_TRY(c, c->cov.ensure((size_t)std::max(B, 1LL) * 4));
    else {
        if (ow == kCovDelta4) {
            HIP_TRY(c, c->cov8.ensure((size_t)std::max(B, 1LL) / 2 + 16));
            HIP_TRY(c, c->cov_anchor.ensure(((size_t)std::max(B, 1LL) / kD4Block + 3) * 4));
        } else
        HIP_TRY(c, c->cov8.ensure((size_t)std::max(B, 1LL) * (size_t)ow + 16));
        const long long cap = std::max<long long>(c->exc_cap, std::max<long long>(4096, B / 64));
        HIP_TRY(c, c->exc_idx.ensure((size_t)cap * 8));
        HIP_TRY(c, c->exc_val.ensure((size_t)cap * 4));
        c->exc_cap = cap;
    }
    HIP_TRY(c, c->tile_first.ensure((size_t)(n_tiles + 1) * 4));
    // delta4: a tile lists windows in slots of its own, named by a tile id; the number bounds the ids that have slots (the tiles are
    // cut by the workers -- a tile is closed by a full array, 63 reads, a long read or the end of a range; ids are drawn 32 at a time;
    // a tile beyond them lists in the shared list)
    const bool ow_is_d4 = ow == kCovDelta4;
    long long extra_cap = ow_is_d4 ? 2 * n_tiles + 2 * (B / kTileCap) + N / 32 + 32LL * wave_grid_waves(true) + 1024 : 0;
    if (const char *ec = getenv("RAFT_EXTRA_CAP")) extra_cap = std::max(0, atoi(ec));   // (tests: tiles without slots of their own)
    if ((n_tiles + 1) * 8 >= (1LL << 31)) return RAFT_HIP_ERR_TOO_LARGE;   // boundary words are indexed with 32 bits
    HIP_TRY(c, c->tile_cuts.ensure((size_t)(n_tiles + 1) * sizeof(TileCut)));
    HIP_TRY(c, c->block_sums.ensure((size_t)256 * 8 * 16 * 2 * 4));
    HIP_TRY(c, c->rep_cnt.ensure((size_t)std::max(N, 1LL) * 4));
    HIP_TRY(c, c->cut_cnt.ensure((size_t)std::max(N, 1LL) * 4));
    HIP_TRY(c, c->frag_cnt.ensure((size_t)std::max(N, 1LL) * 4));
    HIP_TRY(c, c->raw_key.ensure((size_t)std::max(RU, 1LL) * 4));
    HIP_TRY(c, c->raw_s.ensure((size_t)std::max(RU, 1LL) * 4));
    HIP_TRY(c, c->raw_e.ensure((size_t)std::max(RU, 1LL) * 4));
    HIP_TRY(c, c->rep_s.ensure((size_t)std::max(RU, 1LL) * 4));
    HIP_TRY(c, c->rep_e.ensure((size_t)std::max(RU, 1LL) * 4));
    HIP_TRY(c, c->cuts.ensure((size_t)std::max(CU, 1LL) * 4));
    HIP_TRY(c, c->frag_read.ensure((size_t)std::max(CU, 1LL) * 4));
    HIP_TRY(c, c->frag_begin.ensure((size_t)std::max(CU, 1LL) * 4));
    HIP_TRY(c, c->frag_end.ensure((size_t)std::max(CU, 1LL) * 4));
    HIP_TRY(c, c->rep_off.ensure((size_t)(N + 1) * 8));
    HIP_TRY(c, c->cut_off.ensure((size_t)(N + 1) * 8));
    HIP_TRY(c, c->frag_off.ensure((size_t)(N + 1) * 8));
    // the tail's sums: per workgroup of 256 reads four words from the count kernel, three of their prefix
    const int tail_blocks = (int)std::max<long long>(1, (N + 255) / 256);
    HIP_TRY(c, c->tail_buf.ensure((size_t)7 * tail_blocks * 8));
    long long *const tail_part = c->tail_buf.as<long long>(), *const tail_prefix = tail_part + (size_t)4 * tail_blocks;

    hc_mark("sized");
    if (known) {
        // the head in two launches (see above): the scan's first half with the run guess beside it, its second half with the per-read
        // work of tile_first_kernel riding on it
        const bool guess_too = !grouped && n_rec > 0 && want_guess;
        long long *partials = c->scan_tmp.as<long long>();
        scan_totals = partials + (long long)nb_scan * 3;
        GuessBeside gb{(long long)n_rec, d_qid, &ctrl->guess, c->samples.as<int32_t>()};
        hipLaunchKernelGGL((scan_partials_kernel<ReadPrepLoader, 3, GuessBeside>), dim3((unsigned)(nb_scan + (guess_too ? kGuessBlocks : 0))), dim3(kScanThreads), 0, st,
                           prep_ld, N, partials, nb_scan, gb);
        PrepPost pp{n_reads, Q, n_tiles, c->tile_first.as<int32_t>(), c->rep_cnt.as<int32_t>(), &ctrl->err_flags, &ctrl->err_index, grp, eff_runs,
                    (long long)n_rec, B, RU, CU};
        hipLaunchKernelGGL((scan_apply_kernel<ReadPrepLoader, 3, true, PrepPost>), dim3((unsigned)nb_scan), dim3(kScanThreads), 0, st, prep_ld, N, partials, scan_totals,
                           prep_so, pp);
    } else
    hipLaunchKernelGGL(tile_first_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, st, n_reads,
                       c->cov_off.as<long long>(), Q, n_tiles, c->tile_first.as<int32_t>(), &ctrl->err_flags, &ctrl->err_index, grp,
                       eff_runs, (long long)n_rec, c->rep_cnt.as<int32_t>(), scan_totals, no_wait ? in.hint_bins : -1LL);
    hc_mark("head launched");
    if (expand)
        hipLaunchKernelGGL(expand_ids_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(((N + 63) / 64 * eff_runs + 3) / 4, 256 * 16))),
                           dim3(256), 0, st, n_reads, eff_runs, grp, c->exp_qid.as<int32_t>(), &ctrl->err_flags);

    int symmetric = c->prm.symmetric_mode == 1 ? 1 : 0;
    int n_desc = 0;
    long long desc[kMaxSeg];
    bool table_ok = false;
    if (grouped) {
        symmetric = 1;
        n_desc = eff_runs - 1;
    } else if (spec) {
        symmetric = 1;
        n_desc = hg->n_desc;
        for (int i = 0; i < std::min(n_desc, kMaxSeg); ++i) desc[i] = hg->desc_pos[i];
        table_ok = n_desc + 1 <= kMaxSeg;            // (the samples index the stream the pass is built on)
        if (c->prm.symmetric_mode < 0 && !table_ok)  // detecting, and not a handful of sorted runs: look at every record after all
            return run_pass(c, in, false);
    } else if (n_rec > 0) {
        if (hi->err_flags) {
            c->pending_err = code_from_flags(hi->err_flags);
            c->pending_err_index = hi->err_index;
            c->ran = true;
            HIP_TRY(c, hipEventRecord(c->ev_pile0, st)); HIP_TRY(c, hipEventRecord(c->ev_pile1, st));
            HIP_TRY(c, hipEventRecord(c->ev_pass1, st));
            return RAFT_HIP_OK;
        }
        if (c->prm.symmetric_mode < 0) { symmetric = hi->sym_found ? 1 : 0; c->assume_sym = symmetric != 0; }
        n_desc = hi->n_desc;
        for (int i = 0; i < std::min(n_desc, kMaxSeg); ++i) desc[i] = hi->desc_pos[i];
        // the samples index the stream when the sampled run ends are exactly the ones the full pass found
        table_ok = want_guess && n_desc + 1 <= kMaxSeg && hg->n_desc == n_desc;
        for (int i = 0; table_ok && i < n_desc; ++i) {
            bool found = false;
            for (int j = 0; j < n_desc; ++j) found = found || hg->desc_pos[j] == desc[i];
            table_ok = found;
        }
    }
    c->sum.symmetric = symmetric;

    PileupArgs pa{};
    pa.read_len = d_len; pa.cov_off = c->cov_off.as<long long>();
    pa.n_tiles = n_tiles; pa.n_reads = n_reads;
    pa.reso = c->prm.reso; pa.high_cov = c->high_cov; pa.repeat_length = c->prm.repeat_length; pa.flank = c->prm.flanking_length;
    pa.cov = ow == 4 ? c->cov.as<int32_t>() : nullptr;
    pa.covp = ow == 4 ? nullptr : c->cov8.p; pa.n_exc = &ctrl->n_exc; pa.exc_cap = c->exc_cap;
    pa.cov_anchor = ow == kCovDelta4 ? c->cov_anchor.as<int32_t>() : nullptr; pa.d4_shift = c->d4_shift;
    const long long d4_tiles = n_tiles + extra_cap;          // (tile ids with slots of their own)
    if (ow == kCovDelta4) {
        HIP_TRY(c, c->exc_pidx.ensure((size_t)d4_tiles * kExcPerTile * 8));
        HIP_TRY(c, c->exc_pval.ensure((size_t)d4_tiles * kExcPerTile * 4));
        HIP_TRY(c, c->exc_tile_n.ensure((size_t)d4_tiles * 4));
        HIP_TRY(c, hipMemsetAsync(c->exc_tile_n.p, 0, (size_t)d4_tiles * 4, st));
        pa.exc_pidx = c->exc_pidx.as<long long>(); pa.exc_pval = c->exc_pval.as<int32_t>(); pa.exc_tile_n = c->exc_tile_n.as<int32_t>();
    }
    pa.exc_idx = c->exc_idx.as<long long>(); pa.exc_val = c->exc_val.as<int32_t>();
    pa.rep_res_off = c->rep_res_off.as<long long>(); pa.rep_cnt = c->rep_cnt.as<int32_t>();
    pa.raw_key = c->raw_key.as<int32_t>(); pa.raw_s = c->raw_s.as<int32_t>(); pa.raw_e = c->raw_e.as<int32_t>();
    pa.block_sums = c->block_sums.as<long long>(); pa.err_flags = &ctrl->err_flags; pa.err_index = &ctrl->err_index;
    pa.tile_counter = &ctrl->next_tile; pa.slow_counter = &ctrl->slow_next;
    HIP_TRY(c, c->deep_list.ensure((size_t)c->deep_cap * sizeof(DeepTile)));
    pa.deep_list = c->deep_list.p; pa.n_deep = &ctrl->n_deep; pa.deep_cap = (int32_t)std::min<long long>(c->deep_cap, INT32_MAX);
    pa.deep_min = 32768; pa.deep_rep_total = &ctrl->totals[1];
    if (const char *e = getenv("RAFT_DEEP_MIN")) pa.deep_min = std::max(1, atoi(e));     // (tests: ordinary tiles through pileup_deep_kernel)
    {   // n / reso as mulhi + shift, exact for 0 <= n < 2^31: with L = ceil(log2 reso) and
        // m = floor(2^(31+L) / reso) + 1 (< 2^32), n / reso == (n * m) >> (31 + L) == mulhi(n, m) >> (L - 1)
        const unsigned d = (unsigned)c->prm.reso;
        if (d == 1) { pa.div_magic = 0; pa.div_shift = -1; }
        else {
            int L = 0;
            while ((1ull << L) < d) ++L;
            pa.div_magic = (uint32_t)((1ull << (31 + L)) / d + 1ull);
            pa.div_shift = L - 1;
        }
    }

    const bool fast = n_rec > 0 && symmetric && !c->force_bucket && n_desc + 1 <= kMaxSeg;
    if (!grouped && spec && !speculate) {
        // what this pass found out on the way, for the next one over a stream of the same shape (see `speculate`); a pass that turns
        // out to have been built on a wrong guess takes it back (raft_hip_finish)
        c->shape.valid = fast && table_ok && N > 0;
        c->shape.n_reads = n_reads; c->shape.n_rec = n_rec; c->shape.len = d_len; c->shape.qid = d_qid;
        c->shape.reso = c->prm.reso; c->shape.minbins = c->minbins; c->shape.interval_length = c->prm.interval_length;
        c->shape.symmetric_mode = c->prm.symmetric_mode; c->shape.tile_q = c->tile_q;
        c->shape.B = B; c->shape.RU = RU; c->shape.CU = CU; c->shape.n_desc = n_desc;
        for (int i = 0; i < kMaxSeg; ++i) c->shape.desc[i] = i < n_desc ? desc[i] : 0;
    } else if (!speculate && !grouped) c->shape.valid = false;
    bool bwin = false;                                // the general bucketing hands the pileup kernel window records (below)
    SegStarts sb{};
    const long long *seg_end_dev = nullptr;
    if (n_rec == 0) {
        pa.n_seg = 0;
        c->sum.interval_path = 0; c->sum.n_segments = 0; c->sum.n_intervals = 0;
    } else if (grouped) {
        sb.n_seg = eff_runs;                          // (where the runs begin is in the offsets, on the device)
        pa.iv_rid = d_qid; pa.iv_s = d_qs; pa.iv_e = d_qe; pa.n_seg = sb.n_seg;
        pa.iv_w = lean ? d_win : nullptr; pa.grp = grp;
        c->sum.interval_path = 0; c->sum.n_segments = in.n_runs; c->sum.n_intervals = n_rec;   // (more than kMaxSeg runs: merged into one first)
    } else if (fast) {
        std::sort(desc, desc + n_desc);
        sb.n_seg = n_desc + 1;
        sb.start[0] = 0;
        for (int i = 0; i < n_desc; ++i) sb.start[i + 1] = desc[i];
        sb.start[n_desc + 1] = n_rec;
        pa.iv_rid = d_qid; pa.iv_s = d_qs; pa.iv_e = d_qe; pa.n_seg = sb.n_seg;
        c->sum.interval_path = 0; c->sum.n_segments = sb.n_seg; c->sum.n_intervals = n_rec;
    } else {
        const long long cap_iv = symmetric ? (long long)n_rec : 2 * (long long)n_rec;
        HIP_TRY(c, c->b_cnt.ensure((size_t)std::max(N, 1LL) * 4));
        HIP_TRY(c, c->b_off.ensure((size_t)(N + 1) * 8));
        HIP_TRY(c, c->b_rid.ensure((size_t)cap_iv * 4));
        HIP_TRY(c, c->b_s.ensure((size_t)cap_iv * 4));
        HIP_TRY(c, c->b_e.ensure((size_t)cap_iv * 4));
        // large inputs are sorted, not scattered (bucket.hpp): the counting sort's random 12-byte writes took 87 ms for 2.9e8
        // shuffled records; it stays for small inputs, where its three launches cost less than the sort's
        // (... and for a symmetric stream of a few sorted runs that is sent here all the same -- force_bucket, A/B: its scatter is local)
        const bool parted = cap_iv >= (1LL << 20) && (!symmetric || n_desc + 1 > kMaxSeg) && getenv("RAFT_NO_RADIX_SORT") == nullptr;
        // ... and as window records where the wave kernel runs and a window index fits 16 bits: 8 bytes per side through the sort,
        // the kernel's leanest input behind it
        bwin = parted && c->prm.reso <= 32767 && !c->no_bucket_win && getenv("RAFT_NO_BUCKET_WINDOWS") == nullptr;
        if (bwin) {
            const int prc = sort_sides_win(c, st, (long long)n_rec, n_reads, symmetric, d_qid, d_qs, d_qe, d_tid, d_ts, d_te, cap_iv,
                                           c->b_s.as<uint32_t>(), c->b_off.as<long long>(), &ctrl->err_flags, &ctrl->err_index);
            if (prc != RAFT_HIP_OK) return prc;
        } else if (parted) {
            const int prc = sort_sides(c, st, (long long)n_rec, n_reads, symmetric, d_qid, d_qs, d_qe, d_tid, d_ts, d_te, cap_iv,
                                       c->b_rid.as<int32_t>(), c->b_s.as<int32_t>(), c->b_e.as<int32_t>(), c->b_off.as<long long>(),
                                       &ctrl->err_flags, &ctrl->err_index);
            if (prc != RAFT_HIP_OK) return prc;
        }
        if (!parted) {
        HIP_TRY(c, hipMemsetAsync(c->b_cnt.p, 0, (size_t)std::max(N, 1LL) * 4, st));
        const unsigned grid = (unsigned)std::min<long long>((n_rec + 255) / 256, 8192);
        hipLaunchKernelGGL(bucket_hist_kernel, dim3(grid), dim3(256), 0, st, (long long)n_rec, n_reads, symmetric, d_qid,
                           d_tid, c->b_cnt.as<int32_t>(), &ctrl->err_flags, &ctrl->err_index);
        {
            CountLoader<1> ld{{c->b_cnt.as<int32_t>()}};
            ScanOut<1> so{{c->b_off.as<long long>()}};
            exclusive_scan<CountLoader<1>, 1>(st, ld, N, c->scan_tmp.as<long long>(), so);
        }
        HIP_TRY(c, hipMemsetAsync(c->b_cnt.p, 0, (size_t)std::max(N, 1LL) * 4, st)); // reused as the scatter cursor
        hipLaunchKernelGGL(bucket_scatter_kernel, dim3(grid), dim3(256), 0, st, (long long)n_rec, n_reads, symmetric, d_qid,
                           d_qs, d_qe, d_tid, d_ts, d_te, c->b_off.as<long long>(), c->b_cnt.as<int32_t>(),
                           c->b_rid.as<int32_t>(), c->b_s.as<int32_t>(), c->b_e.as<int32_t>());
        }
        sb.n_seg = 1; sb.start[0] = 0; sb.start[1] = cap_iv;
        seg_end_dev = c->b_off.as<long long>() + N;   // the true interval count lives at b_off[N]
        pa.iv_rid = c->b_rid.as<int32_t>(); pa.iv_s = c->b_s.as<int32_t>(); pa.iv_e = c->b_e.as<int32_t>(); pa.n_seg = 1;
        if (bwin) {                                   // (the kernel takes its records' reads from the offsets: pileup_wave.hpp IN = 1)
            pa.iv_w = c->b_s.as<uint32_t>();
            pa.grp.off = c->b_off.as<long long>(); pa.grp.stride = N + 1;
            for (int s2 = 0; s2 < kMaxSeg; ++s2) pa.grp.adj[s2] = 0;
        }
        c->sum.interval_path = 1; c->sum.n_segments = n_desc + 1; c->sum.n_intervals = -1; // read back in finish
        c->sum.flags = bwin ? RAFT_HIP_SUM_BUCKET_WINDOWS : 0;
    }
    // the detection of a pass that assumes a symmetric PAF: one more boundary search of this kernel (pileup.hpp MirrorArgs)
    MirrorArgs mir{};
    if (spec && c->prm.symmetric_mode < 0 && fast) mir = {d_qs, d_qe, d_tid, d_ts, d_te, &ctrl->insp.sym_found};
    hipLaunchKernelGGL(tile_desc_kernel, dim3((unsigned)((n_tiles + 2 + 255) / 256)), dim3(256), 0, st, n_tiles, sb, seg_end_dev,
                       pa.iv_rid, c->tile_first.as<int32_t>(), c->cov_off.as<long long>(), c->tile_cuts.as<TileCut>(),
                       (fast && table_ok) ? c->samples.as<int32_t>() : nullptr, (long long)n_rec,
                       c->sum.interval_path == 1 ? c->b_off.as<long long>() : nullptr, &ctrl->err_flags, mir, grp,
                       speculate ? &ctrl->guess : nullptr);
    // ---- the dominant kernel: ONE launch of a persistent grid of single-wave workers, each drawing ranges of reads from the
    // boundaries tile_desc_kernel cut (workers without a range leave at once)
    unsigned n_sum_blocks = 1;
    hc_mark("tile_desc launched");
    HIP_TRY(c, hipEventRecord(c->ev_pile0, st));
    hc_mark("ev_pile0");
    bool wave_launched = false;
    {
        int n_waves = (int)std::max<long long>(1, std::min<long long>(wave_grid_waves(lean || bwin), n_tiles));
        pa.tile_batch = 1;
        int n_ctr = 8;
#ifdef RAFT_WAVE_DIAG   // (make DEFS=-DRAFT_WAVE_DIAG: run-time switches for tools/mode_probe.py -- workers, parts of the kernel, counters)
        if (const char *e = getenv("RAFT_WAVE_WAVES")) n_waves = std::max(1, std::min(n_waves, atoi(e)));
        if (const char *e = getenv("RAFT_WAVE_MODE")) pa.tile_batch |= std::min(15, std::max(0, atoi(e))) << 20;
        if (const char *e = getenv("RAFT_WAVE_COUNTERS")) n_ctr = std::min(kWaveCounters, std::max(1, atoi(e)));
#endif
        pa.tile_batch |= (n_ctr - 1) << 24;
        pa.tile_counter = c->wave_ctr.as<int32_t>();
        pa.piece_w = (int32_t)std::min<long long>(extra_cap, INT32_MAX);    // (delta4: tile ids below this have slots of their own)
        launch_wave_variant(ow, lean || bwin, st, pa.n_seg, c->tile_cuts.p, &pa, n_waves);
        n_sum_blocks = (unsigned)n_waves;
        wave_launched = true;
        // ---- where the coverage array lies, decided by measurement: OPT-IN (raft_hip_set_placement_trial / RAFT_PLACEMENT_TRIALS=<k>;
        // round 5 ran it by default, round 6 does not: the driver's own A/B showed 0.2 % between the policies, and a one-shot caller
        // paid 2 K - 1 extra launches and K - 1 coverage-sized allocations for nothing).  What this kernel gets from the part follows
        // the array it stores into, and not by the KIND of memory: two hipMalloc blocks of one process gave 2.24 and 2.63 ms, two chunk
        // mappings 2.49 and 2.67.  A context that asked for a trial draws K - 1 more arrays at the first pass that makes
        // a coverage array of a GiB or more -- plain blocks and chunk mappings in turn, each only while the device keeps its reserve
        // free behind it --, runs the kernel into each of them warm, and keeps the one it was fastest with.
        const int kTrials = c->trial_candidates;
        if (ow == 4 && c->cov.cap >= DevBuf::kSpreadMin && c->cov.va_bytes && c->cov_trial_cap != c->cov.cap && !c->is_lane && kTrials > 1 &&
            !DevBuf::policy_explicit().load()) {
            c->cov_trial_cap = c->cov.cap;
            struct TrialGuard {                       // every way out of this block releases the candidates and the events
                std::vector<DevBuf> cand;
                std::vector<hipEvent_t> ev;
                ~TrialGuard()
                {
                    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
                    for (DevBuf &b : cand) b.release();
                }
            } tg;
            tg.cand.resize((size_t)kTrials - 1);
            tg.ev.assign((size_t)2 * kTrials, nullptr);
            std::vector<DevBuf> &cand = tg.cand;
            std::vector<hipEvent_t> &ev = tg.ev;
            int n_cand = 0;
            for (int k = 0; k + 1 < kTrials; ++k) {
                // a candidate is drawn only while an eighth of the device's memory (8 GiB at least) stays free behind it: the same
                // reserve map_chunks keeps for its spare chunks (torch, RCCL and other processes live there)
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
                if (free_b < c->cov.cap + std::max<size_t>(size_t(8) << 30, total_b / 8)) break;
                cand[(size_t)k].big = (k & 1) != 0;       // plain block, chunk mapping, plain block, ...
                if (cand[(size_t)k].ensure(c->cov.cap) != hipSuccess) { (void)hipGetLastError(); break; }
                ++n_cand;
            }
            bool ok = n_cand > 0;
            for (size_t i = 0; ok && i < ev.size(); ++i) ok = hipEventCreate(&ev[i]) == hipSuccess;
            if (ok) {
                // (the launch above was the context's first -- code going to the device, cold translations: 3 ms, or 200 -- and says
                // nothing; the first run into an array pays for its first touch; the second is the measurement.  What a run leaves
                // behind and the next must not see: the reads' repeat counters, the hand-out counters)
                auto one_run = [&](int32_t *cov_p, hipEvent_t e0, hipEvent_t e1) -> int {
                    PileupArgs x = pa;
                    x.cov = cov_p;
                    HIP_TRY(c, hipMemsetAsync(c->rep_cnt.p, 0, (size_t)std::max(N, 1LL) * 4, st));
                    HIP_TRY(c, hipMemsetAsync(c->wave_ctr.p, 0, (size_t)kWaveCounters * kCtrStride * 4, st));
                    HIP_TRY(c, hipMemsetAsync(&ctrl->n_deep, 0, 4, st));
                    if (e0) HIP_TRY(c, hipEventRecord(e0, st));
                    launch_wave_variant(ow, lean || bwin, st, x.n_seg, c->tile_cuts.p, &x, n_waves);
                    if (e1) HIP_TRY(c, hipEventRecord(e1, st));
                    return RAFT_HIP_OK;
                };
                int trc = RAFT_HIP_OK;
                for (int k = 0; k < n_cand && trc == RAFT_HIP_OK; ++k) trc = one_run(cand[(size_t)k].as<int32_t>(), nullptr, nullptr);
                if (trc == RAFT_HIP_OK) trc = one_run(c->cov.as<int32_t>(), ev[0], ev[1]);
                for (int k = 0; k < n_cand && trc == RAFT_HIP_OK; ++k) trc = one_run(cand[(size_t)k].as<int32_t>(), ev[(size_t)2 * k + 2], ev[(size_t)2 * k + 3]);
                if (trc != RAFT_HIP_OK) { (void)hipStreamSynchronize(st); return trc; }      // (nothing in flight may still use a candidate)
                HIP_TRY(c, hipEventSynchronize(ev[(size_t)2 * n_cand + 1]));
                float best = 0.f;
                HIP_TRY(c, hipEventElapsedTime(&best, ev[0], ev[1]));
                c->trial_ms[0] = best; c->trial_ms[1] = 0.0;
                int keep = -1;
                for (int k = 0; k < n_cand; ++k) {
                    float t = 0.f;
                    HIP_TRY(c, hipEventElapsedTime(&t, ev[(size_t)2 * k + 2], ev[(size_t)2 * k + 3]));
                    if (c->trial_ms[1] == 0.0 || t < c->trial_ms[1]) c->trial_ms[1] = t;
                    if (t < best * 0.985f) { best = t; keep = k; }      // (a candidate has to win by more than the noise of two launches)
                }
                c->trial_kept = keep >= 0 ? (cand[(size_t)keep].big ? 2 : 1) : 0;
                if (keep >= 0) { std::swap(c->cov, cand[(size_t)keep]); c->cov_trial_cap = c->cov.cap; }
                // (whichever array is kept holds this pass's coverage: every run wrote all of it)
            }
        }
    }
    hc_mark("pileup launched");
    HIP_TRY(c, hipEventRecord(c->ev_pile1, st));
    // the tiles the wave kernel listed instead of piling them up (2^15 intervals or more: pileup_deep.hpp); nearly always none
    if (wave_launched)
        hipLaunchKernelGGL(pileup_deep_kernel, dim3(1024), dim3(kDeepThreads), 0, st, pa, c->deep_list.as<DeepTile>(), &ctrl->n_deep, pa.deep_cap, ow);
    hc_mark("ev_pile1");
    if (ow == kCovDelta4)      // the windows the tiles listed, gathered into the shared list (whose counter the control block carries)
        hipLaunchKernelGGL(compact_exceptions_kernel, dim3((unsigned)((d4_tiles + kCompactTiles - 1) / kCompactTiles)), dim3(256), 0, st, d4_tiles, kExcPerTile,
                           c->exc_tile_n.as<int32_t>(), c->exc_pidx.as<long long>(), c->exc_pval.as<int32_t>(), &ctrl->n_exc, c->exc_cap, c->exc_idx.as<long long>(), c->exc_val.as<int32_t>());

    // ---- per-read tail: order repeats, mask markers, fragments
    FinalizeArgs fa{};
    fa.n_reads = n_reads; fa.read_len = d_len; fa.rep_res_off = c->rep_res_off.as<long long>();
    fa.rep_cnt = c->rep_cnt.as<int32_t>(); fa.raw_key = c->raw_key.as<int32_t>(); fa.raw_s = c->raw_s.as<int32_t>();
    fa.raw_e = c->raw_e.as<int32_t>(); fa.interval_length = c->prm.interval_length; fa.div = c->div;
    fa.overlap_length = c->prm.overlap_length; fa.cut_cnt = c->cut_cnt.as<int32_t>(); fa.frag_cnt = c->frag_cnt.as<int32_t>();
    fa.rep_off = c->rep_off.as<long long>(); fa.cut_off = c->cut_off.as<long long>(); fa.frag_off = c->frag_off.as<long long>();
    fa.rep_s = c->rep_s.as<int32_t>(); fa.rep_e = c->rep_e.as<int32_t>(); fa.cuts = c->cuts.as<int32_t>();
    fa.frag_read = c->frag_read.as<int32_t>(); fa.frag_begin = c->frag_begin.as<int32_t>(); fa.frag_end = c->frag_end.as<int32_t>();
    fa.err_flags = &ctrl->err_flags; fa.err_index = &ctrl->err_index;
    fa.by_L = make_fast_div(c->prm.interval_length); fa.by_div = make_fast_div(c->div); fa.by_reso = make_fast_div(c->prm.reso);
    fa.long_windows = kTileCap; fa.reso = c->prm.reso; fa.repeat_length = c->prm.repeat_length;
    fa.flank = c->prm.flanking_length; fa.rep_cnt_rw = c->rep_cnt.as<int32_t>(); fa.total_repeat = &ctrl->totals[1];
    fa.tail_part = tail_part; fa.tail_prefix = tail_prefix; fa.tail_blocks = tail_blocks;
    fa.rep_off_w = c->rep_off.as<long long>(); fa.cut_off_w = c->cut_off.as<long long>(); fa.frag_off_w = c->frag_off.as<long long>();
    {
        // The tail: count -> prefix -> fill -> publish (finalize.hpp FinalizeArgs::tail_part).  The fill kernel makes the three offset
        // arrays on its way; one workgroup in between turns the count kernel's per-workgroup sums into bases and into the totals the
        // host is handed; the last kernel, one wave, hands the control block over -- one block (+1024 bytes) of the context's
        // page-locked memory, stamped with the pass's number.
        const unsigned rgrid = (unsigned)tail_blocks;
        TailPublish tp{};
        tp.n_tiles = (long long)n_sum_blocks; tp.tile_sums = c->block_sums.as<long long>(); tp.totals = ctrl->totals;
        tp.bucket_off = c->sum.interval_path == 1 ? c->b_off.as<long long>() : nullptr; tp.tails = ctrl->out_totals;
        tp.ctrl_words = reinterpret_cast<const long long *>(ctrl); tp.n_ctrl_words = (int)(sizeof(Ctrl) / 8);
        tp.host_block = c->pinned_dev + 128; tp.pass_seq = ++c->pass_seq;
        if (N > 0) hipLaunchKernelGGL(finalize_count_kernel, dim3(rgrid), dim3(256), 0, st, fa);
        hipLaunchKernelGGL(tail_prefix_kernel, dim3((unsigned)((tail_blocks + 1023) / 1024)), dim3(1024), 0, st, fa, tp);
        if (N > 0) {
            if (c->emit_cuts) hipLaunchKernelGGL(finalize_fill_kernel<true>, dim3(rgrid), dim3(256), 0, st, fa);
            else hipLaunchKernelGGL(finalize_fill_kernel<false>, dim3(rgrid), dim3(256), 0, st, fa);
        }
        hipLaunchKernelGGL(publish_and_clear_kernel, dim3(1), dim3(64), 0, st, tp, reinterpret_cast<long long *>(ctrl), c->wave_ctr.as<int32_t>(),
                           (int)(offsetof(Ctrl, err_index) / 8), (int)((offsetof(Ctrl, insp) + offsetof(InspectOut, err_index)) / 8));
        c->seq_armed = true;
        c->ctrl_clean = true; c->clean_stream = st;
    }
    c->fa = fa; c->cuts_ready = c->emit_cuts;
    c->pass_width = ow; c->cov_valid = ow == 4;
    hc_mark("tail launched");
    HIP_TRY(c, hipEventRecord(c->ev_pass1, st));
    HIP_TRY(c, hipGetLastError());
    hc_mark("ev_pass1");
    c->ran = true;
    return RAFT_HIP_OK;
}

int raft_hip_run_device(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec,
                        const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe,
                        const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te)
{
    raft_hip_ctx::PassArgs in{};
    in.n_reads = n_reads; in.len = d_len; in.n_rec = n_rec;
    in.col[0] = d_qid; in.col[1] = d_qs; in.col[2] = d_qe; in.col[3] = d_tid; in.col[4] = d_ts; in.col[5] = d_te;
    in.hint_bins = -1;
    return run_pass(c, in, true);
}

static int run_grouped(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec, int32_t n_runs, const int64_t *d_rec_offset,
                       const long long *adj, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe, int64_t n_bins,
                       const uint32_t *d_win = nullptr)
{
    if (!c || !d_rec_offset) return RAFT_HIP_ERR_PARAM;
    if (c->force_bucket && d_qid && c->prm.symmetric_mode == 1)          // (tests, A/B: the counting-sort path needs no offsets)
        return raft_hip_run_device(c, n_reads, d_len, n_rec, d_qid, d_qs, d_qe, nullptr, nullptr, nullptr);
    raft_hip_ctx::PassArgs in{};
    in.n_reads = n_reads; in.len = d_len; in.n_rec = n_rec;
    in.col[0] = d_qid; in.col[1] = d_qs; in.col[2] = d_qe; in.win = d_win;
    in.n_runs = n_runs; in.rec_off = reinterpret_cast<const long long *>(d_rec_offset);
    for (int s = 0; s < kMaxSeg; ++s) in.adj[s] = adj ? adj[s] : 0;
    in.hint_bins = n_bins >= 0 ? n_bins : -1;
    return run_pass(c, in, true);
}

int raft_hip_run_device_grouped(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec, int32_t n_runs,
                                const int64_t *d_rec_offset, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe,
                                int64_t n_bins)
{
    return run_grouped(c, n_reads, d_len, n_rec, n_runs, d_rec_offset, nullptr, d_qid, d_qs, d_qe, n_bins);
}

int raft_hip_run_device_windows(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec, int32_t n_runs,
                                const int64_t *d_rec_offset, const uint32_t *d_win, int64_t n_bins)
{
    if (n_rec > 0 && !d_win) return RAFT_HIP_ERR_PARAM;
    static const uint32_t none = 0;                       // (no records: the column is never read, but says which form this is)
    return run_grouped(c, n_reads, d_len, n_rec, n_runs, d_rec_offset, nullptr, nullptr, nullptr, nullptr, n_bins, d_win ? d_win : &none);
}

int raft_hip_run_host(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                      const int32_t *qid, const int32_t *qs, const int32_t *qe,
                      const int32_t *tid, const int32_t *ts, const int32_t *te)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (n_reads < 0 || n_rec < 0) return RAFT_HIP_ERR_PARAM;
    if (n_reads > 0 && !read_len) return RAFT_HIP_ERR_PARAM;
    // With symmetric_mode = 1 (the tokeniser saw the mirror of record 0, chop.hpp:175-184) only the query side is piled
    // up: the three target columns are neither needed nor uploaded (half of the H2D bytes) and may be NULL.
    const int n_cols = c->prm.symmetric_mode == 1 ? 3 : 6;
    if (n_rec > 0 && (!qid || !qs || !qe || (n_cols == 6 && (!tid || !ts || !te)))) return RAFT_HIP_ERR_PARAM;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    HIP_TRY(c, c->in_len.ensure((size_t)std::max<long long>(n_reads, 1) * 4));
    if (n_reads) HIP_TRY(c, hipMemcpyAsync(c->in_len.p, read_len, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
    const int32_t *src[6] = {qid, qs, qe, tid, ts, te};
    for (int k = 0; k < n_cols; ++k) {
        HIP_TRY(c, c->in_col[k].ensure((size_t)std::max<long long>(n_rec, 1) * 4));
        if (n_rec) HIP_TRY(c, hipMemcpyAsync(c->in_col[k].p, src[k], (size_t)n_rec * 4, hipMemcpyHostToDevice, st));
    }
    const bool six = n_cols == 6;
    return raft_hip_run_device(c, n_reads, c->in_len.as<int32_t>(), n_rec, c->in_col[0].as<int32_t>(),
                               c->in_col[1].as<int32_t>(), c->in_col[2].as<int32_t>(), six ? c->in_col[3].as<int32_t>() : nullptr,
                               six ? c->in_col[4].as<int32_t>() : nullptr, six ? c->in_col[5].as<int32_t>() : nullptr);
}

// the control block as the pass's last workgroup handed it over (totals_kernel: stamped lines, 1024 bytes into the page-locked block)
static Ctrl host_ctrl(const raft_hip_ctx *c)
{
    static_assert(sizeof(Ctrl) % 8 == 0 && sizeof(Ctrl) / 8 <= 48, "the control block travels in one wave's stamped lines");
    long long w[sizeof(Ctrl) / 8];
    unstamp(reinterpret_cast<const volatile long long *>(c->pinned) + 128, (int)(sizeof(Ctrl) / 8), w);
    Ctrl hc;
    memcpy(&hc, w, sizeof(Ctrl));
    return hc;
}

int raft_hip_finish(raft_hip_ctx *c, raft_hip_summary *summary)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (!c->ran) return RAFT_HIP_ERR_STATE;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->finished) {
        // The pass's last workgroup writes the pass's number behind the control block: seen there, everything is done.  The host
        // looks for it itself for a while (a pass is 0.2-3 ms; the runtime's wait sleeps, and waking up costs 20-30 us) and
        // falls back to the runtime's wait -- which is also what reports a device fault.
        bool seen = false;
        // (bounded by what a pass takes: 4 ms; a pipeline lane does not spin at all -- its thread shares the host's cores with
        // the other lanes, the tokeniser's and the formatter's workers, and its pass is a tenth of its transfers)
        if (c->seq_armed && !c->is_lane && getenv("RAFT_NO_SPIN") == nullptr) {
            const volatile long long *lines = reinterpret_cast<const volatile long long *>(c->pinned) + 128;
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(4);
            for (int it = 0; !(seen = stamped_seen(lines, (int)(sizeof(Ctrl) / 8), c->pass_seq)); ++it)
                if ((it & 255) == 255 && std::chrono::steady_clock::now() > t_end) break;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(c, hipStreamSynchronize(c->stream));
        else {                                           // (the number was seen without the runtime: a fault of this pass still surfaces here)
            const hipError_t q = hipStreamQuery(c->stream);
            if (q != hipSuccess && q != hipErrorNotReady) return fail_hip(c, q, "hipStreamQuery after the pass");
        }
        auto ctrl_block = [&]() { return host_ctrl(c); };
        int n_reruns = 0;
        auto again = [&](const raft_hip_ctx::PassArgs &a) -> int {       // the pass once more, this time nothing assumed
            ++n_reruns;
            const int rc = run_pass(c, a, false);
            if (rc != RAFT_HIP_OK) return rc;
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->spec = false;
            return RAFT_HIP_OK;
        };
        if (c->speculated && c->pending_err == RAFT_HIP_OK) {
            const Ctrl hc = ctrl_block();
            if (hc.err_flags & kErrHint) {
                // the stream is not what the context's last pass saw (other lengths, other run ends): the pass again, nothing remembered
                c->shape.valid = false;
                ++n_reruns;
                const int rc = run_pass(c, c->args, true);
                if (rc != RAFT_HIP_OK) return rc;
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
        }
        if (c->grouped && c->pending_err == RAFT_HIP_OK) {
            const Ctrl hc = ctrl_block();
            if (hc.err_flags & kErrHint) {
                // the caller's window count is not what the read lengths give: the same pass, sized by the device's own count
                auto a = c->args;
                a.hint_bins = -1;
                const int rc = again(a);
                if (rc != RAFT_HIP_OK) return rc;
            }
        }
        if (c->grouped && c->pending_err == RAFT_HIP_OK) {
            const Ctrl hc = ctrl_block();
            if ((hc.err_flags & (kErrOrder | kErrReadId)) && !(hc.err_flags & kErrStop) && c->args.col[0]) {
                // a record does not sit where the caller's offsets say: the offsets are dropped and the query column is
                // taken for what it is (the plain pass, after a look at every record)
                auto a = c->args;
                a.rec_off = nullptr; a.n_runs = 0; a.hint_bins = -1;
                const int rc = again(a);
                if (rc != RAFT_HIP_OK) return rc;
            }
        }
        if (c->spec && c->pending_err == RAFT_HIP_OK) {
            // did a kernel meet a record that refutes the sampled guess the pass was built on?
            const Ctrl hc = ctrl_block();
            c->spec = false;
            const bool no_mirror = c->prm.symmetric_mode < 0 && hc.insp.sym_found == 0;   // assumed symmetric, found no mirror
            if (no_mirror) c->assume_sym = false;
            if ((hc.err_flags & (kErrOrder | kErrReadId)) || no_mirror) {   // run it again, this time after looking at every record
                c->shape.valid = false;
                const int rc = again(c->args);
                if (rc != RAFT_HIP_OK) return rc;
            }
        }
        // Three more reasons to run the pass again, each of which may turn up in the re-run of another:
        //  * kErrWide: general bucketing, a side whose windows do not fit 16 bits -> coordinate pairs from now on;
        //  * kErrDeep: more tiles of 2^15 intervals or more than the list for pileup_deep_kernel held -> once more, with room;
        //  * more windows at or above the encoding's limit than the list held -> once more with room for all of them.
        for (int round = 0; round < 4 && c->pending_err == RAFT_HIP_OK; ++round) {
            const Ctrl hc = ctrl_block();
            if (hc.err_flags & kErrStop) break;
            bool rerun = false;
            if ((hc.err_flags & kErrWide) && !c->no_bucket_win) {
                c->no_bucket_win = true;            // a side's windows do not fit 16 bits: this context buckets coordinate pairs from now on
                rerun = true;
            } else if ((hc.err_flags & kErrDeep) && (long long)hc.n_deep > c->deep_cap) {
                c->deep_cap = (long long)hc.n_deep + 64;     // more deep tiles than the list held: once more, with room
                rerun = true;
            } else if (c->pass_width != 4 && (long long)hc.n_exc > c->exc_cap && !(hc.err_flags & ~(kErrOrder | kErrDeep | kErrWide))) {
                c->exc_cap = (long long)hc.n_exc;
                rerun = true;
            }
            if (!rerun) break;
            const int rc = again(c->args);
            if (rc != RAFT_HIP_OK) return rc;
        }
        if (c->pending_err == RAFT_HIP_OK) {
            Ctrl hc;
            hc = host_ctrl(c);   // copied at the end of the pass
            if (c->pass_width != 4) { c->n_exc = (long long)hc.n_exc; c->packed_width = c->pass_width; c->exc_sorted = false; }
            c->sum.n_repeats = hc.out_totals[0]; c->sum.n_cuts = hc.out_totals[1]; c->sum.n_fragments = hc.out_totals[2];
            if (c->sum.interval_path == 1) c->sum.n_intervals = hc.out_totals[3];
            c->sum.total_coverage = (long long)hc.totals[0];
            c->sum.total_repeat_length = (long long)hc.totals[1];
            c->sum.total_read_length = (long long)hc.totals[2];
            if (hc.n_deep > 0) c->sum.flags |= RAFT_HIP_SUM_DEEP_TILES;
            if (hc.err_flags) {
                c->pending_err = code_from_flags(hc.err_flags);
                c->pending_err_index = hc.err_index;
            }
        }
        if (n_reruns > 0) c->sum.flags |= RAFT_HIP_SUM_RERUN;
        c->sum.error_index = c->pending_err ? c->pending_err_index : -1;
        c->finished = true;
    }
    if (summary) *summary = c->sum;
    return c->pending_err;
}

// The cut points (one int per kept marker, 0.4 GB at human scale) are not written by the pass: the fragments are
// derived while the markers are walked.  The first caller that asks for them pays for one more per-read kernel.
static int materialise_cuts(raft_hip_ctx *c)
{
    if (c->cuts_ready) return RAFT_HIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->sum.n_reads > 0) {
        hipLaunchKernelGGL(finalize_cuts_kernel, dim3((unsigned)((c->sum.n_reads + 255) / 256)), dim3(256), 0, c->stream, c->fa);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->cuts_ready = true;
    return RAFT_HIP_OK;
}

// cov[] as int32 after a pass that wrote its encoding directly: decoded on the device, once, for the caller that asks
static int materialise_cov(raft_hip_ctx *c)
{
    if (c->cov_valid) return RAFT_HIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const long long B = c->sum.n_bins;
    HIP_TRY(c, c->cov.ensure((size_t)std::max(B, 1LL) * 4));
    if (B > 0) {
        const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((B / 4 + 255) / 256, 256 * 16));
        if (c->pass_width == kCovDelta4) {
            if (c->d4_shift != 0) return RAFT_HIP_ERR_STATE;      // (a pipeline lane's chunk: its blocks do not begin at its first window)
            HIP_TRY(c, c->abs_bits.ensure(((size_t)B / 32 + 2) * 4));
            hipLaunchKernelGGL(delta4_expand_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((B / 32 + 255) / 256, 256 * 16))), dim3(256), 0, c->stream,
                               c->cov8.as<uint8_t>(), B, c->cov.as<int32_t>(), c->abs_bits.as<unsigned>());
            if (c->n_exc > 0)
                hipLaunchKernelGGL(scatter_exceptions_kernel, dim3((unsigned)std::min<long long>((c->n_exc + 255) / 256, 4096)), dim3(256), 0, c->stream,
                                   c->exc_idx.as<long long>(), c->exc_val.as<int32_t>(), c->n_exc, c->cov.as<int32_t>());
            hipLaunchKernelGGL(delta4_walk_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((B / kD4Block + 255) / 256, 256 * 16))), dim3(256), 0, c->stream,
                               B, c->cov_anchor.as<int32_t>(), c->abs_bits.as<unsigned>(), c->cov.as<int32_t>());
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->cov_valid = true;
            return RAFT_HIP_OK;
        }
        if (c->pass_width == 1) hipLaunchKernelGGL(unpack_cov_kernel<uint8_t>, dim3(grid), dim3(256), 0, c->stream, c->cov8.as<uint8_t>(), B, c->cov.as<int32_t>());
        else hipLaunchKernelGGL(unpack_cov_kernel<uint16_t>, dim3(grid), dim3(256), 0, c->stream, c->cov8.as<uint16_t>(), B, c->cov.as<int32_t>());
        if (c->n_exc > 0)
            hipLaunchKernelGGL(scatter_exceptions_kernel, dim3((unsigned)std::min<long long>((c->n_exc + 255) / 256, 4096)), dim3(256), 0, c->stream,
                               c->exc_idx.as<long long>(), c->exc_val.as<int32_t>(), c->n_exc, c->cov.as<int32_t>());
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->cov_valid = true;
    return RAFT_HIP_OK;
}

int raft_hip_set_output_width(raft_hip_ctx *c, int32_t width)
{
    if (!c || (width != 1 && width != 2 && width != 4 && width != kCovDelta4)) return RAFT_HIP_ERR_PARAM;
    c->out_width = width;
    return RAFT_HIP_OK;
}

int raft_hip_set_emit_cuts(raft_hip_ctx *c, int32_t on)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    c->emit_cuts = on != 0;
    return RAFT_HIP_OK;
}

#ifdef RAFT_DEBUG_SWAP
// (investigation only, tools/placement_probe2.py, make DEFS=-DRAFT_DEBUG_SWAP: swaps one scratch buffer between two idle contexts)
extern "C" int raft_hip_debug_swap(raft_hip_ctx *a, raft_hip_ctx *b, int which)
{
    DevBuf raft_hip_ctx::*m[] = {&raft_hip_ctx::wave_ctr, &raft_hip_ctx::ctrl, &raft_hip_ctx::block_sums, &raft_hip_ctx::tile_cuts, &raft_hip_ctx::cov_off,
                                 &raft_hip_ctx::rep_res_off, &raft_hip_ctx::cutcap_off, &raft_hip_ctx::rep_cnt, &raft_hip_ctx::raw_key, &raft_hip_ctx::raw_s,
                                 &raft_hip_ctx::raw_e, &raft_hip_ctx::cov, &raft_hip_ctx::tile_first, &raft_hip_ctx::scan_tmp,
                                 &raft_hip_ctx::samples, &raft_hip_ctx::cut_cnt, &raft_hip_ctx::frag_cnt, &raft_hip_ctx::rep_off,
                                 &raft_hip_ctx::cut_off, &raft_hip_ctx::frag_off, &raft_hip_ctx::rep_s, &raft_hip_ctx::rep_e, &raft_hip_ctx::cuts,
                                 &raft_hip_ctx::frag_read, &raft_hip_ctx::frag_begin, &raft_hip_ctx::frag_end};
    const int n = (int)(sizeof m / sizeof m[0]);
    if (which < 0 || which >= n) return n;
    std::swap(a->*m[which], b->*m[which]);
    return 0;
}
#endif

int raft_hip_device_alloc(raft_hip_ctx *c, int64_t bytes, void **dptr)
{
    if (!c || !dptr || bytes < 0) return RAFT_HIP_ERR_PARAM;
    *dptr = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return RAFT_HIP_ERR_DEVICE;
    DevBuf *b = new (std::nothrow) DevBuf();
    if (!b) return RAFT_HIP_ERR_NOMEM;
    b->big = true;
    if (b->ensure((size_t)std::max<int64_t>(bytes, 1)) != hipSuccess) { (void)hipGetLastError(); delete b; return RAFT_HIP_ERR_NOMEM; }
    c->user_bufs.push_back(b);
    *dptr = b->p;
    return RAFT_HIP_OK;
}

int raft_hip_device_free(raft_hip_ctx *c, void *dptr)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (!dptr) return RAFT_HIP_OK;
    for (size_t i = 0; i < c->user_bufs.size(); ++i)
        if (c->user_bufs[i]->p == dptr) {
            (void)hipSetDevice(c->device);
            c->user_bufs[i]->release();
            delete c->user_bufs[i];
            c->user_bufs.erase(c->user_bufs.begin() + (long)i);
            return RAFT_HIP_OK;
        }
    return RAFT_HIP_ERR_PARAM;
}

int raft_hip_packed_device(raft_hip_ctx *c, int32_t *width, const void **cov_packed, const int64_t **exc_index,
                           const int32_t **exc_value, int64_t *n_exc)
{
    if (!c || !width) return RAFT_HIP_ERR_PARAM;
    if (!c->finished || c->pending_err) return RAFT_HIP_ERR_STATE;
    *width = c->packed_width;
    const bool have = c->packed_width != 0;
    if (cov_packed) *cov_packed = have ? c->cov8.p : nullptr;
    if (exc_index) *exc_index = have ? c->exc_idx.as<int64_t>() : nullptr;
    if (exc_value) *exc_value = have ? c->exc_val.as<int32_t>() : nullptr;
    if (n_exc) *n_exc = have ? c->n_exc : 0;
    return RAFT_HIP_OK;
}

int raft_hip_packed_anchor_device(raft_hip_ctx *c, const int32_t **cov_anchor, int64_t *n_anchor)
{
    if (!c || !cov_anchor) return RAFT_HIP_ERR_PARAM;
    if (!c->finished || c->pending_err) return RAFT_HIP_ERR_STATE;
    const bool have = c->packed_width == kCovDelta4;
    *cov_anchor = have ? c->cov_anchor.as<int32_t>() : nullptr;
    if (n_anchor) *n_anchor = have ? (c->sum.n_bins + kD4Block - 1) / kD4Block : 0;
    return RAFT_HIP_OK;
}

int raft_hip_outputs_device(raft_hip_ctx *c, raft_hip_outputs *o)
{
    if (!c || !o) return RAFT_HIP_ERR_PARAM;
    if (!c->finished || c->pending_err) return RAFT_HIP_ERR_STATE;
    { const int rc = materialise_cuts(c); if (rc != RAFT_HIP_OK) return rc; }
    { const int rc = materialise_cov(c); if (rc != RAFT_HIP_OK) return rc; }
    o->cov_offset = c->cov_off.as<int64_t>(); o->cov = c->cov.as<int32_t>();
    o->rep_offset = c->rep_off.as<int64_t>(); o->rep_s = c->rep_s.as<int32_t>(); o->rep_e = c->rep_e.as<int32_t>();
    o->cut_offset = c->cut_off.as<int64_t>(); o->cuts = c->cuts.as<int32_t>();
    o->frag_offset = c->frag_off.as<int64_t>(); o->frag_read = c->frag_read.as<int32_t>();
    o->frag_begin = c->frag_begin.as<int32_t>(); o->frag_end = c->frag_end.as<int32_t>();
    return RAFT_HIP_OK;
}

int raft_hip_fetch(raft_hip_ctx *c, int64_t *cov_offset, int32_t *cov, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                   int64_t *cut_offset, int32_t *cuts, int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin,
                   int32_t *frag_end)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (!c->finished || c->pending_err) return RAFT_HIP_ERR_STATE;
    HIP_TRY(c, hipSetDevice(c->device));
    if (cuts) { const int rc = materialise_cuts(c); if (rc != RAFT_HIP_OK) return rc; }
    if (cov) { const int rc = materialise_cov(c); if (rc != RAFT_HIP_OK) return rc; }
    const size_t N1 = (size_t)c->sum.n_reads + 1;
    struct { void *dst; const void *src; size_t bytes; } job[] = {
        {cov_offset, c->cov_off.p, N1 * 8}, {cov, c->cov.p, (size_t)c->sum.n_bins * 4},
        {rep_offset, c->rep_off.p, N1 * 8}, {rep_s, c->rep_s.p, (size_t)c->sum.n_repeats * 4},
        {rep_e, c->rep_e.p, (size_t)c->sum.n_repeats * 4}, {cut_offset, c->cut_off.p, N1 * 8},
        {cuts, c->cuts.p, (size_t)c->sum.n_cuts * 4}, {frag_offset, c->frag_off.p, N1 * 8},
        {frag_read, c->frag_read.p, (size_t)c->sum.n_fragments * 4}, {frag_begin, c->frag_begin.p, (size_t)c->sum.n_fragments * 4},
        {frag_end, c->frag_end.p, (size_t)c->sum.n_fragments * 4}};
    for (auto &j : job)                            // all copies queued on the context's stream, one wait
        if (j.dst && j.bytes) HIP_TRY(c, hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RAFT_HIP_OK;
}

// cov[] -> one or two bytes per window + exception list (pack.hpp), on the device, once per pass and width
static int pack_coverage(raft_hip_ctx *c, int width)
{
    if (c->packed_width == width) return RAFT_HIP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    { const int rc = materialise_cov(c); if (rc != RAFT_HIP_OK) return rc; }   // (a pass that wrote the other width)
    const long long B = c->sum.n_bins;
    const bool d4 = width == kCovDelta4;
    HIP_TRY(c, c->cov8.ensure(d4 ? (size_t)std::max(B, 1LL) / 2 + 16 : (size_t)std::max(B, 1LL) * (size_t)width + 16));
    if (d4) HIP_TRY(c, c->cov_anchor.ensure(((size_t)std::max(B, 1LL) / kD4Block + 3) * 4));
    HIP_TRY(c, c->exc_cnt.ensure(8));
    long long cap = std::max<long long>(c->exc_cap, std::max<long long>(4096, d4 ? B / 64 : B / 512));
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(c, c->exc_idx.ensure((size_t)cap * 8));
        HIP_TRY(c, c->exc_val.ensure((size_t)cap * 4));
        c->exc_cap = cap;
        HIP_TRY(c, hipMemsetAsync(c->exc_cnt.p, 0, 8, c->stream));
        if (B > 0) {
            const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((B / 4 + 1023) / 1024, 256 * 16));
            if (d4) {
                Delta4Out po{c->cov8.as<uint8_t>(), c->cov_anchor.as<int32_t>(), c->exc_cnt.as<unsigned long long>(), cap, c->exc_idx.as<long long>(), c->exc_val.as<int32_t>()};
                hipLaunchKernelGGL(pack_delta4_kernel, dim3(grid), dim3(256), 0, c->stream, c->cov.as<int32_t>(), B, po, c->d4_shift);
            } else if (width == 1) {
                PackOut<uint8_t> po{c->cov8.as<uint8_t>(), c->exc_cnt.as<unsigned long long>(), cap, c->exc_idx.as<long long>(), c->exc_val.as<int32_t>()};
                hipLaunchKernelGGL(pack_cov_kernel<uint8_t>, dim3(grid), dim3(256), 0, c->stream, c->cov.as<int32_t>(), B, po);
            } else {
                PackOut<uint16_t> po{c->cov8.as<uint16_t>(), c->exc_cnt.as<unsigned long long>(), cap, c->exc_idx.as<long long>(), c->exc_val.as<int32_t>()};
                hipLaunchKernelGGL(pack_cov_kernel<uint16_t>, dim3(grid), dim3(256), 0, c->stream, c->cov.as<int32_t>(), B, po);
            }
            HIP_TRY(c, hipGetLastError());
        }
        long long *h = reinterpret_cast<long long *>(c->pinned) + kPackCountWord;
        HIP_TRY(c, hipMemcpyAsync(h, c->exc_cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->n_exc = *h; c->exc_sorted = false;
        if (c->n_exc <= cap) break;
        cap = c->n_exc;                              // (rare) more windows at or above the limit than the list held: once more
    }
    c->packed_width = width;
    return RAFT_HIP_OK;
}

// The kernels append exceptions in no particular order; callers get them ascending by window.  With a byte per window there
// are none on a 32x set; the four-bit encoding lists 0.2-0.3 % of the windows (3.7e6 at human scale) and the host's
// std::sort of a chunk's 3.4e5 pairs held its lane for 25 ms: sorted on the device (radix sort on the index bits in use).
static int sort_exceptions(raft_hip_ctx *c)
{
    if (c->exc_sorted || c->n_exc < 2) { c->exc_sorted = true; return RAFT_HIP_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->n_exc;
    HIP_TRY(c, c->exc_idx2.ensure(std::max(n * 8, c->exc_idx.cap)));
    HIP_TRY(c, c->exc_val2.ensure(std::max(n * 4, c->exc_val.cap)));
    int bits = 1;
    while (bits < 63 && (1LL << bits) <= std::max<long long>(c->sum.n_bins, 1)) ++bits;
    using Key = unsigned long long;           // (window indices are non-negative)
    HIP_TRY(c, c->sort_tmp.ensure(sort_pairs_hist_bytes((long long)n)));
    bool in_b = false;
    HIP_TRY(c, sort_pairs(c->stream, c->exc_idx.as<Key>(), c->exc_val.as<int32_t>(), c->exc_idx2.as<Key>(), c->exc_val2.as<int32_t>(), (long long)n, bits,
                          c->sort_tmp.as<int32_t>(), &in_b));
    if (in_b) { std::swap(c->exc_idx, c->exc_idx2); std::swap(c->exc_val, c->exc_val2); }
    c->exc_sorted = true;
    return RAFT_HIP_OK;
}

static int fetch_packed_impl(raft_hip_ctx *c, int32_t width, int64_t *cov_offset, void *cov_packed, int32_t *cov_anchor, int64_t exc_cap, int64_t *exc_index,
                             int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                             int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end)
{
    if (!c || !n_exc || (width != 1 && width != 2 && width != kCovDelta4)) return RAFT_HIP_ERR_PARAM;
    if (!c->finished || c->pending_err) return RAFT_HIP_ERR_STATE;
    { const int rc = pack_coverage(c, width); if (rc != RAFT_HIP_OK) return rc; }
    *n_exc = c->n_exc;
    // (*n_exc tells the caller what to provide; the size query -- every pointer NULL -- always succeeds)
    if (c->n_exc > exc_cap && (cov_packed || exc_index || exc_value)) return RAFT_HIP_ERR_TOO_LARGE;
    if (exc_index || exc_value) { const int rc = sort_exceptions(c); if (rc != RAFT_HIP_OK) return rc; }   // handed out ascending by window
    const size_t N1 = (size_t)c->sum.n_reads + 1;
    const bool d4 = width == kCovDelta4;
    struct { void *dst; const void *src; size_t bytes; } job[] = {
        {cov_packed, c->cov8.p, d4 ? ((size_t)c->sum.n_bins + 1) / 2 : (size_t)c->sum.n_bins * (size_t)width}, {cov_offset, c->cov_off.p, N1 * 8},
        {d4 ? cov_anchor : nullptr, c->cov_anchor.p, (((size_t)c->sum.n_bins + kD4Block - 1) / kD4Block) * 4},
        {exc_index, c->exc_idx.p, (size_t)c->n_exc * 8}, {exc_value, c->exc_val.p, (size_t)c->n_exc * 4},
        {rep_offset, c->rep_off.p, N1 * 8}, {rep_s, c->rep_s.p, (size_t)c->sum.n_repeats * 4},
        {rep_e, c->rep_e.p, (size_t)c->sum.n_repeats * 4}, {frag_offset, c->frag_off.p, N1 * 8},
        {frag_read, c->frag_read.p, (size_t)c->sum.n_fragments * 4}, {frag_begin, c->frag_begin.p, (size_t)c->sum.n_fragments * 4},
        {frag_end, c->frag_end.p, (size_t)c->sum.n_fragments * 4}};
    for (auto &j : job)
        if (j.dst && j.bytes) HIP_TRY(c, hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RAFT_HIP_OK;
}

int raft_hip_fetch_packed_w(raft_hip_ctx *c, int32_t width, int64_t *cov_offset, void *cov_packed, int64_t exc_cap, int64_t *exc_index,
                            int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                            int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end)
{
    if (width != 1 && width != 2) return RAFT_HIP_ERR_PARAM;
    return fetch_packed_impl(c, width, cov_offset, cov_packed, nullptr, exc_cap, exc_index, exc_value, n_exc, rep_offset, rep_s, rep_e, frag_offset,
                             frag_read, frag_begin, frag_end);
}

int raft_hip_fetch_delta4(raft_hip_ctx *c, int64_t *cov_offset, uint8_t *cov_nib, int32_t *cov_anchor, int64_t exc_cap, int64_t *exc_index,
                          int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                          int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end)
{
    if ((cov_nib != nullptr) != (cov_anchor != nullptr)) return RAFT_HIP_ERR_PARAM;
    return fetch_packed_impl(c, kCovDelta4, cov_offset, cov_nib, cov_anchor, exc_cap, exc_index, exc_value, n_exc, rep_offset, rep_s, rep_e, frag_offset,
                             frag_read, frag_begin, frag_end);
}

int raft_hip_fetch_packed(raft_hip_ctx *c, int64_t *cov_offset, uint8_t *cov8, int64_t exc_cap, int64_t *exc_index,
                          int32_t *exc_value, int64_t *n_exc, int64_t *rep_offset, int32_t *rep_s, int32_t *rep_e,
                          int64_t *frag_offset, int32_t *frag_read, int32_t *frag_begin, int32_t *frag_end)
{
    return raft_hip_fetch_packed_w(c, 1, cov_offset, cov8, exc_cap, exc_index, exc_value, n_exc, rep_offset, rep_s, rep_e, frag_offset,
                                   frag_read, frag_begin, frag_end);
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunked host pipeline: H2D, pass and D2H of consecutive read ranges overlap (PCIe is full duplex; the pass itself is
// two orders of magnitude shorter than either transfer).
//
// A read's outputs depend on nothing but the records whose query is that read (symmetric PAF, repeat.hpp:48-58), so the
// job is cut into chunks of consecutive reads.  hifiasm's PAF is a handful of runs sorted by query id (bucket.hpp), so
// a chunk's records are one contiguous piece per run: the pieces are found on the host by binary search in the
// page-locked qid column and uploaded back to back.  The cut is a guess from samples -- what makes it safe is the
// device: the pieces tile [0, n_rec) by construction, and inspect_kernel rejects any record whose (rebased) query id
// falls outside its chunk's reads; on any such report the whole job is redone in one piece.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Piece { long long lo, hi; };

// Sorted runs of the record stream from 8 k samples + bisection; -1 when there are more than kMaxSeg.
int guess_segments(const int32_t *q, long long n, long long (&start)[kMaxSeg + 1])
{
    const long long S = std::min<long long>(n, 8192);
    int n_seg = 1;
    start[0] = 0;
    long long prev_pos = 0;
    for (long long i = 1; i < S; ++i) {
        const long long pos = i * (n - 1) / (S - 1);
        if (q[pos] < q[prev_pos]) {                  // a run ends in (prev_pos, pos]: first position below q[prev_pos]
            long long lo = prev_pos, hi = pos;
            const int32_t v = q[prev_pos];
            while (hi - lo > 1) {
                const long long mid = lo + (hi - lo) / 2;
                if (q[mid] >= v) lo = mid; else hi = mid;
            }
            if (n_seg == kMaxSeg) return -1;
            start[n_seg++] = hi;
        }
        prev_pos = pos;
    }
    start[n_seg] = n;
    return n_seg;
}

long long lower_bound_ids(const int32_t *q, long long lo, long long hi, int32_t r)   // first position in [lo, hi) with q >= r
{
    while (lo < hi) {
        const long long mid = lo + (hi - lo) / 2;
        if (q[mid] < r) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct ChunkPlan {
    int32_t r0, r1;
    Piece piece[kMaxSeg];
    long long n_rec;
    long long win_lo;            // delta4: windows of the reads before r0 (where the chunk's coverage begins in the caller's array)
};

struct ChunkResult {
    long long n_bins = 0, n_rep = 0, n_frag = 0, n_exc = 0, n_cuts = 0, n_iv = 0;
    long long tot_cov = 0, tot_rep = 0, tot_len = 0;
    int path = 0;
};

struct PipeShared {                                 // the chain of one context's chunks (positions within the context's job)
    std::mutex mu, down_mu;
    std::condition_variable cv;
    int uploaded = 0;                               // chunks whose H2D has been enqueued (ticket of the upload stream)
    int published = 0;                              // chunks whose sizes are known (bases of the next chunk)
    long long base_bins = 0, base_rep = 0, base_frag = 0;
    int error = RAFT_HIP_OK;                        // first failure; every lane stops at its next check
    std::string error_text;
};

} // namespace

// windows of n reads: sum ceil(len / reso), the multiply-high division the kernels use (exact for 0 <= len < 2^31); -1 when
// a length is negative (the pass reports it)
static long long count_windows(const int32_t *len, long long n, int32_t reso_i)
{
    const unsigned reso = (unsigned)reso_i;
    int lg = 0;
    while ((1ull << lg) < reso) ++lg;
    const unsigned long long magic = reso > 1 ? ((1ull << (31 + lg)) / reso + 1ull) : 0ull;
    long long w = 0;
    int32_t any_neg = 0;
    for (long long i = 0; i < n; ++i) {
        const unsigned l = (unsigned)len[i];
        any_neg |= len[i];
        const unsigned q = reso == 1 ? l : (unsigned)(((l * magic) >> 32) >> (lg - 1));
        w += (long long)q + (l - q * reso ? 1 : 0);
    }
    return any_neg < 0 ? -1 : w;
}

static int run_host_grouped_impl(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec, int32_t n_runs,
                                 const int64_t *rec_offset, const int32_t *qs, const int32_t *qe, const uint32_t *win, int64_t n_bins)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (n_reads < 0 || n_rec < 0 || n_runs < 1 || n_runs > kMaxRuns || !rec_offset) return RAFT_HIP_ERR_PARAM;
    if (n_reads > 0 && !read_len) return RAFT_HIP_ERR_PARAM;
    if (n_rec > 0 && (!qs || !qe) && !win) return RAFT_HIP_ERR_PARAM;
    if (c->prm.symmetric_mode != 1) return RAFT_HIP_ERR_PARAM;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t n_off = (size_t)n_runs * ((size_t)n_reads + 1);
    HIP_TRY(c, c->in_len.ensure((size_t)std::max<long long>(n_reads, 1) * 4));
    HIP_TRY(c, c->in_off.ensure(n_off * 8));
    if (n_reads) HIP_TRY(c, hipMemcpyAsync(c->in_len.p, read_len, (size_t)n_reads * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->in_off.p, rec_offset, n_off * 8, hipMemcpyHostToDevice, st));
    const void *src[3] = {nullptr, win ? (const void *)win : (const void *)qs, qe};
    for (int k = 1; k < (win ? 2 : 3); ++k) {
        HIP_TRY(c, c->in_col[k].ensure((size_t)std::max<long long>(n_rec, 1) * 4));
        if (n_rec) HIP_TRY(c, hipMemcpyAsync(c->in_col[k].p, src[k], (size_t)n_rec * 4, hipMemcpyHostToDevice, st));
    }
    if (n_bins < 0) n_bins = count_windows(read_len, n_reads, c->prm.reso);      // (while the copies run)
    if (win)
        return run_grouped(c, n_reads, c->in_len.as<int32_t>(), n_rec, n_runs, c->in_off.as<int64_t>(), nullptr, nullptr, nullptr, nullptr, n_bins,
                           c->in_col[1].as<uint32_t>());
    return run_grouped(c, n_reads, c->in_len.as<int32_t>(), n_rec, n_runs, c->in_off.as<int64_t>(), nullptr, nullptr,
                       c->in_col[1].as<int32_t>(), c->in_col[2].as<int32_t>(), n_bins);
}

int raft_hip_run_host_grouped(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec, int32_t n_runs,
                              const int64_t *rec_offset, const int32_t *qs, const int32_t *qe, int64_t n_bins)
{
    return run_host_grouped_impl(c, n_reads, read_len, n_rec, n_runs, rec_offset, qs, qe, nullptr, n_bins);
}

int raft_hip_run_host_windows(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec, int32_t n_runs,
                              const int64_t *rec_offset, const uint32_t *win, int64_t n_bins)
{
    if (n_rec > 0 && !win) return RAFT_HIP_ERR_PARAM;
    static const uint32_t none = 0;
    return run_host_grouped_impl(c, n_reads, read_len, n_rec, n_runs, rec_offset, nullptr, nullptr, win ? win : &none, n_bins);
}

static int run_monolithic_to_host(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec, const int32_t *qid,
                                  const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts, const int32_t *te,
                                  raft_hip_host_outputs *o, raft_hip_summary *summary, int32_t n_runs = 0,
                                  const int64_t *rec_offset = nullptr, const uint32_t *win = nullptr)
{
    const int keep_width = c->out_width;
    const int width = o->cov_width == kCovDelta4 ? kCovDelta4 : (o->cov_width == 2 ? 2 : 1);
    c->out_width = width;                                 // the pass writes the encoding the caller takes
    int rc = rec_offset ? run_host_grouped_impl(c, n_reads, read_len, n_rec, n_runs, rec_offset, qs, qe, win, -1)
                        : raft_hip_run_host(c, n_reads, read_len, n_rec, qid, qs, qe, tid, ts, te);
    raft_hip_summary s{};
    if (rc == RAFT_HIP_OK) rc = raft_hip_finish(c, &s);
    c->out_width = keep_width;
    s.n_devices_used = 1;
    if (summary) *summary = s;
    if (rc != RAFT_HIP_OK) return rc;
    if (s.n_bins > o->cov8_cap || s.n_repeats > o->rep_cap || s.n_fragments > o->frag_cap) return RAFT_HIP_ERR_TOO_LARGE;
    int64_t n_exc = 0;
    if (width == kCovDelta4 && (s.n_bins + kD4Block - 1) / kD4Block > o->anchor_cap) return RAFT_HIP_ERR_TOO_LARGE;
    rc = fetch_packed_impl(c, width, o->cov_offset, o->cov8, width == kCovDelta4 ? o->cov_anchor : nullptr, o->exc_cap, o->exc_index, o->exc_value, &n_exc,
                           o->rep_offset, o->rep_s, o->rep_e, o->frag_offset, nullptr, o->frag_begin, o->frag_end);
    o->n_exc = n_exc;
    return rc;
}

namespace {

constexpr int kLanes = 4;

// Everything one context (one device) does in a multi-context job: its chunks, where its outputs start in the
// caller's arrays, and the chain that hands each chunk the sizes of the chunks before it.
struct DeviceJob {
    raft_hip_ctx *c = nullptr;
    int first_chunk = 0, n_chunks = 0;
    // first entry of this job in the caller's arrays: windows are known in advance (read lengths); repeats, fragments and
    // exceptions are not, so every job after the first starts at an upper bound and is moved down when all are done
    long long bins0 = 0, rep0 = 0, frag0 = 0;
    long long rep_room = 0, frag_room = 0;
    PipeShared sh;
    long long n_bins = 0, n_rep = 0, n_frag = 0;     // totals of the job (valid after the run)
};

int prepare_lanes(raft_hip_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    // Copies get streams of their own priority levels.  The runtime multiplexes streams onto a few hardware queues per
    // priority level, and a copy holds its queue for its whole duration: on a queue shared with a lane's compute stream
    // the kernels of one chunk sat behind the uploads of the next two (measured: 12 ms of a 0.4 ms pass).
    if (!c->up_stream) {
        int lo_p = 0, hi_p = 0;                      // numerically lowest = highest priority
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, hi_p));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->down_stream, hipStreamNonBlocking, lo_p));
    }
    while ((int)c->lanes.size() < kLanes) {
        raft_hip_ctx *l = nullptr;
        const int rc = raft_hip_create(c->device, &c->prm, &l);
        if (rc != RAFT_HIP_OK) return rc;
        c->lanes.push_back(l);
        hipEvent_t e, d;
        HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->lane_up_ev.push_back(e);
        HIP_TRY(c, hipEventCreateWithFlags(&d, hipEventDisableTiming));
        c->lane_down_ev.push_back(d);
    }
    for (raft_hip_ctx *l : c->lanes) {
        apply_params(l, &c->prm);
        l->tile_q = c->tile_q; l->force_bucket = 0;
        l->is_lane = true;
        l->emit_cuts = false;                         // (raft_hip_host_outputs holds no cut points)
    }
    return RAFT_HIP_OK;
}

} // namespace


// ---------------------------------------------------------------------------------------------------------------------
// Host-routed jobs for record streams that are NOT a handful of runs sorted by query id (a shuffled PAF, a non-symmetric
// one, more than four concatenated files): SURVEY.md §8(e)'s host-routed mode in its general form.  create_pileup's
// bucketing (chop.hpp:155-169: every record into its query's bucket and, while the PAF is not symmetric, into its
// target's) is done by the host's threads as a counting sort by read id -- counts, offsets, scatter -- which leaves the
// intervals grouped by read: consecutive read ranges are then contiguous slices, each a sorted run of its own, and go
// to the contexts (devices) in turn as one-piece passes of the sorted-segment path; a chain of tickets hands each chunk
// the sizes of the chunks before it.  This is also what lifts the 2^29-records-per-pass limit for such inputs.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

void host_parallel(int n_tasks, const std::function<void(int)> &fn)
{
    std::vector<std::thread> th;
    for (int t = 1; t < n_tasks; ++t) th.emplace_back([&fn, t] { fn(t); });
    if (n_tasks > 0) fn(0);
    for (auto &x : th) x.join();
}

} // namespace

// *fallback = true: nothing was done and the caller should take the one-piece pass (which reports data errors exactly).
static int run_routed(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                      const int32_t *qid, const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts, const int32_t *te,
                      int32_t n_chunks, raft_hip_host_outputs *o, raft_hip_summary *summary, bool *fallback)
{
    raft_hip_ctx *c = ctxs[0];
    *fallback = false;
    const int mode = c->prm.symmetric_mode;
    if (mode != 1 && (!tid || !ts || !te)) return RAFT_HIP_ERR_PARAM;
    const bool one_pass_possible = n_rec < (1LL << 29);
    const int cov_width = o->cov_width == 2 ? 2 : 1;
    int T = (int)std::min<long long>(std::max(1u, std::thread::hardware_concurrency()), 32);
    if (n_rec < (1 << 18)) T = 1;
    // ---- ids in range?  the mirror of record 0 (chop.hpp:171-184) when the caller did not say
    std::vector<long long> bad((size_t)T, -1);
    std::vector<int> mirror((size_t)T, 0);
    host_parallel(T, [&](int t) {
        const long long lo = n_rec * t / T, hi = n_rec * (t + 1) / T;
        const bool detect = mode < 0 && n_rec > 0;
        const int32_t q0 = detect ? qid[0] : 0, t0 = detect ? tid[0] : 0, qs0 = detect ? qs[0] : 0, qe0 = detect ? qe[0] : 0, ts0 = detect ? ts[0] : 0,
                      te0 = detect ? te[0] : 0;
        for (long long i = lo; i < hi; ++i) {
            const bool okq = (uint32_t)qid[i] < (uint32_t)n_reads, okt = mode == 1 || (uint32_t)tid[i] < (uint32_t)n_reads;
            if (!(okq && okt)) { if (bad[(size_t)t] < 0) bad[(size_t)t] = i; continue; }
            if (detect && i > 0 && qid[i] == t0 && tid[i] == q0 && ts[i] == qs0 && te[i] == qe0 && qs[i] == ts0 && qe[i] == te0) mirror[(size_t)t] = 1;
        }
    });
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t] >= 0) {
            if (one_pass_possible) { *fallback = true; return RAFT_HIP_OK; }
            raft_hip_summary s{};
            s.n_reads = n_reads; s.n_records = n_rec; s.high_cov = c->high_cov; s.error_index = bad[(size_t)t];
            if (summary) *summary = s;
            return RAFT_HIP_ERR_READ_ID;
        }
    int sym = mode == 1 ? 1 : 0;
    if (mode < 0) for (int t = 0; t < T; ++t) sym |= mirror[(size_t)t];
    // ---- counting sort by read id on the host: counts, offsets, scatter (symmetric: query sides; else also target sides of
    // records whose two reads differ -- bucket.hpp's multiset)
    std::vector<long long> pre;
    std::unique_ptr<int32_t[]> cur, b_rid, b_s, b_e;
    long long total = 0;
    try {
        pre.assign((size_t)n_reads + 1, 0);
        cur.reset(new int32_t[(size_t)n_reads + 1]());
    } catch (const std::bad_alloc &) { return RAFT_HIP_ERR_NOMEM; }
    host_parallel(T, [&](int t) {
        const long long lo = n_rec * t / T, hi = n_rec * (t + 1) / T;
        for (long long i = lo; i < hi; ++i) {
            __atomic_fetch_add(&cur[(size_t)qid[i]], 1, __ATOMIC_RELAXED);
            if (!sym && tid[i] != qid[i]) __atomic_fetch_add(&cur[(size_t)tid[i]], 1, __ATOMIC_RELAXED);
        }
    });
    for (int32_t r = 0; r < n_reads; ++r) {
        if (cur[(size_t)r] < 0) return RAFT_HIP_ERR_TOO_LARGE;          // (2^31 intervals on one read)
        pre[(size_t)r + 1] = pre[(size_t)r] + cur[(size_t)r];
        cur[(size_t)r] = 0;
    }
    total = pre[(size_t)n_reads];
    try {
        b_rid.reset(new int32_t[(size_t)std::max(total, 1LL)]); b_s.reset(new int32_t[(size_t)std::max(total, 1LL)]);
        b_e.reset(new int32_t[(size_t)std::max(total, 1LL)]);
    } catch (const std::bad_alloc &) { return RAFT_HIP_ERR_NOMEM; }
    host_parallel(T, [&](int t) {
        const long long lo = n_rec * t / T, hi = n_rec * (t + 1) / T;
        auto put = [&](int32_t r, int32_t s0, int32_t e0) {
            const long long d = pre[(size_t)r] + __atomic_fetch_add(&cur[(size_t)r], 1, __ATOMIC_RELAXED);
            b_rid[(size_t)d] = r; b_s[(size_t)d] = s0; b_e[(size_t)d] = e0;
        };
        for (long long i = lo; i < hi; ++i) {
            put(qid[i], qs[i], qe[i]);
            if (!sym && tid[i] != qid[i]) put(tid[i], ts[i], te[i]);
        }
    });
    cur.reset();
    // ---- plan: consecutive read ranges of near-equal interval counts, each far below the per-pass limit
    const long long per_pass = 1LL << 27;
    long long want = std::max<long long>(std::max<long long>(n_chunks, n_ctx), (total + per_pass - 1) / per_pass);
    want = std::max<long long>(1, std::min<long long>(want, std::max(n_reads, 1)));
    std::vector<int32_t> bound{0};
    for (long long k = 1; k < want; ++k) {
        const long long target = total * k / want;
        const int32_t r = (int32_t)(std::lower_bound(pre.begin(), pre.end(), target) - pre.begin());
        if (r > bound.back() && r < n_reads) bound.push_back(r);
    }
    bound.push_back(n_reads);
    const int n_ch = (int)bound.size() - 1;
    for (int k = 0; k < n_ch; ++k)
        if (pre[(size_t)bound[(size_t)k + 1]] - pre[(size_t)bound[(size_t)k]] >= (1LL << 29)) return RAFT_HIP_ERR_TOO_LARGE;   // (one read's pile alone)
    const int n_job = std::min(n_ctx, std::max(n_ch, 1));
    raft_hip_params prm1 = c->prm;
    prm1.symmetric_mode = 1;                              // the routed intervals ARE the multiset to pile up: query-side records
    std::vector<raft_hip_params> keep((size_t)n_job);
    for (int d = 0; d < n_job; ++d) {
        keep[(size_t)d] = ctxs[d]->prm;
        const int rc0 = raft_hip_set_params(ctxs[d], &prm1);
        if (rc0 != RAFT_HIP_OK) {
            for (int e = 0; e < d; ++e) (void)raft_hip_set_params(ctxs[e], &keep[(size_t)e]);
            return rc0;
        }
        ctxs[d]->tile_q = c->tile_q;
    }
    // ---- chunk k runs on context k % n_job; a ticket chain publishes the sizes in chunk order
    std::mutex mu;
    std::condition_variable cv;
    int published = 0, err = RAFT_HIP_OK;
    long long err_index = -1;
    bool data_error = false;
    long long base_bins = 0, base_rep = 0, base_frag = 0, base_exc = 0;
    raft_hip_summary tot{};
    tot.n_reads = n_reads; tot.symmetric = sym; tot.high_cov = c->high_cov; tot.n_records = n_rec; tot.error_index = -1;
    tot.interval_path = 0; tot.n_segments = 1; tot.n_devices_used = n_job;
    std::string err_text;
    auto job_main = [&](int d) {
        raft_hip_ctx *jc = ctxs[d];
        const int keep_width = jc->out_width;
        jc->out_width = cov_width;
        auto fail = [&](int code, long long index, bool data) {
            std::lock_guard<std::mutex> g(mu);
            if (err == RAFT_HIP_OK) { err = code; err_index = index; data_error = data; err_text = jc->last_error; }
            cv.notify_all();
        };
        for (int k = d; k < n_ch; k += n_job) {
            { std::lock_guard<std::mutex> g(mu); if (err != RAFT_HIP_OK) break; }
            const int32_t r0 = bound[(size_t)k], r1 = bound[(size_t)k + 1], nr = r1 - r0;
            const long long i0 = pre[(size_t)r0], n_iv = pre[(size_t)r1] - i0;
            int rc = RAFT_HIP_OK;
            raft_hip_summary s{};
            hipError_t e = hipSetDevice(jc->device);
            if (e == hipSuccess) e = jc->in_len.ensure((size_t)std::max(nr, 1) * 4);
            for (int col = 0; col < 3 && e == hipSuccess; ++col) e = jc->in_col[col].ensure((size_t)std::max<long long>(n_iv, 1) * 4);
            if (e == hipSuccess && nr) e = hipMemcpyAsync(jc->in_len.p, read_len + r0, (size_t)nr * 4, hipMemcpyHostToDevice, jc->stream);
            const int32_t *src[3] = {b_rid.get() + i0, b_s.get() + i0, b_e.get() + i0};
            for (int col = 0; col < 3 && e == hipSuccess && n_iv; ++col)
                e = hipMemcpyAsync(jc->in_col[col].p, src[col], (size_t)n_iv * 4, hipMemcpyHostToDevice, jc->stream);
            if (e != hipSuccess) { fail(fail_hip(jc, e, "run_routed: staging"), -1, false); break; }
            if (n_iv > 0 && r0 != 0)
                hipLaunchKernelGGL(rebase_ids_kernel, dim3((unsigned)std::min<long long>((n_iv + 255) / 256, 4096)), dim3(256), 0, jc->stream,
                                   jc->in_col[0].as<int32_t>(), n_iv, r0);
            rc = raft_hip_run_device(jc, nr, jc->in_len.as<int32_t>(), n_iv, jc->in_col[0].as<int32_t>(), jc->in_col[1].as<int32_t>(),
                                     jc->in_col[2].as<int32_t>(), nullptr, nullptr, nullptr);
            if (rc == RAFT_HIP_OK) rc = raft_hip_finish(jc, &s);
            if (rc != RAFT_HIP_OK) {
                // (a data error's index counts the routed intervals, not the caller's records: the one-piece pass reports it
                // properly when the input is small enough for one)
                fail(rc, -1, rc == RAFT_HIP_ERR_COORD || rc == RAFT_HIP_ERR_FRAGMENT || rc == RAFT_HIP_ERR_PARAM || rc == RAFT_HIP_ERR_READ_ID);
                break;
            }
            long long b_bins, b_rep, b_frag, b_exc;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return published == k || err != RAFT_HIP_OK; });
                if (err != RAFT_HIP_OK) break;
                b_bins = base_bins; b_rep = base_rep; b_frag = base_frag; b_exc = base_exc;
            }
            // sizes of the encoding's exception list are known only after it has been made (raft_hip_fetch_packed_w's size query)
            int64_t n_exc = 0;
            rc = raft_hip_fetch_packed_w(jc, cov_width, nullptr, nullptr, 0, nullptr, nullptr, &n_exc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
            const bool fits = rc == RAFT_HIP_OK && b_bins + s.n_bins <= o->cov8_cap && b_rep + s.n_repeats <= o->rep_cap &&
                              b_frag + s.n_fragments <= o->frag_cap;
            const bool exc_fits = b_exc + n_exc <= o->exc_cap;
            if (rc == RAFT_HIP_OK && !fits) { jc->last_error = "host output capacity (coverage / repeats / fragments)"; rc = RAFT_HIP_ERR_TOO_LARGE; }
            if (rc == RAFT_HIP_OK)
                rc = raft_hip_fetch_packed_w(jc, cov_width, o->cov_offset + r0, o->cov8 ? o->cov8 + b_bins * cov_width : nullptr, n_exc,   // (a list that no longer fits is counted, not fetched: the call ends with TOO_LARGE and the job's total)
                                            
                                             (exc_fits && o->exc_index) ? o->exc_index + b_exc : nullptr, (exc_fits && o->exc_value) ? o->exc_value + b_exc : nullptr,
                                             &n_exc, o->rep_offset + r0, o->rep_s ? o->rep_s + b_rep : nullptr, o->rep_e ? o->rep_e + b_rep : nullptr,
                                             o->frag_offset + r0, nullptr, o->frag_begin ? o->frag_begin + b_frag : nullptr,
                                             o->frag_end ? o->frag_end + b_frag : nullptr);
            if (rc != RAFT_HIP_OK) { fail(rc, -1, false); break; }
            // (the fetch wrote nr + 1 offsets counting from this chunk's first entry: the closing one is the next chunk's first)
            for (int32_t r = 0; r < nr + (k == n_ch - 1 ? 1 : 0); ++r) {
                o->cov_offset[r0 + r] += b_bins; o->rep_offset[r0 + r] += b_rep; o->frag_offset[r0 + r] += b_frag;
            }
            if (exc_fits && o->exc_index) for (int64_t i = 0; i < n_exc; ++i) o->exc_index[b_exc + i] += b_bins;
            {
                std::lock_guard<std::mutex> g(mu);
                base_bins += s.n_bins; base_rep += s.n_repeats; base_frag += s.n_fragments; base_exc += n_exc;
                tot.n_bins += s.n_bins; tot.n_repeats += s.n_repeats; tot.n_fragments += s.n_fragments; tot.n_cuts += s.n_cuts;
                tot.n_intervals += s.n_intervals; tot.total_coverage += s.total_coverage; tot.total_repeat_length += s.total_repeat_length;
                tot.total_read_length += s.total_read_length;
                published = k + 1;
                cv.notify_all();
            }
        }
        jc->out_width = keep_width;
        jc->ran = false; jc->finished = false;           // the context holds no pass of the caller's
    };
    {
        std::vector<std::thread> th;
        for (int d = 1; d < n_job; ++d) th.emplace_back([&, d] { job_main(d); });
        job_main(0);
        for (auto &t : th) t.join();
    }
    for (int d = 0; d < n_job; ++d) (void)raft_hip_set_params(ctxs[d], &keep[(size_t)d]);
    (void)hipSetDevice(c->device);
    if (err != RAFT_HIP_OK) {
        if (data_error && one_pass_possible) { *fallback = true; return RAFT_HIP_OK; }
        c->last_error = err_text;
        tot.error_index = err_index;
        if (summary) *summary = tot;
        return err;
    }
    // (a chunk that wrote its closing offsets before its successor wrote its first ones: the successor's are the same values)
    tot.total_windows = tot.n_bins;
    o->n_exc = base_exc;
    if (summary) *summary = tot;
    if (base_exc > o->exc_cap) {
        c->last_error = "raft_hip_run_multi: more windows at or above the encoding's limit than exc_cap (out->n_exc holds the number)";
        return RAFT_HIP_ERR_TOO_LARGE;
    }
    return RAFT_HIP_OK;
}

// What the engine's host side derives from the plain columns of a symmetric, sorted stream before they cross the link (SURVEY.md
// §8(d): the clock of a host-to-host job starts at the int32 columns): per piece of a chunk -- records [lo, hi) of one sorted run,
// reads [r0, r1) -- where every read's records begin (the grouped form of raft_hip_run_device_grouped) and the records as
// window records (one word: first window | one past the last << 16; repeat.hpp:69-72 uses nothing else of an interval).  4 bytes
// per record go up instead of 12, and the pass needs no look at the stream.  The ids are checked on the way (inside the
// chunk's reads, never stepping back): anything else, a negative coordinate or a window beyond 16 bits sends the job to the
// one-piece pass over the columns, which reports or handles it.  T threads share the piece.
// (Two loops, the first branch-free so that the compiler vectorises it: the window indices by multiply-high -- n / reso ==
// (n * m) >> (31 + L) for 0 <= n < 2^31, the identity the kernels use; a hardware division per coordinate made the derivation
// compute-bound at 10 cycles per record -- with the error conditions collected, not branched on; then the id column for the places
// where the read changes.)
static inline __attribute__((always_inline)) bool derive_body(int t, int T, const int32_t *qid, const int32_t *qs, const int32_t *qe, long long lo, long long hi,
                                                              int32_t r0, int32_t r1, int32_t reso, long long at, long long *off, uint32_t *win)
{
    const long long n = hi - lo;
    const int32_t nr = r1 - r0;
    if (n <= 0) { if (t == 0) for (int32_t j = 0; j <= nr; ++j) off[j] = at; return true; }
    const long long a = lo + n * t / T, b = lo + n * (t + 1) / T;
    if (a >= b) return true;
    int L = 0;
    while ((1u << L) < (uint32_t)reso) ++L;
    const uint64_t m = reso > 1 ? ((1ull << (31 + L)) / (uint32_t)reso + 1ull) : 1ull;
    const int sh = reso > 1 ? 31 + L : 0;
    {
        const int32_t *ps = qs + a, *pe = qe + a;
        uint32_t *pw = win + (a - lo);
        const long long cnt = b - a;
        uint32_t neg = 0, far = 0;
        for (long long i = 0; i < cnt; ++i) {
            const int32_t s0 = ps[i], e0 = pe[i];
            neg |= (uint32_t)(s0 | e0);
            const uint32_t first = (uint32_t)(((uint64_t)(uint32_t)s0 * m) >> sh);
            const uint32_t em = (uint32_t)(e0 > 0 ? e0 - 1 : 0);
            const uint32_t last1 = e0 > 0 ? (uint32_t)(((uint64_t)em * m) >> sh) + 1u : 0u;
            const uint32_t w = last1 > first ? (first | (last1 << 16)) : 0u;
            far |= last1 > first ? last1 : 0u;
            pw[i] = w;
        }
        if ((neg >> 31) || (far >> 16)) return false;      // a negative coordinate; a window index beyond 16 bits
    }
    int32_t prev = a == lo ? r0 - 1 : qid[a - 1];
    if (prev < r0 - 1 || prev >= r1) return false;
    for (long long i = a; i < b; ++i) {
        const int32_t q = qid[i];
        if (q != prev) {
            if (q < prev || q >= r1) return false;
            for (int32_t r = prev + 1; r <= q; ++r) off[r - r0] = at + (i - lo);      // (reads without records begin where the next one does)
            prev = q;
        }
    }
    if (b == hi) for (int32_t r = prev + 1; r <= r1; ++r) off[r - r0] = at + n;           // closing entries
    return true;
}
__attribute__((target("avx2"))) static bool derive_slice_avx2(int t, int T, const int32_t *qid, const int32_t *qs, const int32_t *qe, long long lo, long long hi,
                                                              int32_t r0, int32_t r1, int32_t reso, long long at, long long *off, uint32_t *win)
{
    return derive_body(t, T, qid, qs, qe, lo, hi, r0, r1, reso, at, off, win);
}
static bool derive_slice(int t, int T, const int32_t *qid, const int32_t *qs, const int32_t *qe, long long lo, long long hi, int32_t r0, int32_t r1,
                         int32_t reso, long long at, long long *off, uint32_t *win)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return derive_slice_avx2(t, T, qid, qs, qe, lo, hi, r0, r1, reso, at, off, win);
    return derive_body(t, T, qid, qs, qe, lo, hi, r0, r1, reso, at, off, win);
}

// The chunks of one context's job are derived in order by T workers that stay for the whole job -- worker t takes the t-th slice
// of every piece -- into a ring of page-locked staging slots; a lane uploads chunk k when all workers are through with it and
// hands its slot back when the upload is done.  (The first version had every lane derive its own chunk with threads made for
// the purpose: four derivations at a time, each behind its lane's previous chunk, left the link idle a third of the time.)
struct DeriveRing {
    static constexpr int R = 3;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> done;            // workers through with chunk k
    std::vector<char> released, bad;
    bool stop = false;
    int T = 1;
    size_t slot_bytes = 0, off_bytes = 0;
    char *base = nullptr;
    long long *off_of(int kk) const { return reinterpret_cast<long long *>(base + (size_t)(kk % R) * slot_bytes); }
    uint32_t *win_of(int kk) const { return reinterpret_cast<uint32_t *>(base + (size_t)(kk % R) * slot_bytes + off_bytes); }
};

// (n_runs, rec_offset): the grouped form -- the caller's offsets instead of the query column (raft_hip_run_multi_grouped)
static int run_multi_impl(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                          const int32_t *qid, const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts,
                          const int32_t *te, int32_t n_runs, const int64_t *rec_offset, int32_t n_chunks, raft_hip_host_outputs *o,
                          raft_hip_summary *summary, const uint32_t *win = nullptr)
{
    if (!ctxs || n_ctx < 1 || !ctxs[0] || !o) return RAFT_HIP_ERR_PARAM;
    raft_hip_ctx *c = ctxs[0];
    const bool grouped = rec_offset != nullptr;
    if (win && (!grouped || c->prm.reso > 32767)) return RAFT_HIP_ERR_PARAM;
    if (grouped && (n_runs < 1 || n_runs > kMaxRuns || ctxs[0]->prm.symmetric_mode != 1)) return RAFT_HIP_ERR_PARAM;
    const long long ostride = (long long)n_reads + 1;
    auto off_at = [&](int g, long long r) -> long long { return rec_offset[(long long)g * ostride + r]; };
    for (int d = 1; d < n_ctx; ++d) {
        if (!ctxs[d]) return RAFT_HIP_ERR_PARAM;
        for (int e = 0; e < d; ++e) if (ctxs[e] == ctxs[d]) return RAFT_HIP_ERR_PARAM;   // (two contexts may share a device)
    }
    if (n_reads < 0 || n_rec < 0 || n_chunks < 0) return RAFT_HIP_ERR_PARAM;
    if (n_reads > 0 && !read_len) return RAFT_HIP_ERR_PARAM;
    if (n_rec > 0 && ((!qid && !grouped) || ((!qs || !qe) && !win))) return RAFT_HIP_ERR_PARAM;
    if (!o->cov_offset || !o->rep_offset || !o->frag_offset) return RAFT_HIP_ERR_PARAM;
    o->n_exc = 0;
    auto one_piece = [&]() { return run_monolithic_to_host(c, n_reads, read_len, n_rec, qid, qs, qe, tid, ts, te, o, summary, n_runs, rec_offset, win); };
    // (more runs than the chunk plan keeps pieces for -- a PAF concatenated from many files: one piece, merged on the device)
    if (grouped && n_runs > kMaxSeg) return n_rec < (1LL << 29) ? one_piece() : RAFT_HIP_ERR_TOO_LARGE;
    if (o->cov_width != 0 && o->cov_width != 1 && o->cov_width != 2 && o->cov_width != kCovDelta4) return RAFT_HIP_ERR_PARAM;
    const bool d4 = o->cov_width == kCovDelta4;        // four-bit steps (pack.hpp): chunks must begin on multiples of 1024 windows
    if (d4 && (!o->cov_anchor || !o->cov8)) return RAFT_HIP_ERR_PARAM;
    const int cov_width = d4 ? kCovDelta4 : (o->cov_width == 2 ? 2 : 1);   // bytes per window of the coverage's transfer encoding (or the delta4 code)
    long long seg[kMaxSeg + 1];
    int n_seg = -1;
    // chunking needs: the symmetric flag asserted, enough work to split, a record stream of at most kMaxSeg sorted runs
    // (an explicit n_chunks is honoured from tiny inputs on: that is how the tests reach every shape of the plan)
    const bool big_enough = n_chunks > 0 ? (n_rec >= 2 && n_reads >= 2) : (n_rec >= (1 << 24) && n_reads >= 4096);   // (~200 MB up: below that one piece is as fast)
    const bool eligible = c->prm.symmetric_mode == 1 && big_enough && !c->force_bucket;
    if (eligible && grouped) {                       // the runs are what the offsets say (looked at where the plan uses them)
        n_seg = n_runs;
        for (int g = 0; g < n_runs; ++g) seg[g] = off_at(g, 0);
        seg[n_runs] = n_rec;
        for (int g = 0; g < n_runs; ++g)
            if (seg[g] < 0 || seg[g] > seg[g + 1] || off_at(g, n_reads) != seg[g + 1]) n_seg = -1;   // (the one-piece pass reports it)
        if (seg[0] != 0) n_seg = -1;
    } else if (eligible) n_seg = guess_segments(qid, n_rec, seg);
    if (n_seg < 1) {
        // not a handful of sorted runs (or not symmetric): several contexts, an explicit chunk count or more records than one
        // pass takes send the job through the host-routed path; anything else is one piece on the first context
        // (the routed path cuts its chunks where the host's buckets end: no multiples of 1024 windows -- delta4 stays in one piece)
        const bool route = !grouped && !d4 && n_rec > 0 && n_reads > 0 && !c->force_bucket &&
                           ((big_enough && (n_ctx > 1 || n_chunks > 1)) || n_rec >= (1LL << 29));
        if (route) {
            bool fallback = false;
            const int rc = run_routed(ctxs, n_ctx, n_reads, read_len, n_rec, qid, qs, qe, tid, ts, te, n_chunks, o, summary, &fallback);
            if (!fallback) return rc;
        }
        return one_piece();
    }

    // plain columns of a symmetric stream in a few sorted runs: the lanes derive offsets and window records chunk by chunk (above)
    bool derive = !grouped && !win && c->prm.symmetric_mode == 1 && c->prm.reso <= 32767 && n_seg <= kWinMaxRuns && getenv("RAFT_NO_DERIVE") == nullptr;
    if (derive) {
        const long long max_len = 65535LL * c->prm.reso;      // (reads of more windows than a record's 16 bits hold keep their coordinate columns)
        std::atomic<bool> fits{true};
        const int Tl = (int)std::min<long long>(16, std::max<long long>(1, n_reads / (1 << 18)));
        host_parallel(Tl, [&](int t) {
            const long long a = (long long)n_reads * t / Tl, b = (long long)n_reads * (t + 1) / Tl;
            bool f = true;
            for (long long i = a; i < b; ++i) f = f && read_len[i] <= max_len;
            if (!f) fits.store(false);
        });
        derive = fits.load();
    }
    int want = n_chunks > 0 ? std::min(n_chunks, n_reads)
                            : (int)std::min<long long>(std::min<long long>(32LL * n_ctx, std::max<long long>(2LL * n_ctx, n_rec / (24LL << 20))),
                                                       n_reads / 1024);
    // ---- plan: read boundaries that balance the records, then one piece per run and chunk
    std::vector<ChunkPlan> plan;
    {
        auto first_of = [&](int g, long long lo, int32_t r) {   // first record of read r in run g, at or after lo
            if (!grouped) return lower_bound_ids(qid, lo, seg[g + 1], r);
            return std::min(std::max(off_at(g, r), lo), seg[g + 1]);   // (offsets that step back: the device reports them)
        };
        auto below = [&](int32_t r) {                // records with a query id < r (if the runs are sorted)
            long long n = 0;
            for (int k = 0; k < n_seg; ++k) n += first_of(k, seg[k], r) - seg[k];
            return n;
        };
        std::vector<int32_t> bound{0};
        // (derived input: the first chunk's derivation and the last chunk's pass and download are not hidden behind anything --
        // those two chunks are half the others' size)
        const bool ramp = derive && n_chunks == 0 && want >= 6;
        for (int k = 1; k < want; ++k) {
            const long long target = ramp ? (long long)((double)n_rec * (k - 0.5) / (want - 1.0)) : n_rec * k / want;
            int32_t lo = bound.back(), hi = n_reads;
            while (lo < hi) {
                const int32_t mid = lo + (hi - lo) / 2;
                if (below(mid) < target) lo = mid + 1; else hi = mid;
            }
            if (lo > bound.back() && lo < n_reads) bound.push_back(lo);
        }
        bound.push_back(n_reads);
        std::vector<long long> win_before;           // delta4: windows before every boundary
        if (d4) {
            // delta4: a chunk's windows must begin on a multiple of 4 (its nibbles fill whole ushorts of the caller's array;
            // the anchors' blocks may begin anywhere, see PileupArgs::d4_shift): every inner boundary moves forward to the
            // next read that does -- a few reads on.  The windows before the boundaries are counted by one thread per chunk.
            const size_t nb = bound.size() - 1;
            std::vector<long long> wsum(nb, 0);
            host_parallel((int)nb, [&](int k) { wsum[(size_t)k] = count_windows(read_len + bound[(size_t)k], bound[(size_t)k + 1] - bound[(size_t)k], c->prm.reso); });
            std::vector<int32_t> moved{0};
            win_before.push_back(0);
            long long before = 0;                       // windows before the ORIGINAL boundary k
            bool ok = true;
            for (size_t k = 1; k < nb && ok; ++k) {
                ok = wsum[k - 1] >= 0;
                before += wsum[k - 1];
                int32_t r = bound[k];
                long long w = before;
                if (r <= moved.back()) continue;        // (an earlier boundary moved past this one: dropped)
                while (ok && (w & 3) != 0 && r < n_reads) {
                    const long long one = count_windows(read_len + r, 1, c->prm.reso);
                    if (one < 0) ok = false;
                    w += one; ++r;
                }
                if (ok && r < n_reads && (w & 3) == 0) {
                    // (boundaries after this one still count from their ORIGINAL place: `before` is not touched)
                    moved.push_back(r); win_before.push_back(w);
                }
            }
            if (!ok || wsum[nb - 1] < 0) return one_piece();   // (a negative read length: reported by the one-piece pass)
            moved.push_back(n_reads);
            bound.swap(moved);
        }
        std::vector<long long> cur(seg, seg + n_seg);
        for (size_t k = 0; k + 1 < bound.size(); ++k) {
            ChunkPlan cp{};
            cp.r0 = bound[k]; cp.r1 = bound[k + 1]; cp.n_rec = 0;
            cp.win_lo = d4 ? win_before[k] : 0;
            for (int g = 0; g < n_seg; ++g) {
                const long long hi = (k + 2 == bound.size()) ? seg[g + 1] : first_of(g, cur[g], cp.r1);
                cp.piece[g] = Piece{cur[g], hi};
                cp.n_rec += hi - cur[g];
                cur[g] = hi;
            }
            plan.push_back(cp);
        }
    }
    const int n_ch = (int)plan.size();
    if (n_ch < 2) return one_piece();
    int derive_threads = 1;
    if (derive) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        derive_threads = (int)std::max(1u, std::min(16u, hw / 4u));
        if (const char *e = getenv("RAFT_DERIVE_THREADS")) derive_threads = std::max(1, atoi(e));
    }

    // ---- contexts: consecutive chunks each (the plan balances records per chunk), parameters of the first
    const int n_job = std::min(n_ctx, n_ch);
    std::vector<DeviceJob> jobs((size_t)n_job);
    {
        // Where each context's outputs start in the caller's arrays.  Windows are exact (sum of ceil(len / reso) over the
        // reads before: one multiply-high per read, the division the kernels use); repeats and fragments start at the
        // bounds of raft_hip.h and are moved down when all contexts are done.  One context needs none of this.
        const long long minw = c->minbins, L = c->prm.interval_length;
        const unsigned reso = (unsigned)c->prm.reso;
        int lg = 0;
        while ((1ull << lg) < reso) ++lg;
        const unsigned long long magic = reso > 1 ? ((1ull << (31 + lg)) / reso + 1ull) : 0ull;
        auto windows = [&](int32_t len) -> long long {       // exact for 0 <= len < 2^31 (engine.hip run_pass, div_magic)
            if (reso == 1) return len;
            const unsigned q = (unsigned)((((unsigned long long)(unsigned)len * magic) >> 32) >> (lg - 1));
            return (long long)q + ((unsigned)len - q * reso ? 1 : 0);
        };
        long long bins = 0, rep_cap = 0, frag_cap = 0;
        int r = 0;
        for (int d = 0; d < n_job; ++d) {
            DeviceJob &J = jobs[(size_t)d];
            J.c = ctxs[d];
            J.first_chunk = n_ch * d / n_job; J.n_chunks = n_ch * (d + 1) / n_job - J.first_chunk;
            if (d > 0) {
                const int rc0 = raft_hip_set_params(J.c, &c->prm);
                if (rc0 != RAFT_HIP_OK) return rc0;
                J.c->tile_q = c->tile_q;
            }
            J.bins0 = bins; J.rep0 = rep_cap; J.frag0 = frag_cap;
            if (n_job > 1) {
                const int r_end = plan[(size_t)(J.first_chunk + J.n_chunks - 1)].r1;
                long long jb = 0, jl = 0;
                for (; r < r_end; ++r) {
                    if (read_len[r] < 0)             // (reported as RAFT_HIP_ERR_PARAM with its index by the one-piece pass)
                        return one_piece();
                    jb += windows(read_len[r]); jl += read_len[r];
                }
                const long long n_r = r_end - plan[(size_t)J.first_chunk].r0;
                // sum floor(x_i / m) <= floor(sum x_i / m): the per-read bounds of raft_hip.h, summed, are at least these
                J.rep_room = (jb + n_r) / (minw + 1); J.frag_room = jl / L + 2 * n_r;
                bins += jb; rep_cap += J.rep_room; frag_cap += J.frag_room;
            }
            const int rc = prepare_lanes(J.c);
            if (rc != RAFT_HIP_OK) return rc;
        }
        if (n_job == 1) {                            // one context: the caller's capacities are the only limits
            jobs[0].rep_room = o->rep_cap; jobs[0].frag_room = o->frag_cap;
        } else if (rep_cap > o->rep_cap || frag_cap > o->frag_cap || (o->cov8 && bins > o->cov8_cap) || (d4 && (bins + kD4Block - 1) / kD4Block > o->anchor_cap)) {
            c->last_error = "raft_hip_run_multi: cov8_cap / rep_cap / frag_cap below the bounds stated in raft_hip.h";
            return RAFT_HIP_ERR_TOO_LARGE;
        }
    }

    std::vector<std::unique_ptr<DeriveRing>> rings((size_t)n_job);
    if (derive) {
        for (int d = 0; d < n_job; ++d) {
            DeviceJob &J = jobs[(size_t)d];
            auto ring = std::make_unique<DeriveRing>();
            size_t off_b = 0, win_b = 0;
            for (int kk = 0; kk < J.n_chunks; ++kk) {
                const ChunkPlan &cp = plan[(size_t)(J.first_chunk + kk)];
                off_b = std::max(off_b, (size_t)n_seg * ((size_t)(cp.r1 - cp.r0) + 1) * 8);
                win_b = std::max(win_b, (size_t)std::max<long long>(cp.n_rec, 1) * 4);
            }
            ring->off_bytes = (off_b + 255) & ~(size_t)255;
            ring->slot_bytes = (ring->off_bytes + win_b + 255) & ~(size_t)255;
            const size_t need = ring->slot_bytes * DeriveRing::R;
            raft_hip_ctx *jc = J.c;
            if (need > jc->h_stage_cap) {
                HIP_TRY(jc, hipSetDevice(jc->device));
                if (jc->h_stage) (void)hipHostFree(jc->h_stage);
                jc->h_stage = nullptr; jc->h_stage_cap = 0;
                HIP_TRY(jc, hipHostMalloc(&jc->h_stage, need + need / 8, hipHostMallocDefault));
                jc->h_stage_cap = need + need / 8;
            }
            ring->base = reinterpret_cast<char *>(jc->h_stage);
            ring->T = derive_threads;
            ring->done.assign((size_t)J.n_chunks, 0); ring->released.assign((size_t)J.n_chunks, 0); ring->bad.assign((size_t)J.n_chunks, 0);
            rings[(size_t)d] = std::move(ring);
        }
        (void)hipSetDevice(c->device);
    }
    auto derive_worker = [&](int d, int t) {
        DeviceJob &J = jobs[(size_t)d];
        DeriveRing &R = *rings[(size_t)d];
        for (int kk = 0; kk < J.n_chunks; ++kk) {
            {
                std::unique_lock<std::mutex> g(R.mu);
                R.cv.wait(g, [&] { return R.stop || kk < DeriveRing::R || R.released[(size_t)(kk - DeriveRing::R)]; });
                if (R.stop) return;
            }
            const ChunkPlan &cp = plan[(size_t)(J.first_chunk + kk)];
            const int32_t nr = cp.r1 - cp.r0;
            long long at = 0;
            bool good = true;
            for (int g2 = 0; g2 < n_seg; ++g2) {
                good = derive_slice(t, R.T, qid, qs, qe, cp.piece[g2].lo, cp.piece[g2].hi, cp.r0, cp.r1, c->prm.reso, at, R.off_of(kk) + (long long)g2 * (nr + 1),
                                    R.win_of(kk) + at) && good;
                at += cp.piece[g2].hi - cp.piece[g2].lo;
            }
            {
                std::lock_guard<std::mutex> g(R.mu);
                if (!good) R.bad[(size_t)kk] = 1;
                if (++R.done[(size_t)kk] == R.T) R.cv.notify_all();
            }
        }
    };
    std::vector<ChunkResult> res((size_t)n_ch);
    // Exceptions (windows at or above the encoding's limit) have no useful bound per device -- one device may hold all the
    // repeat-rich reads -- so every chunk takes its room from ONE cursor over the caller's list; chunks of different
    // devices interleave there and are put into read order when all are done.  A chunk that no longer fits still counts:
    // the call then returns RAFT_HIP_ERR_TOO_LARGE with the total in out->n_exc, and one retry suffices.
    std::atomic<long long> exc_cursor{0};
    std::vector<long long> exc_at((size_t)n_ch, 0);
    std::atomic<bool> redo{false};                  // a chunk reported a data error: the job is redone in one piece
    const bool trace = getenv("RAFT_PIPE_TRACE") != nullptr;   // host-clock stamps per chunk and stage on stderr
    const auto t_origin = std::chrono::steady_clock::now();
    auto stamp = [&](int k, const char *what) {
        if (trace) fprintf(stderr, "PIPE chunk %2d %-12s %8.3f ms\n", k, what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_origin).count());
    };

    auto lane_main = [&](DeviceJob &J, int li) {
        raft_hip_ctx *jc = J.c;
        raft_hip_ctx *l = jc->lanes[(size_t)li];
        PipeShared &sh = J.sh;
        auto fail = [&](int code, const std::string &text) {
            std::lock_guard<std::mutex> g(sh.mu);
            if (sh.error == RAFT_HIP_OK) { sh.error = code; sh.error_text = text; }
            sh.cv.notify_all();
        };
        auto stop = [&]() { return sh.error != RAFT_HIP_OK || redo.load(); };
        if (hipSetDevice(jc->device) != hipSuccess) { fail(RAFT_HIP_ERR_DEVICE, "hipSetDevice"); return; }
#define LANE_TRY(expr)                                                                  \
        do {                                                                            \
            hipError_t e_ = (expr);                                                     \
            if (e_ != hipSuccess) { fail(fail_hip(l, e_, #expr), l->last_error); goto out; } \
        } while (0)
        for (int kk = li; kk < J.n_chunks; kk += kLanes) {
            const int k = J.first_chunk + kk;        // global chunk index; kk = position in this job's chain
            const ChunkPlan &cp = plan[(size_t)k];
            ChunkResult &cr = res[(size_t)k];
            const int32_t nr = cp.r1 - cp.r0;
            raft_hip_summary s{};
            long long b_bins, b_rep, b_frag, b_exc;
            bool exc_fits;
            hipStream_t st = l->stream;
            // -- upload, in chunk order on the one upload stream (the link is the bottleneck: first come, first served)
            LANE_TRY(l->in_len.ensure((size_t)std::max(nr, 1) * 4));
            const int col_end = (win || derive) ? 2 : 3;         // (window records: one column)
            for (int col = (grouped || derive) ? 1 : 0; col < col_end; ++col) LANE_TRY(l->in_col[col].ensure((size_t)std::max<long long>(cp.n_rec, 1) * 4));
            if (grouped || derive) LANE_TRY(l->in_off.ensure((size_t)n_seg * ((size_t)nr + 1) * 8));
            long long *st_off = nullptr;
            uint32_t *st_win = nullptr;
            if (derive) {
                // the chunk's offsets and window records: derived by the job's workers while earlier chunks travel
                DeriveRing &R = *rings[(size_t)(&J - &jobs[0])];
                bool bad_chunk = false, stopped = false;
                {
                    std::unique_lock<std::mutex> g(R.mu);
                    R.cv.wait(g, [&] { return R.stop || R.done[(size_t)kk] == R.T; });
                    stopped = R.stop; bad_chunk = R.bad[(size_t)kk] != 0;
                }
                stamp(k, "derived");
                if (stopped) goto out;
                if (bad_chunk) { redo.store(true); goto out; }      // (the one-piece pass over the columns reports or handles it)
                st_off = R.off_of(kk); st_win = R.win_of(kk);
            }
            {
                std::unique_lock<std::mutex> g(sh.mu);
                sh.cv.wait(g, [&] { return sh.uploaded == kk || stop(); });
                if (stop()) goto out;
            }
            {
                hipError_t e = hipMemcpyAsync(l->in_len.p, read_len + cp.r0, (size_t)nr * 4, hipMemcpyHostToDevice, jc->up_stream);
                if (derive) {
                    if (e == hipSuccess) e = hipMemcpyAsync(l->in_off.p, st_off, (size_t)n_seg * ((size_t)nr + 1) * 8, hipMemcpyHostToDevice, jc->up_stream);
                    if (e == hipSuccess && cp.n_rec > 0) e = hipMemcpyAsync(l->in_col[1].p, st_win, (size_t)cp.n_rec * 4, hipMemcpyHostToDevice, jc->up_stream);
                }
                const int32_t *src[3] = {qid, win ? reinterpret_cast<const int32_t *>(win) : qs, qe};
                // (grouped: a slice of every run's offsets instead of the query column -- 8 bytes per read and run, not 4 per record)
                for (int g = 0; grouped && !derive && g < n_seg && e == hipSuccess; ++g)
                    e = hipMemcpyAsync(l->in_off.as<long long>() + (long long)g * (nr + 1), rec_offset + (long long)g * ostride + cp.r0,
                                       (size_t)(nr + 1) * 8, hipMemcpyHostToDevice, jc->up_stream);
                for (int col = grouped ? 1 : 0; !derive && col < col_end && e == hipSuccess; ++col) {
                    long long at = 0;
                    for (int g = 0; g < n_seg && e == hipSuccess; ++g) {
                        const long long n = cp.piece[g].hi - cp.piece[g].lo;
                        if (n > 0) e = hipMemcpyAsync(l->in_col[col].as<int32_t>() + at, src[col] + cp.piece[g].lo, (size_t)n * 4,
                                                      hipMemcpyHostToDevice, jc->up_stream);
                        at += n;
                    }
                }
                if (e == hipSuccess) e = hipEventRecord(jc->lane_up_ev[(size_t)li], jc->up_stream);
                stamp(k, "h2d queued");
                {
                    std::lock_guard<std::mutex> g(sh.mu);
                    sh.uploaded = kk + 1;
                    sh.cv.notify_all();
                }
                LANE_TRY(e);
            }
            // The lane's thread waits for the upload itself.  A wait-event parked in the lane's stream would sit in a
            // hardware queue that other lanes' streams share, and hold THEIR kernels until this chunk's upload is done
            // (measured: chunks whose pass was queued at 12 ms ran at 24 ms).
            LANE_TRY(hipEventSynchronize(jc->lane_up_ev[(size_t)li]));
            stamp(k, "h2d done");
            if (derive) {
                DeriveRing &R = *rings[(size_t)(&J - &jobs[0])];
                std::lock_guard<std::mutex> g(R.mu);
                R.released[(size_t)kk] = 1;
                R.cv.notify_all();
            }
            if (cp.n_rec > 0 && cp.r0 != 0 && !grouped && !derive) {
                const unsigned grid = (unsigned)std::min<long long>((cp.n_rec + 255) / 256, 4096);
                hipLaunchKernelGGL(rebase_ids_kernel, dim3(grid), dim3(256), 0, st, l->in_col[0].as<int32_t>(), cp.n_rec, cp.r0);
            }
            // -- the pass on this chunk
            {
                l->out_width = cov_width;            // the pass writes the encoding that travels
                l->d4_shift = d4 ? (int)(cp.win_lo & (kD4Block - 1)) : 0;
                int rc;
                if (derive) {
                    // (the staged offsets count from the chunk's own first record: nothing to rebase)
                    const long long hint = count_windows(read_len + cp.r0, nr, c->prm.reso);
                    rc = run_grouped(l, nr, l->in_len.as<int32_t>(), cp.n_rec, n_seg, l->in_off.as<int64_t>(), nullptr, nullptr, nullptr, nullptr, hint,
                                     cp.n_rec > 0 ? l->in_col[1].as<uint32_t>() : nullptr);
                } else if (grouped) {
                    // the chunk's pieces lie back to back on the device: run g's slice of offsets counts from the caller's
                    // stream and is moved by adj[g] to where the piece went
                    long long adj[kMaxSeg] = {0, 0, 0, 0}, at = 0;
                    for (int g = 0; g < n_seg; ++g) { adj[g] = at - cp.piece[g].lo; at += cp.piece[g].hi - cp.piece[g].lo; }
                    const long long hint = count_windows(read_len + cp.r0, nr, c->prm.reso);
                    if (win) rc = run_grouped(l, nr, l->in_len.as<int32_t>(), cp.n_rec, n_seg, l->in_off.as<int64_t>(), adj, nullptr, nullptr, nullptr, hint,
                                              l->in_col[1].as<uint32_t>());
                    else rc = run_grouped(l, nr, l->in_len.as<int32_t>(), cp.n_rec, n_seg, l->in_off.as<int64_t>(), adj, nullptr,
                                          l->in_col[1].as<int32_t>(), l->in_col[2].as<int32_t>(), hint);
                } else
                    rc = raft_hip_run_device(l, nr, l->in_len.as<int32_t>(), cp.n_rec, l->in_col[0].as<int32_t>(),
                                             l->in_col[1].as<int32_t>(), l->in_col[2].as<int32_t>(), nullptr, nullptr, nullptr);
                stamp(k, "pass queued");
                if (rc == RAFT_HIP_OK) rc = raft_hip_finish(l, &s);
                stamp(k, "pass done");
                if (rc == RAFT_HIP_ERR_READ_ID || rc == RAFT_HIP_ERR_COORD || rc == RAFT_HIP_ERR_FRAGMENT || rc == RAFT_HIP_ERR_PARAM) {
                    // a data error (or a record outside the chunk it was cut into): the one-piece run reports it properly
                    redo.store(true);
                    goto out;
                }
                if (rc != RAFT_HIP_OK) { fail(rc, l->last_error); goto out; }
                rc = pack_coverage(l, cov_width);
                if (rc == RAFT_HIP_OK) rc = sort_exceptions(l);          // (ascending by window, like raft_hip_fetch_packed)
                if (rc != RAFT_HIP_OK) { fail(rc, l->last_error); goto out; }
                stamp(k, "packed");
            }
            cr.n_bins = s.n_bins; cr.n_rep = s.n_repeats; cr.n_frag = s.n_fragments; cr.n_exc = l->n_exc; cr.n_cuts = s.n_cuts;
            cr.n_iv = s.n_intervals; cr.tot_cov = s.total_coverage; cr.tot_rep = s.total_repeat_length; cr.tot_len = s.total_read_length;
            cr.path = s.interval_path;
            // -- where this chunk's outputs go: after those of the job's earlier chunks
            {
                std::unique_lock<std::mutex> g(sh.mu);
                sh.cv.wait(g, [&] { return sh.published == kk || stop(); });
                if (stop()) goto out;
                b_bins = J.bins0 + sh.base_bins; b_rep = J.rep0 + sh.base_rep; b_frag = J.frag0 + sh.base_frag;
                b_exc = exc_cursor.fetch_add(cr.n_exc);
                exc_at[(size_t)k] = b_exc;
                exc_fits = b_exc + cr.n_exc <= o->exc_cap;
                sh.base_bins += cr.n_bins; sh.base_rep += cr.n_rep; sh.base_frag += cr.n_frag;
                sh.published = kk + 1;
                if (sh.base_rep > J.rep_room || sh.base_frag > J.frag_room ||
                    (o->cov8 && J.bins0 + sh.base_bins > o->cov8_cap) ||
                    (d4 && (J.bins0 + sh.base_bins + kD4Block - 1) / kD4Block > o->anchor_cap)) {
                    if (sh.error == RAFT_HIP_OK) { sh.error = RAFT_HIP_ERR_TOO_LARGE; sh.error_text = "host output capacity (coverage / repeats / fragments)"; }
                }
                sh.cv.notify_all();
                if (sh.error != RAFT_HIP_OK) goto out;
            }
            {
                const long long n1 = (long long)nr + ((k == n_ch - 1) ? 1 : 0);   // the closing entry belongs to the last chunk
                auto add_base = [&](DevBuf &b, long long n, long long base) {
                    if (base != 0 && n > 0)
                        hipLaunchKernelGGL(add_base_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, st,
                                           b.as<long long>(), n, base);
                };
                // offsets count from the job's first entry (rep / frag of later jobs are moved down afterwards)
                add_base(l->cov_off, n1, b_bins); add_base(l->rep_off, n1, b_rep - J.rep0); add_base(l->frag_off, n1, b_frag - J.frag0);
                add_base(l->exc_idx, cr.n_exc, b_bins);
                const int d4_sh = l->d4_shift, d4_j0 = d4_sh ? 1 : 0;
                if (d4 && b_bins != cp.win_lo) { fail(RAFT_HIP_ERR_DEVICE, "delta4: a chunk's windows do not begin where the plan put them"); goto out; }
                struct { void *dst; const void *src; size_t bytes; } job[] = {
                    {o->cov8 ? o->cov8 + (d4 ? b_bins / 2 : b_bins * cov_width) : nullptr, l->cov8.p,
                     d4 ? ((size_t)cr.n_bins + 1) / 2 : (size_t)cr.n_bins * (size_t)cov_width},
                    // (anchors: the block the chunk begins in belongs to the chunk before unless it begins with it)
                    {d4 ? o->cov_anchor + (b_bins - d4_sh) / kD4Block + d4_j0 : nullptr, l->cov_anchor.as<int32_t>() + d4_j0,
                     d4 ? (size_t)(((long long)d4_sh + cr.n_bins + kD4Block - 1) / kD4Block - d4_j0) * 4 : 0},
                    {o->cov_offset + cp.r0, l->cov_off.p, (size_t)n1 * 8},
                    {(o->exc_index && exc_fits) ? o->exc_index + b_exc : nullptr, l->exc_idx.p, (size_t)cr.n_exc * 8},
                    {(o->exc_value && exc_fits) ? o->exc_value + b_exc : nullptr, l->exc_val.p, (size_t)cr.n_exc * 4},
                    {o->rep_offset + cp.r0, l->rep_off.p, (size_t)n1 * 8},
                    {o->rep_s ? o->rep_s + b_rep : nullptr, l->rep_s.p, (size_t)cr.n_rep * 4},
                    {o->rep_e ? o->rep_e + b_rep : nullptr, l->rep_e.p, (size_t)cr.n_rep * 4},
                    {o->frag_offset + cp.r0, l->frag_off.p, (size_t)n1 * 8},
                    {o->frag_begin ? o->frag_begin + b_frag : nullptr, l->frag_begin.p, (size_t)cr.n_frag * 4},
                    {o->frag_end ? o->frag_end + b_frag : nullptr, l->frag_end.p, (size_t)cr.n_frag * 4}};
                // the download stream takes over once the lane's last kernel is done; the lane waits for its own copies only
                stamp(k, "bases known");
                LANE_TRY(hipEventRecord(jc->lane_down_ev[(size_t)li], st));
                {
                    std::lock_guard<std::mutex> g(sh.down_mu);       // one chunk's copies stay together on the stream
                    LANE_TRY(hipStreamWaitEvent(jc->down_stream, jc->lane_down_ev[(size_t)li], 0));
                    stamp(k, "d2h wait set");
                    for (auto &j : job)
                        if (j.dst && j.bytes) { LANE_TRY(hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, jc->down_stream)); if (trace) stamp(k, "d2h copy"); }
                    LANE_TRY(hipEventRecord(jc->lane_down_ev[(size_t)li], jc->down_stream));
                }
                stamp(k, "d2h queued");
                LANE_TRY(hipEventSynchronize(jc->lane_down_ev[(size_t)li]));
                stamp(k, "d2h done");
            }
        }
    out:
#undef LANE_TRY
        {   // a lane that stops early must not leave the others waiting for its tickets
            std::lock_guard<std::mutex> g(sh.mu);
            sh.cv.notify_all();
        }
        if (derive && (sh.error != RAFT_HIP_OK || redo.load())) {      // ... nor the workers for its slots
            DeriveRing &R = *rings[(size_t)(&J - &jobs[0])];
            std::lock_guard<std::mutex> g(R.mu);
            R.stop = true;
            R.cv.notify_all();
        }
    };

    {
        std::vector<std::thread> th, workers;
        if (derive)
            for (int d = 0; d < n_job; ++d)
                for (int t = 0; t < derive_threads; ++t) workers.emplace_back([&, d, t] { derive_worker(d, t); });
        for (int d = 0; d < n_job; ++d)
            for (int li = 0; li < kLanes; ++li)
                if (d || li) th.emplace_back([&, d, li] { lane_main(jobs[(size_t)d], li); });
        lane_main(jobs[0], 0);
        for (auto &t : th) t.join();
        if (derive)
            for (int d = 0; d < n_job; ++d) {       // (a job that ended early leaves workers waiting for slots)
                std::lock_guard<std::mutex> g(rings[(size_t)d]->mu);
                rings[(size_t)d]->stop = true;
                rings[(size_t)d]->cv.notify_all();
            }
        for (auto &t : workers) t.join();
    }
    int err = RAFT_HIP_OK;
    for (DeviceJob &J : jobs) {
        (void)hipSetDevice(J.c->device);
        (void)hipStreamSynchronize(J.c->up_stream);
        (void)hipStreamSynchronize(J.c->down_stream);
        for (raft_hip_ctx *l : J.c->lanes) (void)hipStreamSynchronize(l->stream);
        J.c->ran = false; J.c->finished = false;   // the contexts hold no pass: fetch / outputs_device do not apply
        J.n_bins = J.sh.base_bins; J.n_rep = J.sh.base_rep; J.n_frag = J.sh.base_frag;
        if (J.sh.error != RAFT_HIP_OK && err == RAFT_HIP_OK) { err = J.sh.error; c->last_error = J.sh.error_text; }
    }
    (void)hipSetDevice(c->device);
    if (redo.load()) return one_piece();

    raft_hip_summary s{};
    s.n_reads = n_reads; s.symmetric = 1; s.high_cov = c->high_cov; s.n_segments = n_seg; s.n_records = n_rec; s.error_index = -1;
    s.n_devices_used = n_job;
    for (const ChunkResult &cr : res) {
        s.n_bins += cr.n_bins; s.n_repeats += cr.n_rep; s.n_fragments += cr.n_frag; s.n_cuts += cr.n_cuts; s.n_intervals += cr.n_iv;
        s.total_coverage += cr.tot_cov; s.total_repeat_length += cr.tot_rep; s.total_read_length += cr.tot_len;
        s.interval_path |= cr.path;
    }
    s.total_windows = s.n_bins;
    if (summary) *summary = s;
    if (err != RAFT_HIP_OK) return err;
    o->n_exc = exc_cursor.load();
    if (o->n_exc > o->exc_cap) {
        c->last_error = "raft_hip_run_multi: more windows at or above the encoding's limit than exc_cap (out->n_exc holds the number)";
        return RAFT_HIP_ERR_TOO_LARGE;
    }
    // ---- exceptions: chunks of different devices took their room in the order they finished; hand them out in read order
    if (n_job > 1 && o->n_exc > 0) {
        bool ordered = true;
        long long at = 0;
        for (int k = 0; k < n_ch; ++k) { ordered = ordered && exc_at[(size_t)k] == at; at += res[(size_t)k].n_exc; }
        if (!ordered) {
            std::vector<int64_t> ti((size_t)o->n_exc);
            std::vector<int32_t> tv((size_t)o->n_exc);
            at = 0;
            for (int k = 0; k < n_ch; ++k) {
                const long long n = res[(size_t)k].n_exc, from = exc_at[(size_t)k];
                if (o->exc_index) memcpy(ti.data() + at, o->exc_index + from, (size_t)n * 8);
                if (o->exc_value) memcpy(tv.data() + at, o->exc_value + from, (size_t)n * 4);
                at += n;
            }
            if (o->exc_index) memcpy(o->exc_index, ti.data(), (size_t)o->n_exc * 8);
            if (o->exc_value) memcpy(o->exc_value, tv.data(), (size_t)o->n_exc * 4);
        }
    }
    // ---- later jobs wrote repeats / fragments at their upper-bound positions: close the gaps
    {
        long long rep_at = jobs[0].n_rep, frag_at = jobs[0].n_frag;
        for (int d = 1; d < n_job; ++d) {
            DeviceJob &J = jobs[(size_t)d];
            const int32_t ra = plan[(size_t)J.first_chunk].r0, rb = plan[(size_t)(J.first_chunk + J.n_chunks - 1)].r1;
            auto move32 = [](int32_t *a, long long to, long long from, long long n) { if (a && n && to != from) memmove(a + to, a + from, (size_t)n * 4); };
            move32(o->rep_s, rep_at, J.rep0, J.n_rep); move32(o->rep_e, rep_at, J.rep0, J.n_rep);
            move32(o->frag_begin, frag_at, J.frag0, J.n_frag); move32(o->frag_end, frag_at, J.frag0, J.n_frag);
            const int32_t r_hi = rb + ((d == n_job - 1) ? 1 : 0);
            for (int32_t r = ra; r < r_hi; ++r) { o->rep_offset[r] += rep_at; o->frag_offset[r] += frag_at; }
            rep_at += J.n_rep; frag_at += J.n_frag;
        }
    }
    return RAFT_HIP_OK;
}

// What the first job of a fresh process pays once -- the engine's code object going to the device at the first launch, the
// four lanes (sub-contexts with their streams, events and page-locked blocks), the small per-context buffers -- is 70-80 ms
// on the MI355X box: five times the work of a 4.4e7-record job.  The CLI calls this beside the tokenising of its inputs.
int raft_hip_warm_up(raft_hip_ctx *c)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    int rc = prepare_lanes(c);
    if (rc != RAFT_HIP_OK) return rc;
    {   // the copy engines behind the pipeline's two copy streams come up at their first large copy (measured: the first
        // 40 MB download of a process sat 10 ms in hipMemcpyAsync)
        HIP_TRY(c, hipSetDevice(c->device));
        void *h = nullptr, *d = nullptr;
        const size_t n = 4u << 20;
        if (hipHostMalloc(&h, n, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, n) == hipSuccess) {
            (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->up_stream);
            (void)hipStreamSynchronize(c->up_stream);
            (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->down_stream);
            (void)hipStreamSynchronize(c->down_stream);
        }
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        (void)hipGetLastError();
    }
    const int32_t len[2] = {400, 300}, qs[2] = {0, 10}, qe[2] = {120, 200};
    const int64_t off[3] = {0, 1, 2};
    std::vector<raft_hip_ctx *> all(c->lanes);
    all.push_back(c);
    for (raft_hip_ctx *l : all) {
        const raft_hip_params keep = l->prm;
        const raft_hip_params p1{50, 30, 1.5, 10000, 10000, 20000, 500, 1000, 1};   // (the reference's defaults: the two reads stay whole)
        apply_params(l, &p1);
        const int keep_width = l->out_width;
        for (int w = 1; w <= 2 && rc == RAFT_HIP_OK; ++w) {           // (both widths of the transfer encoding: their own kernels)
            l->out_width = w;
            rc = raft_hip_run_host_grouped(l, 2, len, 2, 1, off, qs, qe, -1);
            raft_hip_summary s{};
            if (rc == RAFT_HIP_OK) rc = raft_hip_finish(l, &s);
        }
        l->out_width = keep_width;
        apply_params(l, &keep);
        l->ran = false; l->finished = false;
        if (rc != RAFT_HIP_OK) { c->last_error = l->last_error; break; }
    }
    return rc;
}

// The device buffers of a job, allocated ahead of it: ~35 allocations per lane (5 ms), the staging of a chunk's columns
// (hundreds of MB: 2 ms each) -- inside the first job's clock unless somebody knows its shape earlier.  The CLI does, after
// loading the reads: their lengths, and the record count to within a few per cent from the size of the overlaps file.  A
// pass over the expected chunk's reads WITHOUT records sizes everything that follows the reads; the record-sized buffers
// are sized directly.  Buffers only grow, so an estimate that falls short costs what it would have cost anyway.
int raft_hip_reserve(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec_estimate, int32_t n_ctx, int32_t cov_width)
{
    if (!c || n_reads < 0 || (n_reads > 0 && !read_len) || n_rec_estimate < 0 || n_ctx < 1) return RAFT_HIP_ERR_PARAM;
    if (n_reads == 0) return RAFT_HIP_OK;
    const bool chunked = n_rec_estimate >= (1 << 24) && n_reads >= 4096;        // (run_multi_impl's own rule)
    long long chunks = 1;
    if (chunked) chunks = std::max<long long>(1, std::min<long long>(std::min<long long>(32LL * n_ctx, std::max<long long>(2LL * n_ctx, n_rec_estimate / (24LL << 20))), n_reads / 1024));
    const int32_t nr = (int32_t)std::min<long long>(n_reads, n_reads / chunks + n_reads / chunks / 4 + 64);
    const long long nrec = n_rec_estimate / chunks + n_rec_estimate / chunks / 4 + 1024;
    int rc = RAFT_HIP_OK;
    std::vector<raft_hip_ctx *> who;
    if (chunked) {
        rc = prepare_lanes(c);
        if (rc != RAFT_HIP_OK) return rc;
        const long long per_ctx = (chunks + n_ctx - 1) / n_ctx;
        for (int li = 0; li < std::min<long long>(kLanes, per_ctx); ++li) who.push_back(c->lanes[(size_t)li]);
    } else who.push_back(c);
    std::vector<int64_t> zeros((size_t)nr + 1, 0);
    for (raft_hip_ctx *l : who) {
        HIP_TRY(l, hipSetDevice(l->device));
        for (int col = 1; col < 3; ++col) HIP_TRY(l, l->in_col[col].ensure((size_t)nrec * 4));
        HIP_TRY(l, l->exp_qid.ensure((size_t)nrec * 4));
        HIP_TRY(l, l->in_off.ensure((size_t)kMaxSeg * ((size_t)nr + 1) * 8));
        const raft_hip_params keep = l->prm;
        raft_hip_params p1 = c->prm;
        p1.symmetric_mode = 1;
        apply_params(l, &p1);
        const int keep_width = l->out_width;
        l->out_width = cov_width == kCovDelta4 ? kCovDelta4 : (cov_width == 2 ? 2 : 1);
        rc = raft_hip_run_host_grouped(l, nr, read_len, 0, 1, zeros.data(), nullptr, nullptr, -1);
        raft_hip_summary s{};
        if (rc == RAFT_HIP_OK) rc = raft_hip_finish(l, &s);
        l->out_width = keep_width;
        apply_params(l, &keep);
        l->ran = false; l->finished = false;
        if (rc == RAFT_HIP_ERR_NOMEM || rc == RAFT_HIP_ERR_DEVICE) { c->last_error = l->last_error; return rc; }   // (data errors are the job's to report)
    }
    return RAFT_HIP_OK;
}

int raft_hip_run_multi(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                       const int32_t *qid, const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts,
                       const int32_t *te, int32_t n_chunks, raft_hip_host_outputs *o, raft_hip_summary *summary)
{
    return run_multi_impl(ctxs, n_ctx, n_reads, read_len, n_rec, qid, qs, qe, tid, ts, te, 0, nullptr, n_chunks, o, summary);
}

int raft_hip_run_multi_grouped(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                               int32_t n_runs, const int64_t *rec_offset, const int32_t *qs, const int32_t *qe, int32_t n_chunks,
                               raft_hip_host_outputs *o, raft_hip_summary *summary)
{
    if (!rec_offset) return RAFT_HIP_ERR_PARAM;
    return run_multi_impl(ctxs, n_ctx, n_reads, read_len, n_rec, nullptr, qs, qe, nullptr, nullptr, nullptr, n_runs, rec_offset, n_chunks, o,
                          summary);
}

int raft_hip_run_multi_windows(raft_hip_ctx *const *ctxs, int32_t n_ctx, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                               int32_t n_runs, const int64_t *rec_offset, const uint32_t *win, int32_t n_chunks,
                               raft_hip_host_outputs *o, raft_hip_summary *summary)
{
    if (!rec_offset || (n_rec > 0 && !win)) return RAFT_HIP_ERR_PARAM;
    static const uint32_t none = 0;
    return run_multi_impl(ctxs, n_ctx, n_reads, read_len, n_rec, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, n_runs, rec_offset, n_chunks, o,
                          summary, win ? win : &none);
}

int raft_hip_run_pipelined(raft_hip_ctx *c, int32_t n_reads, const int32_t *read_len, int64_t n_rec, const int32_t *qid,
                           const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts, const int32_t *te,
                           int32_t n_chunks, raft_hip_host_outputs *o, raft_hip_summary *summary)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    return raft_hip_run_multi(&c, 1, n_reads, read_len, n_rec, qid, qs, qe, tid, ts, te, n_chunks, o, summary);
}

// Page-locking of caller memory.  The host pipelines move gigabytes each way; from pageable memory the runtime stages them
// through its own bounce buffers.  Measured on the MI355X box (tools/pin_rate.py): hipHostRegister pins pages that have been
// touched at ~120 GB/s (16 ms for 2 GB) and untouched ones at ~20 GB/s (their first touch), after which copies run at the
// link's 53 GB/s.
int raft_hip_host_register(void *ptr, uint64_t bytes)
{
    if (!ptr || bytes == 0) return RAFT_HIP_ERR_PARAM;
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable);
    if (e == hipSuccess) return RAFT_HIP_OK;
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? RAFT_HIP_ERR_NOMEM : RAFT_HIP_ERR_DEVICE;
}

int raft_hip_host_unregister(void *ptr)
{
    if (!ptr) return RAFT_HIP_ERR_PARAM;
    if (hipHostUnregister(ptr) == hipSuccess) return RAFT_HIP_OK;
    (void)hipGetLastError();
    return RAFT_HIP_ERR_DEVICE;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pre-split PAF (BASELINE configs[3]): every rank holds a contiguous slice of the record stream -- in its
// grouped form: per sorted run of the slice, where every read's records begin -- and the reads are owned by ranks in
// contiguous ranges bounds[g] .. bounds[g+1].  A run sorted by read id is sorted by OWNER too, so what rank p has for rank g
// is one contiguous piece per run: nothing is bucketed, copied or sorted before it leaves -- the pieces of the two
// coordinate columns go out from where they lie, with the matching slice of the run's offsets (rebased by the receiver),
// and what arrives is grouped input again: one run per (peer, run) with records for this rank.  More than kMaxSeg of
// them are merged on the device by the pass itself (bucket.hpp merge_runs_kernel).  The query ids never travel.
//   raft_hip_exchange        one process per GPU: RCCL -- counts by ncclAllGather, payload by grouped ncclSend / ncclRecv
//                            over xGMI (librccl is loaded when first used: half a gigabyte that a single-GPU run never maps)
//   raft_hip_exchange_local  one process, several contexts: peer copies (hipMemcpyPeerAsync over xGMI)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // (a process that has PyTorch-ROCm loaded gets that one: same soname)
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        bool ok = true;
        auto sym = [&](const char *n) { void *p = dlsym(h, n); ok = ok && p; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = ok;
    });
    return api.ok ? &api : nullptr;
}

struct RunBases { long long base[kMaxRuns]; };

// off[k][r] = base[k] + raw[k][r] - raw[k][0]: a received slice of a peer's offsets counts from that peer's stream
__global__ __launch_bounds__(256) void rebase_offsets_kernel(int32_t n_runs, long long n1, const long long *raw, RunBases b, long long *off)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1 * n_runs) return;
    const int k = (int)(i / n1);
    off[i] = b.base[k] + raw[i] - raw[(long long)k * n1];
}

struct XRun { int peer, run; long long lo, n; };            // a run this rank receives: records [lo, lo + n) of peer's run

bool slice_ok(const raft_hip_slice &sl, int32_t n_reads_total)
{
    // (d_qe == NULL: d_qs holds window records, one word per record -- raft_hip_run_device_windows' form; the same on every rank)
    return sl.n_rec >= 0 && sl.n_runs >= 1 && sl.n_runs <= kMaxSeg && sl.rec_offset && (sl.n_rec == 0 || sl.d_qs) && n_reads_total >= 0;
}

} // namespace

// A slice of a non-symmetric PAF as ONE run sorted by read id, in grouped form (see include/raft_hip.h): the expansion and the
// sort are the general bucketing path's (bucket.hpp expand_sides_kernel / unzip_sorted_kernel around the device radix sort).
int raft_hip_group_sides(raft_hip_ctx *c, int32_t n_reads_total, int64_t n_rec, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe,
                         const int32_t *d_tid, const int32_t *d_ts, const int32_t *d_te, int32_t symmetric, raft_hip_slice *out)
{
    if (!c || !out || n_reads_total < 0 || n_rec < 0) return RAFT_HIP_ERR_PARAM;
    if (n_rec > 0 && (!d_qid || !d_qs || !d_qe || (!symmetric && (!d_tid || !d_ts || !d_te)))) return RAFT_HIP_ERR_PARAM;
    const long long n_ent = n_rec * (symmetric ? 1 : 2), N1 = (long long)n_reads_total + 1;
    if (n_ent >= (1LL << 31)) return RAFT_HIP_ERR_TOO_LARGE;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->gs_off_host.assign((size_t)N1, 0);
    long long n_valid = 0;
    if (n_ent > 0) {
        HIP_TRY(c, c->gs_rid.ensure((size_t)n_ent * 4));
        HIP_TRY(c, c->gs_s.ensure((size_t)n_ent * 4)); HIP_TRY(c, c->gs_e.ensure((size_t)n_ent * 4));
        HIP_TRY(c, c->gs_off.ensure((size_t)N1 * 8));
        HIP_TRY(c, c->gs_err.ensure(16));
        HIP_TRY(c, hipMemsetAsync(c->gs_err.p, 0, 8, st));
        HIP_TRY(c, hipMemsetAsync(c->gs_err.as<char>() + 8, 0xff, 8, st));
        int32_t *gerr = c->gs_err.as<int32_t>();
        long long *gerr_index = reinterpret_cast<long long *>(c->gs_err.as<char>() + 8);
        {
            const int prc = sort_sides(c, st, (long long)n_rec, n_reads_total, symmetric ? 1 : 0, d_qid, d_qs, d_qe, d_tid, d_ts, d_te, n_ent,
                                       c->gs_rid.as<int32_t>(), c->gs_s.as<int32_t>(), c->gs_e.as<int32_t>(), c->gs_off.as<long long>(), gerr, gerr_index);
            if (prc != RAFT_HIP_OK) return prc;
        }
        HIP_TRY(c, hipGetLastError());
        long long err[2] = {0, -1};
        HIP_TRY(c, hipMemcpyAsync(c->gs_off_host.data(), c->gs_off.p, (size_t)N1 * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(err, c->gs_err.p, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if ((int32_t)err[0] & kErrReadId) {
            c->last_error = "raft_hip_group_sides: record " + std::to_string(err[1]) + " names a read outside [0, n_reads_total)";
            return RAFT_HIP_ERR_READ_ID;
        }
        n_valid = c->gs_off_host[(size_t)n_reads_total];
    }
    *out = raft_hip_slice{n_valid, 1, reinterpret_cast<const int64_t *>(c->gs_off_host.data()), c->gs_s.as<int32_t>(), c->gs_e.as<int32_t>(), nullptr};
    return RAFT_HIP_OK;
}

namespace {
struct FirstRecord { int32_t v[6]; };
// hit: a record other than record 0 itself that is record 0 with query and target swapped (chop.hpp:171-184)
__global__ __launch_bounds__(256) void mirror_search_kernel(long long n_rec, long long first_index, FirstRecord f, const int32_t *qid, const int32_t *qs,
                                                            const int32_t *qe, const int32_t *tid, const int32_t *ts, const int32_t *te, int32_t *found)
{
    bool hit = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += (long long)gridDim.x * blockDim.x)
        hit |= i != first_index && qid[i] == f.v[3] && tid[i] == f.v[0] && ts[i] == f.v[1] && te[i] == f.v[2] && qs[i] == f.v[4] && qe[i] == f.v[5];
    if (__ballot(hit) != 0ull && (threadIdx.x & 63) == 0) atomicOr(found, 1);
}
bool records_ok(const raft_hip_records &r)
{
    return r.n_rec >= 0 && (r.n_rec == 0 || (r.d_qid && r.d_qs && r.d_qe && r.d_tid && r.d_ts && r.d_te));
}
// the search of one rank's slice, queued on its context's stream; the flag lands in the context's 16-byte error word
int queue_mirror_search(raft_hip_ctx *c, const raft_hip_records &r, const FirstRecord &f, bool holds_first)
{
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->gs_err.ensure(16));
    HIP_TRY(c, hipMemsetAsync(c->gs_err.p, 0, 16, c->stream));
    if (r.n_rec > 0)
        hipLaunchKernelGGL(mirror_search_kernel, dim3((unsigned)std::min<long long>((r.n_rec + 255) / 256, 256 * 16)), dim3(256), 0, c->stream, (long long)r.n_rec,
                           holds_first ? 0LL : -1LL, f, r.d_qid, r.d_qs, r.d_qe, r.d_tid, r.d_ts, r.d_te, c->gs_err.as<int32_t>());
    HIP_TRY(c, hipGetLastError());
    return RAFT_HIP_OK;
}
int read_first_record(raft_hip_ctx *c, const raft_hip_records &r, FirstRecord *f)
{
    HIP_TRY(c, hipSetDevice(c->device));
    const int32_t *col[6] = {r.d_qid, r.d_qs, r.d_qe, r.d_tid, r.d_ts, r.d_te};
    for (int k = 0; k < 6; ++k) HIP_TRY(c, hipMemcpyAsync(&f->v[k], col[k], 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RAFT_HIP_OK;
}
} // namespace

int raft_hip_presplit_symmetric_local(raft_hip_ctx *const *ctxs, int32_t world, const raft_hip_records *slices, int32_t *symmetric)
{
    if (!ctxs || world < 1 || !slices || !symmetric) return RAFT_HIP_ERR_PARAM;
    for (int p = 0; p < world; ++p) if (!ctxs[p] || !records_ok(slices[p])) return RAFT_HIP_ERR_PARAM;
    *symmetric = 0;
    if (slices[0].n_rec == 0) return RAFT_HIP_OK;          // (record 0 is rank 0's first record: without it nothing can mirror it)
    FirstRecord f{};
    { const int rc = read_first_record(ctxs[0], slices[0], &f); if (rc != RAFT_HIP_OK) return rc; }
    for (int p = 0; p < world; ++p) { const int rc = queue_mirror_search(ctxs[p], slices[p], f, p == 0); if (rc != RAFT_HIP_OK) return rc; }
    for (int p = 0; p < world; ++p) {
        raft_hip_ctx *c = ctxs[p];
        int32_t found = 0;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipMemcpyAsync(&found, c->gs_err.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (found) *symmetric = 1;
    }
    return RAFT_HIP_OK;
}

int raft_hip_presplit_symmetric(raft_hip_ctx *c, void *comm_v, int32_t rank, int32_t world, const raft_hip_records *mine, int32_t *symmetric)
{
    if (!c || !mine || !symmetric || world < 1 || rank < 0 || rank >= world || (world > 1 && !comm_v)) return RAFT_HIP_ERR_PARAM;
    RcclApi *r = world > 1 || comm_v ? rccl_api() : nullptr;
    if ((world > 1 || comm_v) && !r) { c->last_error = "librccl.so.1 could not be loaded"; return RAFT_HIP_ERR_DEVICE; }
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(comm_v);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // Every rank reaches both collectives whatever it finds wrong with its own arguments: a rank with bad arguments says so in
    // its row, and all ranks return the same error once the rows are in.
    const bool ok_mine = records_ok(*mine);
    constexpr size_t kRow = 8;                              // per rank: six columns of its first record, "has records", "arguments fine"
    std::vector<long long> rows(kRow * (size_t)world, 0);
    long long *my = rows.data() + kRow * (size_t)rank;
    my[7] = ok_mine ? 1 : 0;
    if (ok_mine && mine->n_rec > 0 && rank == 0) {
        FirstRecord f{};
        const int rc = read_first_record(c, *mine, &f);
        if (rc != RAFT_HIP_OK) my[7] = 0;
        else { for (int k = 0; k < 6; ++k) my[k] = f.v[k]; my[6] = 1; }
    }
    HIP_TRY(c, c->x_cnt.ensure(std::max<size_t>(rows.size() * 8, 64)));
    auto gather = [&](size_t words) -> int {                // this rank's `words` of rows[] to everybody, everybody's back to the host
        if (!comm_v) return RAFT_HIP_OK;
        HIP_TRY(c, hipMemcpyAsync(c->x_cnt.as<long long>() + words * (size_t)rank, rows.data() + words * (size_t)rank, words * 8, hipMemcpyHostToDevice, st));
        const ncclResult_t e = r->AllGather(c->x_cnt.as<long long>() + words * (size_t)rank, c->x_cnt.p, words, ncclInt64, comm, st);
        if (e != ncclSuccess) { c->last_error = std::string("ncclAllGather: ") + r->GetErrorString(e); return RAFT_HIP_ERR_DEVICE; }
        HIP_TRY(c, hipMemcpyAsync(rows.data(), c->x_cnt.p, words * (size_t)world * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        return RAFT_HIP_OK;
    };
    { const int rc = gather(kRow); if (rc != RAFT_HIP_OK) return rc; }
    bool all_ok = true;
    for (int p = 0; p < world; ++p) all_ok = all_ok && rows[kRow * (size_t)p + 7] == 1;
    FirstRecord f{};
    const bool have_first = all_ok && rows[6] == 1;
    for (int k = 0; k < 6; ++k) f.v[k] = (int32_t)rows[(size_t)k];
    // second round: one word per rank
    std::vector<long long> flags((size_t)world, 0);
    if (have_first) {
        // (whatever fails here is announced in the second round, not returned: the other ranks are on their way into that collective)
        int32_t found = 0;
        const bool ok = queue_mirror_search(c, *mine, f, rank == 0) == RAFT_HIP_OK &&
                        hipMemcpyAsync(&found, c->gs_err.p, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        flags[(size_t)rank] = ok ? (found ? 1 : 0) : -1;
    }
    rows.assign((size_t)world, 0);
    rows[(size_t)rank] = flags[(size_t)rank];
    { const int rc = gather(1); if (rc != RAFT_HIP_OK) return rc; }
    if (!all_ok) { c->last_error = "raft_hip_presplit_symmetric: a rank was handed columns that do not fit its record count"; return RAFT_HIP_ERR_PARAM; }
    *symmetric = 0;
    for (int p = 0; p < world; ++p) {
        if (rows[(size_t)p] < 0) { c->last_error = "raft_hip_presplit_symmetric: the search failed on rank " + std::to_string(p); return RAFT_HIP_ERR_DEVICE; }
        if (rows[(size_t)p] > 0) *symmetric = 1;
    }
    return RAFT_HIP_OK;
}

int raft_hip_comm_unique_id(void *id128)
{
    RcclApi *r = rccl_api();
    if (!r || !id128) return RAFT_HIP_ERR_DEVICE;
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    return r->GetUniqueId(reinterpret_cast<ncclUniqueId *>(id128)) == ncclSuccess ? RAFT_HIP_OK : RAFT_HIP_ERR_DEVICE;
}

int raft_hip_comm_create(int device_id, const void *id128, int32_t rank, int32_t world, void **comm)
{
    RcclApi *r = rccl_api();
    if (!r || !id128 || !comm || world < 1 || rank < 0 || rank >= world) return RAFT_HIP_ERR_PARAM;
    if (hipSetDevice(device_id) != hipSuccess) return RAFT_HIP_ERR_DEVICE;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    if (r->CommInitRank(&c, world, id, rank) != ncclSuccess) return RAFT_HIP_ERR_DEVICE;
    *comm = c;
    return RAFT_HIP_OK;
}

void raft_hip_comm_destroy(void *comm)
{
    RcclApi *r = rccl_api();
    if (r && comm) (void)r->CommDestroy(reinterpret_cast<ncclComm_t>(comm));
}

int raft_hip_exchange_local(raft_hip_ctx *const *ctxs, int32_t world, int32_t n_reads_total, const int64_t *bounds,
                            const raft_hip_slice *slices, raft_hip_received *outs)
{
    if (!ctxs || world < 1 || !bounds || !slices || !outs) return RAFT_HIP_ERR_PARAM;
    const long long N1 = (long long)n_reads_total + 1;
    for (int p = 0; p < world; ++p) {
        if (!ctxs[p] || !slice_ok(slices[p], n_reads_total)) return RAFT_HIP_ERR_PARAM;
        if (bounds[p] < 0 || bounds[p] > bounds[p + 1] || bounds[p + 1] > n_reads_total) return RAFT_HIP_ERR_PARAM;
    }
    if (bounds[0] != 0 || bounds[world] != n_reads_total) return RAFT_HIP_ERR_PARAM;
    const bool one_col = slices[0].d_qe == nullptr;       // window records: one column travels
    for (int p = 1; p < world; ++p) if ((slices[p].d_qe == nullptr) != one_col && slices[p].n_rec > 0 && slices[0].n_rec > 0) return RAFT_HIP_ERR_PARAM;
    for (int g = 0; g < world; ++g) {
        raft_hip_ctx *c = ctxs[g];
        const long long b0 = bounds[g], b1 = bounds[g + 1], n1 = b1 - b0 + 1;
        std::vector<XRun> runs;
        long long n_rec = 0;
        for (int p = 0; p < world; ++p)
            for (int j = 0; j < slices[p].n_runs; ++j) {
                const long long lo = slices[p].rec_offset[j * N1 + b0], hi = slices[p].rec_offset[j * N1 + b1];
                if (lo < 0 || hi < lo || hi > slices[p].n_rec) return RAFT_HIP_ERR_PARAM;       // (offsets that leave the slice)
                if (hi > lo) { runs.push_back(XRun{p, j, lo, hi - lo}); n_rec += hi - lo; }
            }
        if ((int)runs.size() > kMaxRuns) { c->last_error = "raft_hip_exchange: more than 16 runs arrive at one rank"; return RAFT_HIP_ERR_TOO_LARGE; }
        const int K = std::max<int>(1, (int)runs.size());
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, c->x_qs.ensure((size_t)std::max(n_rec, 1LL) * 4));
        if (!one_col) HIP_TRY(c, c->x_qe.ensure((size_t)std::max(n_rec, 1LL) * 4));
        HIP_TRY(c, c->x_off.ensure((size_t)K * (size_t)n1 * 8));
        std::vector<long long> off((size_t)K * (size_t)n1, 0);
        long long base = 0;
        for (size_t k = 0; k < runs.size(); ++k) {
            const XRun &x = runs[k];
            const int64_t *src = slices[x.peer].rec_offset + x.run * N1 + b0;
            for (long long r = 0; r < n1; ++r) off[k * (size_t)n1 + (size_t)r] = base + (src[r] - src[0]);
            const int pd = ctxs[x.peer]->device;
            if (pd == c->device) {
                HIP_TRY(c, hipMemcpyAsync(c->x_qs.as<int32_t>() + base, slices[x.peer].d_qs + x.lo, (size_t)x.n * 4, hipMemcpyDeviceToDevice, c->stream));
                if (!one_col) HIP_TRY(c, hipMemcpyAsync(c->x_qe.as<int32_t>() + base, slices[x.peer].d_qe + x.lo, (size_t)x.n * 4, hipMemcpyDeviceToDevice, c->stream));
            } else {
                HIP_TRY(c, hipMemcpyPeerAsync(c->x_qs.as<int32_t>() + base, c->device, slices[x.peer].d_qs + x.lo, pd, (size_t)x.n * 4, c->stream));
                if (!one_col) HIP_TRY(c, hipMemcpyPeerAsync(c->x_qe.as<int32_t>() + base, c->device, slices[x.peer].d_qe + x.lo, pd, (size_t)x.n * 4, c->stream));
            }
            base += x.n;
        }
        if (runs.empty()) for (long long r = 0; r < n1; ++r) off[(size_t)r] = 0;
        HIP_TRY(c, hipMemcpyAsync(c->x_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));                         // (`off` leaves scope; the peers' columns may be reused)
        outs[g] = raft_hip_received{(int32_t)(b1 - b0), K, n_rec, c->x_off.as<int64_t>(), c->x_qs.as<int32_t>(), one_col ? nullptr : c->x_qe.as<int32_t>()};
    }
    return RAFT_HIP_OK;
}

int raft_hip_exchange(raft_hip_ctx *c, void *comm_v, int32_t rank, int32_t world, int32_t n_reads_total, const int64_t *bounds,
                      const raft_hip_slice *mine, raft_hip_received *out)
{
    RcclApi *r = rccl_api();
    if (!c || !comm_v || !bounds || !mine || !out || world < 1 || rank < 0 || rank >= world) return RAFT_HIP_ERR_PARAM;
    if (!r) { c->last_error = "librccl.so.1 could not be loaded"; return RAFT_HIP_ERR_DEVICE; }
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(comm_v);
    const long long N1 = (long long)n_reads_total + 1;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // ---- Every rank must reach every collective: a rank that returned on its own would leave its peers waiting in theirs for
    // ever.  So NOTHING a single rank finds wrong on its side ends the call before the rows are gathered -- arguments that do not
    // fit (kBadRow), a device allocation or copy that failed (kNoMemRow) travel in the rank's row, and ALL ranks return the same
    // error once the matrix is in, before any send or receive is posted.  The row also carries what the rank's receive buffers
    // hold at the moment: every rank can then tell whether ANY rank has to grow a buffer for what is about to arrive, and only
    // in that case a second, one-word gather ("my buffers are ready" / "they are not") follows -- a rank whose allocation fails
    // there is announced the same way.  (What is left on this side of the first gather is its own 8 * world^2 * 5 bytes on the
    // device -- 2.5 KB for 8 ranks, made at the context's first exchange.)
    constexpr long long kBadRow = -2, kNoMemRow = -3;
    const size_t row = (size_t)world * kMaxSeg + 2;       // piece sizes per (destination, run); records / offset entries the receive buffers hold
    std::vector<long long> cnt(row * (size_t)world, 0);
    long long *my = cnt.data() + (size_t)rank * row;
    bool mine_ok = slice_ok(*mine, n_reads_total) && bounds[0] == 0 && bounds[world] == n_reads_total;
    for (int g = 0; g < world && mine_ok; ++g) {
        if (bounds[g] < 0 || bounds[g] > bounds[g + 1] || bounds[g + 1] > n_reads_total) { mine_ok = false; break; }
        for (int j = 0; j < kMaxSeg; ++j) {
            long long n = -1;                                                // (-1: the slice has no such run)
            if (j < mine->n_runs) {
                const long long lo = mine->rec_offset[j * N1 + bounds[g]], hi = mine->rec_offset[j * N1 + bounds[g + 1]];
                if (lo < 0 || hi < lo || hi > mine->n_rec) { mine_ok = false; break; }
                n = hi - lo;
            }
            my[(size_t)g * kMaxSeg + (size_t)j] = n;
        }
    }
    const bool one_col = mine->d_qe == nullptr;           // window records: one column travels (the same on every rank: the caller's protocol)
    my[row - 2] = (long long)std::min(c->x_qs.cap, one_col ? c->x_qs.cap : c->x_qe.cap) / 4;
    my[row - 1] = (long long)std::min(c->x_off.cap, c->x_raw.cap) / 8;
    // this rank's offsets on the device, from where their slices are sent: the caller's copy, or uploaded here
    const long long *d_send_off = mine_ok ? reinterpret_cast<const long long *>(mine->d_rec_offset) : nullptr;
    hipError_t my_err = hipSuccess;
    if (mine_ok && !d_send_off) {
        my_err = c->x_send_off.ensure((size_t)mine->n_runs * (size_t)N1 * 8);
        if (my_err == hipSuccess) my_err = hipMemcpyAsync(c->x_send_off.p, mine->rec_offset, (size_t)mine->n_runs * (size_t)N1 * 8, hipMemcpyHostToDevice, st);
        d_send_off = c->x_send_off.as<long long>();
    }
    if (!mine_ok) for (size_t i = 0; i < row; ++i) my[i] = kBadRow;
    else if (my_err != hipSuccess) { (void)hipGetLastError(); for (size_t i = 0; i < row; ++i) my[i] = kNoMemRow; }
    HIP_TRY(c, c->x_cnt.ensure(cnt.size() * 8));                             // (the one allocation ahead of the first gather: see above)
    auto nccl_fail = [&](ncclResult_t e, const char *what) { c->last_error = std::string(what) + ": " + r->GetErrorString(e); return RAFT_HIP_ERR_DEVICE; };
    {
        // (a copy that fails here leaves the gather to send whatever the buffer holds -- possible only with a broken device, which
        // the stream's synchronize below reports on this rank; the collective itself is still entered)
        const hipError_t e1 = hipMemcpyAsync(c->x_cnt.as<long long>() + (size_t)rank * row, my, row * 8, hipMemcpyHostToDevice, st);
        const ncclResult_t ge = r->AllGather(c->x_cnt.as<long long>() + (size_t)rank * row, c->x_cnt.p, row, ncclInt64, comm, st);
        if (ge != ncclSuccess) return nccl_fail(ge, "ncclAllGather(piece sizes)");
        if (e1 != hipSuccess) return fail_hip(c, e1, "hipMemcpyAsync(piece sizes)");
        HIP_TRY(c, hipMemcpyAsync(cnt.data(), c->x_cnt.p, cnt.size() * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
    }
    // ---- the same verdict on every rank
    for (int p = 0; p < world; ++p) {
        const long long v = cnt[(size_t)p * row];
        if (v == kBadRow) {
            c->last_error = "raft_hip_exchange: rank " + std::to_string(p) + " was handed bounds or offsets that do not fit its slice";
            return RAFT_HIP_ERR_PARAM;
        }
        if (v == kNoMemRow) {
            c->last_error = "raft_hip_exchange: rank " + std::to_string(p) + " could not stage its offsets on its device";
            return RAFT_HIP_ERR_NOMEM;
        }
    }
    bool any_grows = false;
    for (int g = 0; g < world; ++g) {
        int arriving = 0;
        long long n_in = 0;
        for (int p = 0; p < world; ++p)
            for (int j = 0; j < kMaxSeg; ++j) {
                const long long n = cnt[(size_t)p * row + (size_t)g * kMaxSeg + (size_t)j];
                if (n > 0) { ++arriving; n_in += n; }
            }
        if (arriving > kMaxRuns) {
            c->last_error = "raft_hip_exchange: more than 16 runs arrive at rank " + std::to_string(g);
            return RAFT_HIP_ERR_TOO_LARGE;
        }
        const long long n1g = bounds[g + 1] - bounds[g] + 1;
        any_grows = any_grows || std::max(n_in, 1LL) > cnt[(size_t)g * row + row - 2] || (long long)std::max(arriving, 1) * n1g > cnt[(size_t)g * row + row - 1];
    }
    // ---- what arrives here: one run per (peer, run) with records for this rank
    const long long b0 = bounds[rank], n1 = bounds[rank + 1] - b0 + 1;
    std::vector<XRun> runs;
    long long n_rec = 0;
    for (int p = 0; p < world; ++p)
        for (int j = 0; j < kMaxSeg; ++j) {
            const long long n = cnt[(size_t)p * row + (size_t)rank * kMaxSeg + (size_t)j];
            if (n > 0) { runs.push_back(XRun{p, j, 0, n}); n_rec += n; }
        }
    const int K = std::max<int>(1, (int)runs.size());
    {
        hipError_t ea = c->x_qs.ensure((size_t)std::max(n_rec, 1LL) * 4);
        if (ea == hipSuccess && !one_col) ea = c->x_qe.ensure((size_t)std::max(n_rec, 1LL) * 4);
        if (ea == hipSuccess) ea = c->x_off.ensure((size_t)K * (size_t)n1 * 8);
        if (ea == hipSuccess) ea = c->x_raw.ensure((size_t)K * (size_t)n1 * 8);
        if (ea != hipSuccess) (void)hipGetLastError();
        if (any_grows) {                                  // (every rank computed the same `any_grows` from the same matrix)
            std::vector<long long> ready((size_t)world, 0);
            ready[(size_t)rank] = ea == hipSuccess ? 1 : 0;
            const hipError_t e1 = hipMemcpyAsync(c->x_cnt.as<long long>() + rank, &ready[(size_t)rank], 8, hipMemcpyHostToDevice, st);
            const ncclResult_t ge = r->AllGather(c->x_cnt.as<long long>() + rank, c->x_cnt.p, 1, ncclInt64, comm, st);
            if (ge != ncclSuccess) return nccl_fail(ge, "ncclAllGather(buffers ready)");
            if (e1 != hipSuccess) return fail_hip(c, e1, "hipMemcpyAsync(buffers ready)");
            HIP_TRY(c, hipMemcpyAsync(ready.data(), c->x_cnt.p, (size_t)world * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            for (int p = 0; p < world; ++p)
                if (ready[(size_t)p] != 1) {
                    c->last_error = "raft_hip_exchange: rank " + std::to_string(p) + " has no device memory for what it is about to receive";
                    return RAFT_HIP_ERR_NOMEM;
                }
        } else if (ea != hipSuccess) return fail_hip(c, ea, "raft_hip_exchange: receive buffers");   // (cannot happen: nothing had to grow)
    }
    RunBases rb{};
    {
        long long base = 0;
        for (size_t k = 0; k < runs.size(); ++k) { rb.base[k] = base; base += runs[k].n; }
    }
    // ---- the exchange: per ordered pair of ranks the sends and the receives are issued in the same order (run by run:
    // qs, qe, offsets), all inside one group -- xGMI is point-to-point, every pair has its own link
    {
        const ncclResult_t gs = r->GroupStart();
        if (gs != ncclSuccess) return nccl_fail(gs, "ncclGroupStart");
        // (a failed post must not leave the group open: the first error is kept, the group is closed, then the call returns)
        ncclResult_t first = ncclSuccess;
        const char *what = "";
        auto post = [&](ncclResult_t e, const char *w) { if (e != ncclSuccess && first == ncclSuccess) { first = e; what = w; } return first == ncclSuccess; };
        for (int g = 0; g < world && first == ncclSuccess; ++g)
            for (int j = 0; j < mine->n_runs && first == ncclSuccess; ++j) {
                const long long lo = mine->rec_offset[j * N1 + bounds[g]], n = mine->rec_offset[j * N1 + bounds[g + 1]] - lo;
                if (n <= 0) continue;
                if (!post(r->Send(mine->d_qs + lo, (size_t)n, ncclInt32, g, comm, st), "ncclSend(qs)")) break;
                if (!one_col && !post(r->Send(mine->d_qe + lo, (size_t)n, ncclInt32, g, comm, st), "ncclSend(qe)")) break;
                post(r->Send(d_send_off + j * N1 + bounds[g], (size_t)(bounds[g + 1] - bounds[g] + 1), ncclInt64, g, comm, st), "ncclSend(offsets)");
            }
        for (size_t k = 0; k < runs.size() && first == ncclSuccess; ++k) {
            if (!post(r->Recv(c->x_qs.as<int32_t>() + rb.base[k], (size_t)runs[k].n, ncclInt32, runs[k].peer, comm, st), "ncclRecv(qs)")) break;
            if (!one_col && !post(r->Recv(c->x_qe.as<int32_t>() + rb.base[k], (size_t)runs[k].n, ncclInt32, runs[k].peer, comm, st), "ncclRecv(qe)")) break;
            post(r->Recv(c->x_raw.as<long long>() + (long long)k * n1, (size_t)n1, ncclInt64, runs[k].peer, comm, st), "ncclRecv(offsets)");
        }
        const ncclResult_t ge = r->GroupEnd();
        if (first != ncclSuccess) return nccl_fail(first, what);
        if (ge != ncclSuccess) return nccl_fail(ge, "ncclGroupEnd");
    }
    if (runs.empty()) HIP_TRY(c, hipMemsetAsync(c->x_off.p, 0, (size_t)n1 * 8, st));
    else
        hipLaunchKernelGGL(rebase_offsets_kernel, dim3((unsigned)((n1 * K + 255) / 256)), dim3(256), 0, st, K, n1, c->x_raw.as<long long>(), rb,
                           c->x_off.as<long long>());
    HIP_TRY(c, hipGetLastError());
    *out = raft_hip_received{(int32_t)(n1 - 1), K, n_rec, c->x_off.as<int64_t>(), c->x_qs.as<int32_t>(), one_col ? nullptr : c->x_qe.as<int32_t>()};
    return RAFT_HIP_OK;                                      // (in stream order: a pass on this context's stream may follow at once)
}

int raft_hip_last_timing(raft_hip_ctx *c, double *pileup_seconds, double *pass_seconds)
{
    if (!c) return RAFT_HIP_ERR_PARAM;
    if (!c->finished) return RAFT_HIP_ERR_STATE;
    HIP_TRY(c, hipEventSynchronize(c->ev_pass1));          // (finish may have seen the pass's number before the runtime saw its last event)
    float ms = 0.f;
    if (pileup_seconds) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pile0, c->ev_pile1)); *pileup_seconds = ms * 1e-3; }
    if (pass_seconds) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pass0, c->ev_pass1)); *pass_seconds = ms * 1e-3; }
    return RAFT_HIP_OK;
}

int raft_hip_selftest(int device_id)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return RAFT_HIP_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return RAFT_HIP_ERR_DEVICE;
    const int n = 256;
    int h_in[n], h_a[n], h_b[n];
    unsigned long long h_bal[n / 64];
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h_in[i] = (int)(s >> 20) - 2048; }
    int *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    unsigned long long *d_bal = nullptr;
    int rc = RAFT_HIP_ERR_DEVICE;
    if (hipMalloc(&d_in, sizeof h_in) == hipSuccess && hipMalloc(&d_a, sizeof h_a) == hipSuccess &&
        hipMalloc(&d_b, sizeof h_b) == hipSuccess && hipMalloc(&d_bal, sizeof h_bal) == hipSuccess &&
        hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(n), 0, 0, d_in, d_a, d_b, d_bal);
        if (hipMemcpy(h_a, d_a, sizeof h_a, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(h_b, d_b, sizeof h_b, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(h_bal, d_bal, sizeof h_bal, hipMemcpyDeviceToHost) == hipSuccess) {
            rc = RAFT_HIP_OK;
            for (int w = 0; w < n / 64; ++w) {
                int run = 0;
                unsigned long long bal = 0;
                for (int l = 0; l < 64; ++l) {
                    run += h_in[w * 64 + l];
                    if (h_in[w * 64 + l] & 1) bal |= 1ull << l;
                    if (h_a[w * 64 + l] != run || h_b[w * 64 + l] != run) rc = 100 + w;
                }
                if (bal != h_bal[w]) rc = 200 + w;
            }
        }
    }
    (void)hipFree(d_in); (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_bal);
    return rc;
}

} // extern "C"

// ---- the pre-split job of ONE process (BASELINE configs[3] behind the CLI: main.cpp:21-87 + chop.hpp:331-373 with the record stream
// cut into `world` contiguous slices, one per rank; ranks are contexts of this process, on as many devices as the caller made them
// on).  Every step is the native one: the slices go up as they are; raft_hip_presplit_symmetric_local finds the flag
// (chop.hpp:171-184); raft_hip_group_sides expands the sides each slice piles up (chop.hpp:165-169) and groups them by read;
// raft_hip_exchange_local routes every interval to the rank that owns its read (contiguous read ranges of equal window counts);
// each rank runs the grouped pass on what arrived and its share of the outputs lands in the caller's arrays, in read order --
// the fragment numbering (chop.hpp:195 read_num) and the stdout sums (repeat.hpp:93-97) are global because the CSR arrays are.
int raft_hip_run_presplit_local(raft_hip_ctx *const *ctxs, int32_t world, int32_t n_reads, const int32_t *read_len, int64_t n_rec,
                                const int32_t *qid, const int32_t *qs, const int32_t *qe, const int32_t *tid, const int32_t *ts, const int32_t *te,
                                raft_hip_host_outputs *out, raft_hip_summary *summary)
{
    if (!ctxs || world < 1 || world > 64 || n_reads < 0 || n_rec < 0 || !out) return RAFT_HIP_ERR_PARAM;
    for (int r = 0; r < world; ++r) if (!ctxs[r]) return RAFT_HIP_ERR_PARAM;
    if (n_reads > 0 && !read_len) return RAFT_HIP_ERR_PARAM;
    if (n_rec > 0 && (!qid || !qs || !qe || !tid || !ts || !te)) return RAFT_HIP_ERR_PARAM;
    const int width = out->cov_width == 2 ? 2 : 1;
    if (out->cov_width != 0 && out->cov_width != 1 && out->cov_width != 2) return RAFT_HIP_ERR_PARAM;   // (four-bit steps: chunks would have to begin on multiples of four windows)
    if (!out->cov_offset || !out->cov8 || !out->rep_offset || !out->rep_s || !out->rep_e || !out->frag_offset || !out->frag_begin || !out->frag_end)
        return RAFT_HIP_ERR_PARAM;
    raft_hip_ctx *c0 = ctxs[0];
    const int reso = c0->prm.reso;
    // read ranges of (nearly) equal window counts: what every rank can compute from the read lengths alone
    std::vector<int64_t> win_off((size_t)n_reads + 1, 0), bounds((size_t)world + 1, 0);
    for (int32_t i = 0; i < n_reads; ++i) {
        if (read_len[i] < 0) { if (summary) { memset(summary, 0, sizeof *summary); summary->error_index = i; } return RAFT_HIP_ERR_PARAM; }
        win_off[(size_t)i + 1] = win_off[(size_t)i] + ((int64_t)read_len[i] + reso - 1) / reso;
    }
    const int64_t W = win_off[(size_t)n_reads];
    if (W > out->cov8_cap) return RAFT_HIP_ERR_TOO_LARGE;
    for (int g = 1; g < world; ++g)
        bounds[(size_t)g] = std::lower_bound(win_off.begin(), win_off.end(), (int64_t)((__int128)W * g / world)) - win_off.begin();
    bounds[(size_t)world] = n_reads;
    for (int g = 1; g <= world; ++g) bounds[(size_t)g] = std::min<int64_t>(std::max(bounds[(size_t)g], bounds[(size_t)g - 1]), n_reads);

    std::vector<int> rcs((size_t)world, RAFT_HIP_OK);
    auto each_rank = [&](const std::function<int(int)> &f) -> int {
        std::vector<std::thread> th;
        for (int r = 1; r < world; ++r) th.emplace_back([&, r] { rcs[(size_t)r] = f(r); });
        rcs[0] = f(0);
        for (auto &t : th) t.join();
        for (int r = 0; r < world; ++r) if (rcs[(size_t)r] != RAFT_HIP_OK) { if (r) c0->last_error = "rank " + std::to_string(r) + ": " + ctxs[r]->last_error; return rcs[(size_t)r]; }
        return RAFT_HIP_OK;
    };
    // 1. every rank's slice of the six columns, on its device
    std::vector<raft_hip_records> recs((size_t)world);
    const int32_t *src[6] = {qid, qs, qe, tid, ts, te};
    int rc = each_rank([&](int r) -> int {
        raft_hip_ctx *c = ctxs[r];
        const int64_t lo = n_rec * r / world, hi = n_rec * (r + 1) / world, n = hi - lo;
        HIP_TRY(c, hipSetDevice(c->device));
        for (int k = 0; k < 6; ++k) {
            HIP_TRY(c, c->in_col[k].ensure((size_t)std::max<int64_t>(n, 1) * 4));
            if (n) HIP_TRY(c, hipMemcpyAsync(c->in_col[k].p, src[k] + lo, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        recs[(size_t)r] = raft_hip_records{n, c->in_col[0].as<int32_t>(), c->in_col[1].as<int32_t>(), c->in_col[2].as<int32_t>(),
                                           c->in_col[3].as<int32_t>(), c->in_col[4].as<int32_t>(), c->in_col[5].as<int32_t>()};
        return RAFT_HIP_OK;
    });
    if (rc != RAFT_HIP_OK) return rc;
    // 2. the symmetric flag of the whole stream
    int32_t sym = 0;
    rc = raft_hip_presplit_symmetric_local(ctxs, world, recs.data(), &sym);
    if (rc != RAFT_HIP_OK) return rc;
    // 3. every slice's sides, grouped by read
    std::vector<raft_hip_slice> slices((size_t)world);
    rc = each_rank([&](int r) -> int {
        const raft_hip_records &q = recs[(size_t)r];
        return raft_hip_group_sides(ctxs[r], n_reads, q.n_rec, q.d_qid, q.d_qs, q.d_qe, q.d_tid, q.d_ts, q.d_te, sym, &slices[(size_t)r]);
    });
    if (rc != RAFT_HIP_OK) {
        if (summary) { memset(summary, 0, sizeof *summary); summary->error_index = -1; }
        return rc;
    }
    // 4. ONE exchange step
    std::vector<raft_hip_received> got((size_t)world);
    rc = raft_hip_exchange_local(ctxs, world, n_reads, bounds.data(), slices.data(), got.data());
    if (rc != RAFT_HIP_OK) return rc;
    // 5. every rank's pass over what arrived for its reads
    std::vector<raft_hip_summary> sums((size_t)world);
    rc = each_rank([&](int r) -> int {
        raft_hip_ctx *c = ctxs[r];
        const int64_t b0 = bounds[(size_t)r], b1 = bounds[(size_t)r + 1];
        const int32_t nr = (int32_t)(b1 - b0);
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, c->in_len.ensure((size_t)std::max<int32_t>(nr, 1) * 4));
        if (nr) HIP_TRY(c, hipMemcpyAsync(c->in_len.p, read_len + b0, (size_t)nr * 4, hipMemcpyHostToDevice, c->stream));
        const int keep_mode = c->prm.symmetric_mode, keep_width = c->out_width;
        c->prm.symmetric_mode = 1;                         // (the sides are expanded: a grouped pass piles up what it is given)
        c->out_width = width;
        int prc = raft_hip_run_device_grouped(c, nr, c->in_len.as<int32_t>(), got[(size_t)r].n_rec, got[(size_t)r].n_runs, got[(size_t)r].d_rec_offset, nullptr,
                                              got[(size_t)r].d_qs, got[(size_t)r].d_qe, win_off[(size_t)b1] - win_off[(size_t)b0]);
        if (prc == RAFT_HIP_OK) prc = raft_hip_finish(c, &sums[(size_t)r]);
        c->prm.symmetric_mode = keep_mode; c->out_width = keep_width;
        return prc;
    });
    if (rc != RAFT_HIP_OK) {
        if (summary) {
            memset(summary, 0, sizeof *summary); summary->error_index = -1;
            for (int r = 0; r < world; ++r) if (rcs[(size_t)r] != RAFT_HIP_OK) { *summary = sums[(size_t)r]; break; }
        }
        return rc;
    }
    // 6. the ranks' shares, in read order
    int64_t n_exc = 0, rep_at = 0, frag_at = 0;
    bool overflow = false;
    for (int r = 0; r < world && rc == RAFT_HIP_OK; ++r) {
        raft_hip_ctx *c = ctxs[r];
        const int64_t b0 = bounds[(size_t)r], b1 = bounds[(size_t)r + 1], w0 = win_off[(size_t)b0];
        const raft_hip_summary &sr = sums[(size_t)r];
        if (rep_at + sr.n_repeats > out->rep_cap || frag_at + sr.n_fragments > out->frag_cap) { rc = RAFT_HIP_ERR_TOO_LARGE; break; }
        int64_t ne = 0;
        const int64_t room = std::max<int64_t>(out->exc_cap - n_exc, 0);
        int frc = overflow ? raft_hip_fetch_packed_w(c, width, nullptr, nullptr, 0, nullptr, nullptr, &ne, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)
                           : raft_hip_fetch_packed_w(c, width, out->cov_offset + b0, out->cov8 + (size_t)w0 * (size_t)width, room, out->exc_index ? out->exc_index + n_exc : nullptr,
                                                     out->exc_value ? out->exc_value + n_exc : nullptr, &ne, out->rep_offset + b0, out->rep_s + rep_at, out->rep_e + rep_at,
                                                     out->frag_offset + b0, nullptr, out->frag_begin + frag_at, out->frag_end + frag_at);
        if (frc == RAFT_HIP_ERR_TOO_LARGE && !overflow) { overflow = true; frc = RAFT_HIP_OK; }   // (the later ranks only say how many they have)
        if (frc != RAFT_HIP_OK) { rc = frc; break; }
        if (!overflow) {
            for (int64_t i = b0; i <= b1; ++i) { out->cov_offset[i] += w0; out->rep_offset[i] += rep_at; out->frag_offset[i] += frag_at; }
            if (out->exc_index) for (int64_t i = 0; i < ne; ++i) out->exc_index[n_exc + i] += w0;
        }
        n_exc += ne; rep_at += sr.n_repeats; frag_at += sr.n_fragments;
    }
    out->n_exc = n_exc;
    if (rc == RAFT_HIP_OK && overflow) rc = RAFT_HIP_ERR_TOO_LARGE;
    if (summary) {
        raft_hip_summary t{};
        t.n_reads = n_reads; t.symmetric = sym; t.high_cov = c0->high_cov; t.interval_path = 1; t.n_segments = world; t.n_records = n_rec;
        t.error_index = -1; t.n_devices_used = world;
        for (int r = 0; r < world; ++r) {
            const raft_hip_summary &sr = sums[(size_t)r];
            t.n_intervals += sr.n_intervals; t.n_bins += sr.n_bins; t.n_repeats += sr.n_repeats; t.n_cuts += sr.n_cuts; t.n_fragments += sr.n_fragments;
            t.total_coverage += sr.total_coverage; t.total_windows += sr.total_windows; t.total_repeat_length += sr.total_repeat_length;
            t.total_read_length += sr.total_read_length;
        }
        *summary = t;
    }
    return rc;
}
