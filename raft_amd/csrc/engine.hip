// engine.hip -- context, launch sequence and C ABI (include/raft_hip.h) of the
// MI355X engine.  One context = one device + one stream + grow-only device
// buffers; one pass = the kernels listed in DESIGN.md §Kernels, in order.
#include "engine_ctx.hpp"

#include "bucket.hpp"
#include "sort_pairs.hpp"
#include "device_scan.hpp"
#include "finalize.hpp"
#include "pack.hpp"
#include "pileup.hpp"
#include "pileup_wave.hpp"
#include "wave_launch.hpp"
#include "pileup_deep.hpp"

namespace {

struct ReadPrepLoader {               // per read: windows, reserved repeat slots, marker capacity
    const int32_t *len;
    int32_t reso, minbins, L;
    int32_t long_windows, piece_w;    // reads longer than long_windows are piled up in pieces of piece_w windows
    int32_t *err_flags;
    long long *err_index;
    FastDiv by_reso, by_mb1, by_L;    // reso, minbins + 1, L as divisors (three hardware divisions per read, one of them 64 bits wide,
                                      // twice per pass, were most of what the two scan kernels executed)
    int32_t *seen;                    // the lengths as this scan saw them (engine_ctx.hpp len_seen)
    __device__ void operator()(long long i, long long (&v)[3]) const
    {
        int l = len[i];
        seen[i] = l;
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

void raft::launch_rebase_ids(hipStream_t st, int32_t *ids, long long n, int32_t base)
{
    hipLaunchKernelGGL(rebase_ids_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 4096))), dim3(256), 0, st, ids, n, base);
}
void raft::launch_add_base(hipStream_t st, long long *a, long long n, long long base)
{
    hipLaunchKernelGGL(add_base_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, 1024))), dim3(256), 0, st, a, n, base);
}

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
    DevBuf *all[] = {&c->len_seen, &c->deep_list, &c->tail_buf, &c->wave_ctr, &c->ctrl, &c->scan_tmp, &c->cov_off, &c->rep_res_off, &c->tile_first, &c->tile_cuts,
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
int sort_sides(raft_hip_ctx *c, hipStream_t st, long long n_rec, int32_t n_reads, int symmetric, const int32_t *d_qid, const int32_t *d_qs,
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
    int bits = 1;
    while (bits < 32 && (1LL << bits) <= (long long)n_reads) ++bits;               // keys 0 .. n_reads (the sides that do not exist)
    HIP_TRY(c, c->sort_tmp.ensure(rs_items_tmp_bytes(cap_iv)));
    bool in_b = false;
    // (the first pass makes its items from the columns: no expansion kernel, no 16 bytes per side written and read back)
    const SideSource src{(long long)n_rec, n_reads, symmetric, make_fast_div(c->prm.reso), d_qid, d_qs, d_qe, d_tid, d_ts, d_te, err_flags, err_index};
    HIP_TRY(c, radix_sort_items(st, src, c->rs_v0.as<unsigned long long>(), c->rs_v1.as<unsigned long long>(), cap_iv, bits, c->sort_tmp.p, &in_b));
    const unsigned long long *sorted = in_b ? c->rs_v1.as<unsigned long long>() : c->rs_v0.as<unsigned long long>();
    const unsigned g2 = (unsigned)std::max<long long>(1, std::min<long long>((cap_iv + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(unzip_items_kernel, dim3(g2), dim3(256), 0, st, cap_iv, n_reads, sorted, o_win, off, c->gaps.as<GapList>());
    hipLaunchKernelGGL(fill_gaps_kernel, dim3(64), dim3(256), 0, st, c->gaps.as<GapList>(), off);
    HIP_TRY(c, hipGetLastError());
    return RAFT_HIP_OK;
}

int run_pass(raft_hip_ctx *c, const raft_hip_ctx::PassArgs &in, bool verify_in_kernels)
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
    // window records go to the pileup kernel's own instantiation (pileup_wave.hpp IN = 1) where the runs are few; anything else gets
    // coordinate columns that fall into the same windows (bucket.hpp unpack_windows_kernel) and takes the paths those have
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
                           make_fast_div(c->minbins < INT32_MAX ? c->minbins + 1 : 1), make_fast_div(c->prm.interval_length), nullptr};
    HIP_TRY(c, c->len_seen.ensure((size_t)std::max(N, 1LL) * 4));
    prep_ld.seen = c->len_seen.as<int32_t>();
    ScanOut<3> prep_so{{c->cov_off.as<long long>(), c->rep_res_off.as<long long>(), nullptr}};   // (marker capacities: only their sum is used, to size the cut points' array)
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
    // (the geometry of the remembered pass, if nobody has written it since: engine_ctx.hpp geom_id.  RAFT_NO_KEEP_GEOMETRY=1: scanned again)
    const bool keep_geom = speculate && !grouped && c->geom_id != 0 && c->shape.geom_id == c->geom_id && getenv("RAFT_NO_KEEP_GEOMETRY") == nullptr;
    if (!keep_geom) ++c->geom_id;
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
    // boundary's first read, records and window).  Three tiles' worth (four until round 5: on the human-scale set the kernel likes
    // short ranges -- 2.42 / 2.45 / 2.49 / 2.56 ms at two / three / four / eight tiles' worth in one context -- and tile_desc_kernel
    // long ones; the pass is shortest at three, profiles/r05_quantum_sweep.txt).  Smaller sets keep the three tiles' worth down to
    // two ranges per worker, and two tiles' worth below that: a draw is an atomic and two boundary records a worker waits for, and
    // with five to fifteen tiles per worker those waits cost more than the kernel's last draws do (profiles/r06_quantum_small_sets.txt:
    // an eighth of the human-scale set 0.424 -> 0.410 ms per step, 200 k reads 0.289 -> 0.276, 50 k reads 0.186 -> 0.178; until
    // round 6 such sets were cut into eight ranges per worker or single tiles' worth).
    const long long q3 = 3LL * (kTileCap / 128) * 128, q1 = (kTileCap / 128) * 128;
    const int Q = c->tile_q ? std::max(256, c->tile_q) : (int)std::max(2 * q1, std::min(q3, (B / (2LL * wave_grid_waves(true))) / 128 * 128));
    // A set with many tiles per worker gets a GRADED quantum (raft_types.hpp Quantum): most of each eighth of the set in ranges of
    // eight tiles' worth, its last tenth in ranges of two -- half the boundaries tile_desc_kernel has to look up, and the kernel's end
    // waits for a short draw.  (RAFT_GRADED_QUANTUM=0: uniform, as until round 6.)
    static const bool graded_off = [] { const char *e = getenv("RAFT_GRADED_QUANTUM"); return e && atoi(e) == 0; }();
    const bool graded = !c->tile_q && !graded_off && B / kTileCap >= 64LL * wave_grid_waves(true) && (B + 7) / 8 + 128 < (1LL << 31);
    const Quantum qz = graded ? graded_quantum(B, 8 * (int)q1, 2 * (int)q1, 0.9) : uniform_quantum(Q);
    const long long n_tiles = qz.n_ranges(B);

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
        if (keep_geom) {
            // ONE launch: the lengths against the ones the geometry was made from (kErrHint), the repeat counters cleared; the run guess beside it
            const int vb = (int)((N + kVerifyReads - 1) / kVerifyReads);
            hipLaunchKernelGGL((verify_lengths_kernel<GuessBeside>), dim3((unsigned)(vb + (guess_too ? kGuessBlocks : 0))), dim3(256), 0, st, n_reads, d_len,
                               c->len_seen.as<int32_t>(), c->rep_cnt.as<int32_t>(), &ctrl->err_flags, vb, gb);
        } else {
        hipLaunchKernelGGL((scan_partials_kernel<ReadPrepLoader, 3, GuessBeside>), dim3((unsigned)(nb_scan + (guess_too ? kGuessBlocks : 0))), dim3(kScanThreads), 0, st,
                           prep_ld, N, partials, nb_scan, gb);
        PrepPost pp{n_reads, qz, n_tiles, c->tile_first.as<int32_t>(), c->rep_cnt.as<int32_t>(), &ctrl->err_flags, &ctrl->err_index, grp, eff_runs,
                    (long long)n_rec, B, RU, CU};
        hipLaunchKernelGGL((scan_apply_kernel<ReadPrepLoader, 3, true, PrepPost>), dim3((unsigned)nb_scan), dim3(kScanThreads), 0, st, prep_ld, N, partials, scan_totals,
                           prep_so, pp);
        }
    } else
    hipLaunchKernelGGL(tile_first_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, st, n_reads,
                       c->cov_off.as<long long>(), qz, n_tiles, c->tile_first.as<int32_t>(), &ctrl->err_flags, &ctrl->err_index, grp,
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
    // (a speculative pass over a stream whose last pass listed no deep tile does not launch the side kernel: 4 us of a 0.4 ms pass; a
    // tile that is deep after all finds no room in the list, and raft_hip_finish runs the pass again the long way)
    c->deep_skipped = speculate && !c->shape.had_deep && getenv("RAFT_DEEP_MIN") == nullptr;
    pa.deep_list = c->deep_list.p; pa.n_deep = &ctrl->n_deep; pa.deep_cap = c->deep_skipped ? 0 : (int32_t)std::min<long long>(c->deep_cap, INT32_MAX);
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
        c->shape.B = B; c->shape.RU = RU; c->shape.CU = CU; c->shape.n_desc = n_desc; c->shape.geom_id = c->geom_id;
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
    if (wave_launched && !c->deep_skipped)
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

int run_grouped(raft_hip_ctx *c, int32_t n_reads, const int32_t *d_len, int64_t n_rec, int32_t n_runs, const int64_t *d_rec_offset,
                const long long *adj, const int32_t *d_qid, const int32_t *d_qs, const int32_t *d_qe, int64_t n_bins,
                const uint32_t *d_win)
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

// the control block as the pass's last workgroup handed it over (publish_and_clear_kernel: stamped lines, 1024 bytes into the page-locked block)
Ctrl host_ctrl(const raft_hip_ctx *c)
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
            } else if ((hc.err_flags & kErrDeep) && ((long long)hc.n_deep > c->deep_cap || c->deep_skipped)) {
                // more deep tiles than the list held -- or a pass that was launched without the side kernel met one: once more, with room
                c->deep_cap = std::max(c->deep_cap, (long long)hc.n_deep + 64);
                c->shape.had_deep = true;
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
            c->shape.had_deep = hc.n_deep > 0;
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
int pack_coverage(raft_hip_ctx *c, int width)
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
int sort_exceptions(raft_hip_ctx *c)
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

int fetch_packed_impl(raft_hip_ctx *c, int32_t width, int64_t *cov_offset, void *cov_packed, int32_t *cov_anchor, int64_t exc_cap, int64_t *exc_index,
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

