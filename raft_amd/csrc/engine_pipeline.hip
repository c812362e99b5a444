// engine_pipeline.hip -- the host-to-host entry points (include/raft_hip.h raft_hip_run_host_grouped ... raft_hip_run_pipelined):
// page-locked host columns in, every output in host memory out, with upload / pass / download of consecutive read ranges overlapped,
// over one context or several (one per GPU), and the host-routed path for streams that are not a handful of sorted runs.  Everything
// here is host code around the passes of engine.hip.
#include "engine_ctx.hpp"

extern "C" {

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
                launch_rebase_ids(jc->stream, jc->in_col[0].as<int32_t>(), n_iv, r0);
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
                launch_rebase_ids(st, l->in_col[0].as<int32_t>(), cp.n_rec, cp.r0);
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
                    if (base != 0 && n > 0) launch_add_base(st, b.as<long long>(), n, base);
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
                        if (j.dst && j.bytes) {
                            // (trace: a copy call that holds its caller.  In a process's first passes one or two of them take ~7 ms of
                            // the caller's CPU time each, whatever their size: the runtime picks another SDMA engine when the stream's
                            // last one is busy, and an engine's first use sets its queue up -- hsa_amd_memory_async_copy_on_engine ->
                            // a KFD SVM ioctl, profiles/r06_sdma_first_use.txt.  A long-lived context stops seeing them.)
                            timespec c0{}, c1{};
                            const auto w0 = std::chrono::steady_clock::now();
                            if (trace) clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
                            LANE_TRY(hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, jc->down_stream));
                            if (trace) {
                                clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
                                const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                                if (wall > 0.5)
                                    fprintf(stderr, "PIPE chunk %2d d2h copy %d of %zu bytes held its caller %.3f ms (thread cpu %.3f ms)\n", k, (int)(&j - job), j.bytes, wall,
                                            (c1.tv_sec - c0.tv_sec) * 1e3 + (c1.tv_nsec - c0.tv_nsec) * 1e-6);
                                stamp(k, "d2h copy");
                            }
                        }
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
    if (chunked) {
        // ... and the page-locked ring the lanes derive plain columns into (run_multi_impl, DeriveRing): 300 MB for a 4.4e7-record
        // job, whose page-locking was 50 of the 64 ms that job's engine call took (profiles/r06_s18_cli_s500k.txt: its two chunks
        // were through after 8 ms)
        const size_t off_b = ((size_t)kWinMaxRuns * ((size_t)nr + 1) * 8 + 255) & ~(size_t)255;
        const size_t need = ((off_b + (size_t)nrec * 4 + 255) & ~(size_t)255) * DeriveRing::R;
        if (need > c->h_stage_cap && !getenv("RAFT_NO_DERIVE")) {
            HIP_TRY(c, hipSetDevice(c->device));
            if (c->h_stage) (void)hipHostFree(c->h_stage);
            c->h_stage = nullptr; c->h_stage_cap = 0;
            HIP_TRY(c, hipHostMalloc(&c->h_stage, need, hipHostMallocDefault));
            c->h_stage_cap = need;
        }
    }
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


} // extern "C"
