// engine_exchange.hip -- pre-split PAF (BASELINE configs[3]): the symmetric flag across ranks, a slice's sides grouped
// by read, the exchange that routes every interval to the rank that owns its read (RCCL over xGMI, or peer copies between the
// contexts of one process), and the whole pre-split job of one process.
#include "engine_ctx.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

extern "C" {

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
