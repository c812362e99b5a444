// engine_placement.hip -- where buffers lie: the per-device pool of physical chunks (trim, what it holds), the placement policy, the
// placement trial's report, device memory for the callers' input columns, page-locking of caller memory.  The buffers themselves
// (DevBuf: virtual ranges over pooled chunks) are in engine_ctx.hpp; DESIGN.md I.4 says why any of this exists.
#include "engine_ctx.hpp"

extern "C" {

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


} // extern "C"
