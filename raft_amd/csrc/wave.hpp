// wave.hpp -- wavefront (64-lane) primitives for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace raft {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id()
{
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Inclusive +-scan across the 64 lanes with DPP row shifts and row broadcasts
// (the sequence LLVM's AMDGPU atomic optimizer emits for GFX9-family waves).
// Requires EXEC = all ones.
__device__ __forceinline__ int wave_incl_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2,3
    return v;
}

// Inclusive max-scan, same steps; `neutral` is what a lane without a source keeps (the identity of max for the caller's
// values).  wave_shr1: lane l takes lane l - 1's value, lane 0 keeps `first`.
__device__ __forceinline__ int wave_incl_scan_max(int v, int neutral)
{
    // (a lane without a source keeps its own value -- max(v, v) -- so that each step is ONE v_max_i32_dpp instead of a move of the
    // neutral element, a dpp move and a max)
    (void)neutral;
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false)); // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false)); // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false)); // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false)); // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false)); // row_bcast:15 into rows 1,3
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false)); // row_bcast:31 into rows 2,3
    return v;
}
__device__ __forceinline__ int wave_shr1(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);        // wave_shr:1 (GFX9 family)
}

// Same scan through ds_bpermute shuffles; used by the self test as the cross-check
// and by kernels that are not on the hot path.
__device__ __forceinline__ int wave_incl_scan_add_shfl(int v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        int o = __shfl_up(v, d, kWave);
        if (lane >= d) v += o;
    }
    return v;
}

// 64-bit inclusive scan: the same six DPP steps on (low, high) dword pairs (a shuffle version costs twelve
// ds_bpermute round trips per call)
__device__ __forceinline__ long long wave_incl_scan_add64(long long v)
{
    unsigned long long x = (unsigned long long)v;
#define RAFT_DPP64_STEP(ctrl, rows)                                                                          \
    {                                                                                                        \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)x, ctrl, rows, 0xf, false);          \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), ctrl, rows, 0xf, false);  \
        x += ((unsigned long long)hi << 32) | lo;                                                            \
    }
    RAFT_DPP64_STEP(0x111, 0xf) // row_shr:1
    RAFT_DPP64_STEP(0x112, 0xf) // row_shr:2
    RAFT_DPP64_STEP(0x114, 0xf) // row_shr:4
    RAFT_DPP64_STEP(0x118, 0xf) // row_shr:8
    RAFT_DPP64_STEP(0x142, 0xa) // row_bcast:15 into rows 1,3
    RAFT_DPP64_STEP(0x143, 0xc) // row_bcast:31 into rows 2,3
#undef RAFT_DPP64_STEP
    return (long long)x;
}

__device__ __forceinline__ long long wave_reduce_add64(long long v)
{
    const long long s = wave_incl_scan_add64(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)s, kWave - 1);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)s >> 32), kWave - 1);
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Sum over the wave, returned in a scalar register: the DPP scan's last lane.  (An xor-shuffle tree costs six
// dependent ds_bpermute round trips, ~600 cycles per call in the pileup kernels.)
__device__ __forceinline__ int wave_reduce_add(int v)
{
    return __builtin_amdgcn_readlane(wave_incl_scan_add(v), kWave - 1);
}

// Values that are the same in every lane but were fetched with vector loads (the compiler cannot prove the
// memory is not clobbered by the kernel's own stores, so it will not use scalar loads): move them to SGPRs so
// that everything computed from them -- loop bounds, branch conditions, addresses -- stays scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni(long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Position (0..63) of the highest set bit; m != 0.
__device__ __forceinline__ int top_bit(unsigned long long m) { return 63 - __clzll((long long)m); }

} // namespace raft
