// wave_launch.hpp -- launch interface of pileup_wave_kernel (pileup_wave.hpp); its instantiations live in a translation unit
// of their own (wave_launch.hip) so that they compile beside engine.hip.
#pragma once
#include <hip/hip_runtime.h>

#ifndef RAFT_WAVE_SLOTS
#define RAFT_WAVE_SLOTS 4096      // 16-bit window slots of one wave's LDS array (a multiple of 512)
#endif
#ifndef RAFT_WAVE_WPS
#define RAFT_WAVE_WPS 4           // waves per SIMD asked of the register allocator (64 VGPRs at 8)
#endif
#ifndef RAFT_WAVE_WPB
#define RAFT_WAVE_WPB 1           // waves per workgroup (independent of each other: no barrier anywhere)
#endif

namespace raft {
constexpr int kWaveSlots = RAFT_WAVE_SLOTS;
// waves of the full persistent grid: every SIMD of the chip holds RAFT_WAVE_WPS of them (n_waves of a launch: at most that, a multiple of RAFT_WAVE_WPB)
int wave_grid_waves(bool win);
// ow: bytes per window written (4, 1, 2, or 8 = four-bit steps); win: window records (pileup_wave.hpp IN = 1)
// (cuts: const TileCut *, pa: const PileupArgs * -- untyped here because wave_launch.hip includes the shared kernel headers under
// a namespace of its own, so that the kernels those headers define do not exist twice in the library)
void launch_wave_variant(int ow, bool win, hipStream_t st, int n_seg, const void *cuts, const void *pa, int n_waves);
} // namespace raft
