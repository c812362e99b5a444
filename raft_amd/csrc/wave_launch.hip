// wave_launch.hip -- instantiations and launches of pileup_wave_kernel.
#include "wave_launch.hpp"
// The shared headers define (non-template) kernels; this translation unit sees them under a namespace of its own.
#define raft raft_wave_tu
#include "pack.hpp"
#include "pileup_wave.hpp"
#undef raft

namespace raft {
using namespace raft_wave_tu;

namespace {
// (coordinate columns hold three registers per record slot, twice: their instantiations may ask for fewer waves per SIMD)
#ifndef RAFT_WAVE_WPS_COLS
#define RAFT_WAVE_WPS_COLS RAFT_WAVE_WPS
#endif
constexpr int kWpsWin = RAFT_WAVE_WPS, kWpsCols = RAFT_WAVE_WPS_COLS, kWpb = RAFT_WAVE_WPB;
// prefetch slots per lane (64 records each) by the number of sorted runs: a HiFi wave tile of ~1700 windows holds ~250 records
constexpr int kIterWin = RAFT_WAVE_SLOTS / 1024 + 1;
// (coordinate columns: three registers per slot, held twice -- the current tile's and the next one's; what does not fit the
// slots is fetched synchronously)
#ifndef RAFT_WAVE_COLS_ITER
#define RAFT_WAVE_COLS_ITER 4
#endif
constexpr int kIterCols = RAFT_WAVE_COLS_ITER;
constexpr int kWaveSlots = RAFT_WAVE_SLOTS;

template <int OW, int IN>
void launch_ow(hipStream_t st, int n_seg, const TileCut *cuts, const PileupArgs &pa, int n_waves)
{
    constexpr int kIter = IN == 1 ? kIterWin : kIterCols;
    constexpr int kWps = IN == 1 ? kWpsWin : kWpsCols;
    const unsigned grid = (unsigned)((n_waves + kWpb - 1) / kWpb);
    if (n_seg <= 1)
        hipLaunchKernelGGL((pileup_wave_kernel<kWaveSlots, 1, kIter + 1, OW, IN, kWpb, kWps>), dim3(grid), dim3(64 * kWpb), 0, st, cuts, pa);
    else if (n_seg == 2)
        hipLaunchKernelGGL((pileup_wave_kernel<kWaveSlots, 2, 2 * kIter, OW, IN, kWpb, kWps>), dim3(grid), dim3(64 * kWpb), 0, st, cuts, pa);
    else if constexpr (IN == 0)
        hipLaunchKernelGGL((pileup_wave_kernel<kWaveSlots, 4, 4 * (kIter - 1 < 3 ? kIter - 1 : 3), OW, IN, kWpb, kWps>), dim3(grid), dim3(64 * kWpb), 0, st, cuts, pa);
}
} // namespace

int wave_grid_waves(bool win) { return 256 * 4 * (win ? kWpsWin : kWpsCols); }

void launch_wave_variant(int ow, bool win, hipStream_t st, int n_seg, const void *cuts_v, const void *pa_v, int n_waves)
{
    const TileCut *cuts = static_cast<const TileCut *>(cuts_v);
    const PileupArgs &pa = *static_cast<const PileupArgs *>(pa_v);
    if (win) {
        if (ow == kCovDelta4) launch_ow<kCovDelta4, 1>(st, n_seg, cuts, pa, n_waves);
        else if (ow == 1) launch_ow<1, 1>(st, n_seg, cuts, pa, n_waves);
        else if (ow == 2) launch_ow<2, 1>(st, n_seg, cuts, pa, n_waves);
        else launch_ow<4, 1>(st, n_seg, cuts, pa, n_waves);
    } else {
        if (ow == kCovDelta4) launch_ow<kCovDelta4, 0>(st, n_seg, cuts, pa, n_waves);
        else if (ow == 1) launch_ow<1, 0>(st, n_seg, cuts, pa, n_waves);
        else if (ow == 2) launch_ow<2, 0>(st, n_seg, cuts, pa, n_waves);
        else launch_ow<4, 0>(st, n_seg, cuts, pa, n_waves);
    }
}

} // namespace raft
