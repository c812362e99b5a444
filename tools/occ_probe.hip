// How many 256-thread workgroups share a CU for a given LDS footprint / VGPR budget?  Each workgroup of a 256*K grid
// notes s_memrealtime when it starts, spins ~300 us and exits; workgroups that start late were not co-resident.
// build: hipcc -O2 --offload-arch=gfx950 tools/occ_probe.hip -o /tmp/occ_probe ; run: /tmp/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int NV, int SG = 0>
__global__ __launch_bounds__(256) void probe(unsigned long long *start, int spin_ticks)
{
    extern __shared__ int lds[];
    // SG: touch a high scalar register so that the wave's SGPR allocation grows (16-register granules)
    if (SG == 1) asm volatile("s_mov_b32 s79, 0" ::: "s79");
    if (SG == 2) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    if (SG == 3) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    float keep[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) keep[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    lds[threadIdx.x] = (int)t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < NV; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += keep[i];
    if (s == 12345.678f) start[blockIdx.x] = (unsigned long long)lds[(threadIdx.x + 1) & 255];
}

template <int NV, int SG = 0>
int late_count(int k, int lds_bytes)
{
    const int grid = 256 * k;
    unsigned long long *d;
    hipMalloc(&d, grid * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<NV, SG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL((probe<NV, SG>), dim3(grid), dim3(256), lds_bytes, 0, d, 30000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    const unsigned long long o = *std::min_element(h.begin(), h.end());
    int late = 0;
    for (auto v : h) if (v - o > 15000) ++late;   // started more than 150 us after the first
    return late;
}

int main()
{
    for (int k : {5}) {
        printf("K=%d workgroups per CU, small register footprint: late workgroups by LDS bytes\n", k);
        for (int lds = 30 * 1024; lds <= 33 * 1024; lds += 256) printf("  lds %6d late %d\n", lds, late_count<8>(k, lds));
    }
    printf("K=5, lds 30832, register footprint sweep (floats kept live per thread)\n");
    printf("  NV 40 late %d\n", late_count<40>(5, 30832));
    printf("  NV 64 late %d\n", late_count<64>(5, 30832));
    printf("  NV 72 late %d\n", late_count<72>(5, 30832));
    printf("  NV 80 late %d\n", late_count<80>(5, 30832));
    printf("  NV 88 late %d\n", late_count<88>(5, 30832));
    printf("K=5, lds 30832, 84 VGPRs, scalar register footprint: s79 late %d, s95 late %d, s99 late %d\n", late_count<80, 1>(5, 30832),
           late_count<80, 2>(5, 30832), late_count<80, 3>(5, 30832));
    printf("K=4 same: s79 late %d, s95 late %d, s99 late %d\n", late_count<80, 1>(4, 30832), late_count<80, 2>(4, 30832), late_count<80, 3>(4, 30832));
    printf("K=5, lds 30832, 44 VGPRs: s79 late %d, s95 late %d, s99 late %d\n", late_count<40, 1>(5, 30832), late_count<40, 2>(5, 30832), late_count<40, 3>(5, 30832));
    return 0;
}
