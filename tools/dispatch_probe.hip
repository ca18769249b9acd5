// Which workgroups of a launch share a CU?  Each workgroup records (XCC, SE, CU) from its hardware-id registers and its start
// time; the host prints, per CU, the workgroup ids it ran -- the basis for a blockIdx -> tile mapping that lets co-resident
// workgroups share operand tiles in the CU's vector L1.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o tools/dispatch_probe.bin && tools/dispatch_probe.bin [grid] [lds_kb] [spin_us]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned* out, unsigned long long* t0, long spin_ticks) {
    extern __shared__ float lds[];
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hwid;
        out[2 * blockIdx.x + 1] = xcc;
        t0[blockIdx.x] = start;
    }
    lds[threadIdx.x] = (float)hwid;
    while ((long)(__builtin_amdgcn_s_memrealtime() - start) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] < 0.f) out[0] = 0;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 1280;
    const int lds_kb = argc > 2 ? atoi(argv[2]) : 18;
    const double spin_us = argc > 3 ? atof(argv[3]) : 8.0;
    unsigned* out; unsigned long long* t0;
    hipMalloc(&out, sizeof(unsigned) * 2 * grid); hipMalloc(&t0, sizeof(unsigned long long) * grid);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds_kb * 1024, 0, out, t0, (long)(spin_us * 100.0));   // 100 MHz realtime clock
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * grid); std::vector<unsigned long long> ht(grid);
    hipMemcpy(h.data(), out, sizeof(unsigned) * 2 * grid, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t0, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    const unsigned long long first = *std::min_element(ht.begin(), ht.end());
    std::map<unsigned, std::vector<int>> per_cu;
    for (int b = 0; b < grid; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("grid %d, %d KB LDS, spin %.1f us: %zu distinct CUs\n", grid, lds_kb, spin_us, per_cu.size());
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 12) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 0xf);
        for (int b : kv.second) printf(" %d(+%.2fus)", b, (double)(ht[b] - first) / 100.0);
        printf("\n");
    }
    // how regular is it?  difference between consecutive workgroup ids on one CU
    std::map<int, int> hist;
    for (auto& kv : per_cu)
        for (size_t i = 1; i < kv.second.size(); ++i) hist[kv.second[i] - kv.second[i - 1]]++;
    printf("id difference between consecutive workgroups of a CU:");
    for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
