// Vector-memory pipe rate of one CU as a function of the lanes a load instruction has active (developer microbenchmark, not part of the product):
// every wave issues independent global_load_dwordx4 (or dwordx2 / dword) from a table that stays in the CU's vector L1, with `active` of its 64
// lanes enabled in one of two patterns (the first lanes / evenly spread).  Prints shader cycles per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O2 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WIDTH> __global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, unsigned mask, int iters, unsigned long long lanes, float* out, int strideB) {
    const unsigned lane = threadIdx.x & 63;
    if (!((lanes >> lane) & 1)) return;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float acc = 0.f;
    const char* base = (const char*)tab;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            h = h * 1664525u + 1013904223u;
            const unsigned off = ((h >> 8) & mask) * (unsigned)strideB;
            if (WIDTH == 4) { float4 v = *(const float4*)(base + off); acc += v.x + v.w; }
            else if (WIDTH == 2) { float2 v = *(const float2*)(base + off); acc += v.x + v.y; }
            else { acc += *(const float*)(base + off); }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const int tableBytes = 1 << 20;
    float4* tab; float* out;
    hipMalloc(&tab, tableBytes); hipMemset(tab, 0, tableBytes); hipMalloc(&out, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
    printf("CUs %d clock %.0f MHz sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", cus, clk / 1e6, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    struct Pat { const char* name; unsigned long long lanes; };
    std::vector<Pat> pats = {{"64", ~0ull}, {"32 first", 0xffffffffull}, {"32 spread", 0x5555555555555555ull}, {"16 first", 0xffffull}, {"16 spread", 0x1111111111111111ull},
                             {"8 first", 0xffull}, {"8 spread", 0x0101010101010101ull}, {"4 first", 0xfull}, {"1", 1ull}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int footprint : {16 * 1024, 512 * 1024}) for (int strideB : {16, 80}) for (int width : {4, 2, 1}) {
        const unsigned mask = (unsigned)(footprint / strideB) - 1 > 0 ? (1u << (31 - __builtin_clz((unsigned)(footprint / strideB)))) - 1 : 0;
        printf("table %d KB, record stride %d B, load width %d dwords: cycles per wave-instruction per CU\n", footprint / 1024, strideB, width);
        for (auto& pt : pats) {
            const int iters = 400, blocksPerCU = 8;
            auto launch = [&]() {
                if (width == 4) hipLaunchKernelGGL(k<4>, dim3(cus * blocksPerCU), dim3(256), 0, 0, tab, mask, iters, pt.lanes, out, strideB);
                else if (width == 2) hipLaunchKernelGGL(k<2>, dim3(cus * blocksPerCU), dim3(256), 0, 0, tab, mask, iters, pt.lanes, out, strideB);
                else hipLaunchKernelGGL(k<1>, dim3(cus * blocksPerCU), dim3(256), 0, 0, tab, mask, iters, pt.lanes, out, strideB);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instrPerCU = (double)blocksPerCU * 4 * iters * 8;
            printf("   lanes %-10s %8.3f ms  %7.2f cycles\n", pt.name, ms, ms * 1e-3 * clk / instrPerCU);
        }
    }
    return 0;
}
