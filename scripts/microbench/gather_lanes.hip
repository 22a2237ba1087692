// Micro-benchmark: what does a divergent global_load_dwordx4 cost -- per INSTRUCTION or per ACTIVE LANE?  (profiles/r03/tuning.md)
//   L64/k  : every k-th lane follows a chain over 64-B records (4 x dwordx4 per active lane and step), the others idle
//   L32    : all lanes, 32-B records (2 x dwordx4)
//   Q4x4   : a quad follows FOUR chains at once: instruction m loads record of chain m, lane j its bytes 16 j .. (coalesced 64 B per quad);
//            the same number of load instructions and chains per wave as L64/1
// table: 0.5 MiB (L2-resident) and 64 MiB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline float4 ld16(const void* b, uint32_t off) { return *(const float4*)((const char*)b + off); }
__device__ inline uint32_t mix1(float4 a) { return __float_as_uint(a.x) + __float_as_uint(a.y) * 3u + __float_as_uint(a.z) * 5u + __float_as_uint(a.w) * 7u; }

template <int LOADS>
__global__ __launch_bounds__(256) void kL(const void* buf, uint32_t mask, int iters, uint32_t every, uint32_t* out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = gt * 2654435761u & mask, acc = 0;
    if (threadIdx.x % every == 0)
        for (int i = 0; i < iters; ++i) {
            const uint32_t off = idx * (LOADS * 16u);
            float4 v[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; ++k) v[k] = ld16(buf, off + 16u * k);
            uint32_t m = 0;
#pragma unroll
            for (int k = 0; k < LOADS; ++k) m += mix1(v[k]) * (2u * k + 1u);
            acc += m;
            idx = (m + i + gt * 0x9E3779B9u) & mask;
        }
    out[gt] = acc;
}

__global__ __launch_bounds__(256) void kQ(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = gt & 3u, quad = gt >> 2;
    uint32_t idx[4], acc = 0;
    for (int m = 0; m < 4; ++m) idx[m] = (quad * 4u + m) * 2654435761u & mask;
    for (int i = 0; i < iters; ++i) {
        float4 v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = ld16(buf, idx[m] * 64u + 16u * j);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t x = mix1(v[m]) * (2u * j + 1u);
            x += __shfl_xor(x, 1, 64);
            x += __shfl_xor(x, 2, 64);
            acc += x;
            idx[m] = (x + i + (quad * 4u + m) * 0x9E3779B9u) & mask;
        }
    }
    out[gt] = acc;
}

int main(int argc, char** argv) {
    const int iters = 200, grid = 2048;
    uint32_t* out; CK(hipMalloc(&out, grid * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) { const int lg = argc > 1 ? atoi(argv[a]) : 13;
        const size_t bytes = ((size_t)1 << lg) * 64;
        void* buf; CK(hipMalloc(&buf, bytes));
        std::vector<uint32_t> h(bytes / 4);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
        CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
        printf("table %.1f MiB\n", bytes / 1048576.0);
        for (int k = 0; k < 7; ++k) {
            float ms = 0;
            const uint32_t every[7] = {1, 2, 4, 8, 1, 1, 1};
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (k < 4) hipLaunchKernelGGL(kL<4>, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 64 - 1), iters, every[k], out);
                if (k == 4) hipLaunchKernelGGL(kL<2>, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 32 - 1), iters, 1u, out);
                if (k == 5) hipLaunchKernelGGL(kL<1>, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 16 - 1), iters, 1u, out);
                if (k == 6) hipLaunchKernelGGL(kQ, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 64 - 1), iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const char* name[7] = {"L64/1", "L64/2", "L64/4", "L64/8", "L32  ", "L16  ", "Q4x4 "};
            const double waves = grid * 4.0, loads[7] = {4, 4, 4, 4, 2, 1, 4};
            const double recs = (double)grid * 256 / every[k] * iters;
            const double clk_per_instr = ms * 1e-3 * 2.4e9 * 256 / (waves * iters * loads[k]);  // CU-clocks per wave-level load instruction
            printf("  %s: %.3f ms  %.1f G rec/s  %.1f CU-clk per load instruction\n", name[k], ms, recs / ms / 1e6, clk_per_instr);
        }
        CK(hipFree(buf));
    }
    return 0;
}
