// Micro-benchmark: dependent random gathers by TEAMS of lanes (profiles/r03/tuning.md).
//   L64 / L128 : every lane follows its own chain over 64-B / 128-B records (4 / 8 x global_load_dwordx4 per lane and step)
//   T4 / T8    : a team of 4 / 8 consecutive lanes follows ONE chain over 64-B / 128-B records: one dwordx4 per lane and
//                step (lane j of the team reads bytes 16 j ..), the next index comes from the team's data (xor over lanes)
// Each variant is timed with the chip full (2048 blocks x 256) -> records/s, and with ONE wave per CU -> ns per dependent step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline float4 ld16(const void* b, uint32_t off) { return *(const float4*)((const char*)b + off); }
__device__ inline uint32_t mix1(float4 a) { return __float_as_uint(a.x) + __float_as_uint(a.y) * 3u + __float_as_uint(a.z) * 5u + __float_as_uint(a.w) * 7u; }

template <int LOADS>  // per-lane chain over records of LOADS * 16 bytes
__global__ __launch_bounds__(256) void kL(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx = gt * 2654435761u & mask, acc = 0;
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

template <int TEAM>  // a team of TEAM lanes, one chain, records of TEAM * 16 bytes
__global__ __launch_bounds__(256) void kT(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t team = gt / TEAM, j = gt % TEAM;
    uint32_t idx = team * 2654435761u & mask, acc = 0;
    for (int i = 0; i < iters; ++i) {
        const float4 v = ld16(buf, idx * (TEAM * 16u) + 16u * j);
        uint32_t m = mix1(v) * (2u * j + 1u);
#pragma unroll
        for (int s = 1; s < TEAM; s <<= 1) m += __shfl_xor(m, s, 64);  // team-wide sum: every lane holds it
        acc += m;
        idx = (m + i + team * 0x9E3779B9u) & mask;
    }
    out[gt] = acc;
}

int main(int argc, char** argv) {
    int iters = 200;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* out; CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int lg = 12; lg <= 22; lg += 5) {  // bytes: 2^lg * 128 : 512 KiB, 16 MiB, 512 MiB
        const size_t bytes = ((size_t)1 << lg) * 128;
        void* buf; CK(hipMalloc(&buf, bytes));
        std::vector<uint32_t> h(bytes / 4);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
        CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
        const uint32_t m64 = (uint32_t)(bytes / 64 - 1), m128 = (uint32_t)(bytes / 128 - 1);
        printf("table %.1f MiB\n", bytes / 1048576.0);
        for (int mode = 0; mode < 2; ++mode) {  // 0: full chip, 1: one wave per CU (latency of a dependent step)
            const int grid = mode == 0 ? 2048 : cus, block = mode == 0 ? 256 : 64;
            for (int k = 0; k < 4; ++k) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (k == 0) hipLaunchKernelGGL(kL<4>, dim3(grid), dim3(block), 0, 0, buf, m64, iters, out);
                    if (k == 1) hipLaunchKernelGGL(kL<8>, dim3(grid), dim3(block), 0, 0, buf, m128, iters, out);
                    if (k == 2) hipLaunchKernelGGL(kT<4>, dim3(grid), dim3(block), 0, 0, buf, m64, iters, out);
                    if (k == 3) hipLaunchKernelGGL(kT<8>, dim3(grid), dim3(block), 0, 0, buf, m128, iters, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                }
                const char* name[4] = {"L64  per-lane 64 B ", "L128 per-lane 128 B", "T4   team-4  64 B  ", "T8   team-8  128 B "};
                const int team[4] = {1, 1, 4, 8}, rb[4] = {64, 128, 64, 128};
                const double chains = (double)grid * block / team[k], recs = chains * iters;
                if (mode == 0) printf("  full chip  %s: %.3f ms  %.1f G rec/s  %.0f GB/s\n", name[k], ms, recs / ms / 1e6, recs * rb[k] / ms / 1e6);
                else printf("  1 wave/CU  %s: %.3f ms  %.0f ns per dependent step\n", name[k], ms, ms * 1e6 / iters);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
