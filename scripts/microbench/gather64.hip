// Micro-benchmark: what does a 64-byte-per-lane random gather cost on one CU's vector L1?
//   A: every lane issues 4 x global_load_dwordx4 on its own 64-B record (the BVH pair fetch today)
//   B: the 4 lanes of a quad fetch ONE record per instruction (16 B each, one coalesced 64-B access), four
//      records in four instructions, and the pieces are transposed through LDS
//   C: the same gather served from LDS (4 x ds_read_b128 per lane)
// Each lane follows a dependent chain (next index from the loaded data), like a traversal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline float4 ld16(const void* b, uint32_t off) { return *(const float4*)((const char*)b + off); }
__device__ inline uint32_t mix(float4 a, float4 b, float4 c, float4 d) {
    return __float_as_uint(a.x) + __float_as_uint(a.w) * 3u + __float_as_uint(b.y) * 5u + __float_as_uint(c.z) * 7u + __float_as_uint(d.w) * 11u + __float_as_uint(d.x);
}

__global__ __launch_bounds__(256) void kA(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t gt = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = gt * 2654435761u & mask, acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t off = idx << 6;
        float4 a = ld16(buf, off), b = ld16(buf, off + 16), c = ld16(buf, off + 32), d = ld16(buf, off + 48);
        uint32_t m = mix(a, b, c, d);
        acc += m;
        idx = (m + i + gt * 0x9E3779B9u) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// A2: the same chain, but only the second 32 bytes of each record are read (2 x dwordx4 per lane)
__global__ __launch_bounds__(256) void kA2(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t gt = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = gt * 2654435761u & mask, acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t off = idx << 6;
        float4 c = ld16(buf, off + 32), d = ld16(buf, off + 48);
        uint32_t m = mix(c, d, c, d);
        acc += m;
        idx = (m + i + gt * 0x9E3779B9u) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void kB(const void* buf, uint32_t mask, int iters, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char stage[256 * 80];
    const uint32_t tid = threadIdx.x, j = tid & 3u, q = tid & ~3u;
    const uint32_t gt = blockIdx.x * 256 + tid;
    uint32_t idx = gt * 2654435761u & mask, acc = 0;
    for (int i = 0; i < iters; ++i) {
        float4 p[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t src = __shfl(idx, (int)((tid & 63u & ~3u) + m), 64);
            p[m] = ld16(buf, (src << 6) + j * 16u);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) *(float4*)(stage + (q + m) * 80u + j * 16u) = p[m];
        // quad lanes are in the same wave: LDS ops of one wave are ordered, no barrier needed
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        float4 a = *(float4*)(stage + tid * 80u), b = *(float4*)(stage + tid * 80u + 16), c = *(float4*)(stage + tid * 80u + 32),
               d = *(float4*)(stage + tid * 80u + 48);
        uint32_t m = mix(a, b, c, d);
        acc += m;
        idx = (m + i + gt * 0x9E3779B9u) & mask;
    }
    out[blockIdx.x * 256 + tid] = acc;
}

__global__ __launch_bounds__(256) void kC(const void* buf, uint32_t mask_unused, int iters, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char top[1024 * 64];  // 1024 records in LDS
    for (uint32_t k = threadIdx.x; k < 1024 * 4; k += 256) *(float4*)(top + k * 16) = ld16(buf, k * 16);
    __syncthreads();
    const uint32_t gt = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = gt * 2654435761u & 1023u, acc = 0;
    for (int i = 0; i < iters; ++i) {
        const char* r = top + (idx << 6);
        float4 a = *(const float4*)r, b = *(const float4*)(r + 16), c = *(const float4*)(r + 32), d = *(const float4*)(r + 48);
        uint32_t m = mix(a, b, c, d);
        acc += m;
        idx = (m + i + gt * 0x9E3779B9u) & 1023u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    int iters = 200, grid = 2048;
    uint32_t* out; CK(hipMalloc(&out, grid * 256 * 4));
    std::vector<uint32_t> ha(grid * 256), hb(grid * 256);
    for (int lg = 10; lg <= 24; lg += 2) {  // records: 64 KiB .. 1 GiB
        size_t recs = (size_t)1 << lg, bytes = recs * 64;
        void* buf; CK(hipMalloc(&buf, bytes));
        std::vector<uint32_t> h(bytes / 4);
        uint32_t s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 4; }
        CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms[4];
        for (int k = 0; k < 4; ++k) {
            if (k == 2 && lg != 10) { ms[2] = 0; continue; }
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(recs - 1), iters, out);
                if (k == 1) hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(recs - 1), iters, out);
                if (k == 3) hipLaunchKernelGGL(kA2, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(recs - 1), iters, out);
                if (k == 2) hipLaunchKernelGGL(kC, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(recs - 1), iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[k], e0, e1));
            }
            if (k == 0) CK(hipMemcpy(ha.data(), out, ha.size() * 4, hipMemcpyDeviceToHost));
            if (k == 1) CK(hipMemcpy(hb.data(), out, hb.size() * 4, hipMemcpyDeviceToHost));
        }
        double gathers = (double)grid * 256 * iters;
        printf("records 2^%d (%8.1f MiB): A per-lane %.3f ms (%.1f G rec/s, %.0f GB/s)  B quad-coop %.3f ms (%.1f G rec/s, %.0f GB/s)  %s", lg,
               bytes / 1048576.0, ms[0], gathers / ms[0] / 1e6, gathers * 64 / ms[0] / 1e6, ms[1], gathers / ms[1] / 1e6,
               gathers * 64 / ms[1] / 1e6, ha == hb ? "match" : "MISMATCH");
        printf("  A2 per-lane 32B %.3f ms (%.1f G rec/s)", ms[3], gathers / ms[3] / 1e6);
        if (ms[2] > 0) printf("  C lds %.3f ms (%.1f G rec/s)", ms[2], gathers / ms[2] / 1e6);
        printf("\n");
        CK(hipFree(buf));
    }
    return 0;
}
