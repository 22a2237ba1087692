// Calibration of rocprofv3's FETCH_SIZE for the traversal's access shape: known byte counts read ONCE
// from a table far beyond the Infinity Cache, per access shape.  Run under `rocprofv3 --pmc FETCH_SIZE` (scripts/fetch_calib.sh).
//   k_stream     : wide coalesced read of the whole table, 16 B per lane (the guide's reference shape: FETCH_SIZE = 1/2 bytes)
//   k_gather64   : lane i reads the 64-byte record perm[i] with 4 x global_load_dwordx4 (a child-pair fetch)
//   k_gather48   : lane i reads 48 bytes of a 48-byte record perm[i] (a triangle fetch)
//   k_gather128  : lane i reads the 128-byte record perm[i] (8 loads)
//   k_gather16   : lane i reads 16 bytes of the 64-byte record perm[i] (one sector of a line)
// Every record is read exactly once (perm is a permutation), so the bytes REQUESTED are records x size; the permutation itself
// is another 4 bytes per record, read coalesced.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ inline float4 ld16(const void* b, size_t off) { return *(const float4*)((const char*)b + off); }
__device__ inline uint32_t mix1(float4 a) { return __float_as_uint(a.x) ^ __float_as_uint(a.y) ^ __float_as_uint(a.z) ^ __float_as_uint(a.w); }

__global__ __launch_bounds__(256) void k_stream(const void* buf, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) acc ^= mix1(ld16(buf, i * 16));
    if (acc == 0x12345678u) out[0] = acc;
}
template <int REC, int LOADS>
__global__ __launch_bounds__(256) void k_gather(const void* buf, const uint32_t* perm, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        const size_t off = (size_t)perm[i] * REC;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) acc ^= mix1(ld16(buf, off + 16 * k));
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20;  // MiB
    void* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    uint32_t* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto make_perm = [&](size_t n, uint32_t** dp) -> int {
        std::vector<uint32_t> p(n);
        std::iota(p.begin(), p.end(), 0u);
        std::mt19937 g(7);
        std::shuffle(p.begin(), p.end(), g);
        CK(hipMalloc(dp, n * 4)); CK(hipMemcpy(*dp, p.data(), n * 4, hipMemcpyHostToDevice));
        return 0;
    };
    uint32_t *p64, *p48, *p128;
    const size_t n64 = bytes / 64, n48 = bytes / 48, n128 = bytes / 128;
    if (make_perm(n64, &p64) || make_perm(n48, &p48) || make_perm(n128, &p128)) return 1;
    const int grid = 8192;
    float ms;
#define RUN(name, requested, ...)                                                                        \
    CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));              \
    CK(hipEventElapsedTime(&ms, e0, e1));                                                                  \
    printf("%-12s requested %14.0f bytes  %8.3f ms  %7.1f GB/s requested\n", name, (double)(requested), ms, (double)(requested) / ms / 1e6);
    RUN("k_stream", bytes, hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, buf, bytes / 16, out));
    RUN("k_gather64", n64 * 64.0 + n64 * 4.0, hipLaunchKernelGGL((k_gather<64, 4>), dim3(grid), dim3(256), 0, 0, buf, p64, n64, out));
    RUN("k_gather48", n48 * 48.0 + n48 * 4.0, hipLaunchKernelGGL((k_gather<48, 3>), dim3(grid), dim3(256), 0, 0, buf, p48, n48, out));
    RUN("k_gather128", n128 * 128.0 + n128 * 4.0, hipLaunchKernelGGL((k_gather<128, 8>), dim3(grid), dim3(256), 0, 0, buf, p128, n128, out));
    RUN("k_gather16", n64 * 16.0 + n64 * 4.0, hipLaunchKernelGGL((k_gather<64, 1>), dim3(grid), dim3(256), 0, 0, buf, p64, n64, out));
    return 0;
}
