// probe9 — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS repo's access widths:
// the microarchitecture guide calibrates "FETCH_SIZE reads 1/2 of a wide coalesced stream" for 16 B/lane loads only; the chain
// kernels read [position][128 streams] rows with 8 B per lane (512-byte rows per wave) and the Q28 kernel with 4 B per lane.
// Every kernel below moves a KNOWN number of bytes (2 GiB, well past the 256 MiB Infinity Cache):
//   p9_rd4 / p9_rd8 / p9_rd16   coalesced row loads, 4 / 8 / 16 bytes per lane
//   p9_wr8 / p9_wr16            coalesced row stores
//   p9_wr16_lines               16-byte stores, one 128-byte line per lane, eight consecutive stores fill the line (the direct
//                               pair-line pattern of a wave that holds both sides of an S/PDIF pair: 64 lines per store instruction)
//   p9_wr4_stride8              4-byte stores at an 8-byte stride (one side of a pair written in place: every line half-filled)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes): tools/probe/probe9_run.sh divides the counters
// by the known bytes.  The program also prints each kernel's own GB/s (hipEvents).
// Build: hipcc --offload-arch=gfx950 -O3 -o probe9 probe9.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr size_t kBytes = 2ull << 30;
constexpr int kRows = 16;      // rows per wave visit, like a 16-frame chunk

__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(u32x2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <typename V>
__device__ __forceinline__ void rd_rows(const V *__restrict__ src, uint32_t *sink, size_t n_vec) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    uint32_t acc = 0;
    for (size_t base = wave * (kRows * 64); base + kRows * 64 <= n_vec; base += n_waves * (kRows * 64)) {
        V v[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) v[i] = src[base + i * 64 + lane];
#pragma unroll
        for (int i = 0; i < kRows; ++i) acc ^= fold(v[i]);
    }
    if (acc == 0x12345u) sink[0] = acc;      // never true for the zero-filled buffer plus the pattern below
}
template <typename V>
__device__ __forceinline__ void wr_rows(V *__restrict__ dst, size_t n_vec, V val) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t base = wave * (kRows * 64); base + kRows * 64 <= n_vec; base += n_waves * (kRows * 64)) {
#pragma unroll
        for (int i = 0; i < kRows; ++i) dst[base + i * 64 + lane] = val;
    }
}
__global__ void p9_rd4(const uint32_t *s, uint32_t *k, size_t n) { rd_rows<uint32_t>(s, k, n); }
__global__ void p9_rd8(const u32x2 *s, uint32_t *k, size_t n) { rd_rows<u32x2>(s, k, n); }
__global__ void p9_rd16(const u32x4 *s, uint32_t *k, size_t n) { rd_rows<u32x4>(s, k, n); }
__global__ void p9_wr8(u32x2 *d, size_t n) { wr_rows<u32x2>(d, n, u32x2{threadIdx.x, blockIdx.x}); }
__global__ void p9_wr16(u32x4 *d, size_t n) { wr_rows<u32x4>(d, n, u32x4{threadIdx.x, blockIdx.x, 1u, 2u}); }
// one 128-byte line per lane and visit: lane l owns line (visit * 64 + l); store j writes its bytes [16 j, 16 j + 16)
__global__ void p9_wr16_lines(u32x4 *d, size_t n_lines) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t line = wave * 64 + lane; line < n_lines; line += n_waves * 64) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[line * 8 + j] = u32x4{(uint32_t)line, (uint32_t)j, 1u, 2u};
    }
}
// lane l owns line (visit * 64 + l) and writes words 0, 2, 4 .. 30 of it (the left side of sixteen frames)
__global__ void p9_wr4_stride8(uint32_t *d, size_t n_lines) {
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t line = wave * 64 + lane; line < n_lines; line += n_waves * 64) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d[line * 32 + 2 * j] = (uint32_t)line + j;
    }
}

int main() {
    void *buf; uint32_t *sink;
    CK(hipMalloc(&buf, kBytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, kBytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(256 * 8), block(256);
    auto report = [&](const char *name, double bytes, float ms) { printf("%-16s %8.3f ms  %8.1f GB/s  (%.0f bytes known)\n", name, ms, bytes / ms * 1e-6, bytes); };
    float ms;
#define RUN(name, bytes, ...) \
    for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(name, grid, block, 0, 0, __VA_ARGS__); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); } \
    report(#name, bytes, ms);
    RUN(p9_rd4, (double)kBytes, (const uint32_t *)buf, sink, kBytes / 4)
    RUN(p9_rd8, (double)kBytes, (const u32x2 *)buf, sink, kBytes / 8)
    RUN(p9_rd16, (double)kBytes, (const u32x4 *)buf, sink, kBytes / 16)
    RUN(p9_wr8, (double)kBytes, (u32x2 *)buf, kBytes / 8)
    RUN(p9_wr16, (double)kBytes, (u32x4 *)buf, kBytes / 16)
    RUN(p9_wr16_lines, (double)kBytes, (u32x4 *)buf, kBytes / 128)
    RUN(p9_wr4_stride8, (double)kBytes / 2, (uint32_t *)buf, kBytes / 128)
    return 0;
}
