// Hardware probe (round 1): answers design questions for the DSPi chain kernel.
//  (a) v_mul/v_add f32 scalar vs packed (v_pk_*) issue rate
//  (b) f32 FTZ boundary semantics vs x86 MXCSR FTZ|DAZ
//  (c) f32 division / int conversion exactness vs x86
//  (d) float4 streaming copy bandwidth
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fgpu-flush-denormals-to-zero probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

// ---- (a) VALU rate: biquad-like dependent chains, unfused mul+add ----
template <int ILP>
__global__ void valu_scalar(float* out, float a, float b, int iters) {
    float x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) { x[k] = x[k] * a; x[k] = x[k] + b; }
    }
    float s = 0; for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void valu_packed(float* out, float a, float b, int iters) {
    float2v x[ILP];
    float2v av = {a, a}, bv = {b, b};
    for (int k = 0; k < ILP; k++) { x[k].x = threadIdx.x * 1e-3f + k; x[k].y = x[k].x + 0.5f; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) { x[k] = x[k] * av; x[k] = x[k] + bv; }
    }
    float s = 0; for (int k = 0; k < ILP; k++) s += x[k].x + x[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// integer mul24 rate (Q28 path)
template <int ILP>
__global__ void valu_i24(int* out, int a, int iters) {
    int x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) { x[k] = __mul24(x[k] >> 16, a) + (x[k] & 0xffff); }
    }
    int s = 0; for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- (b,c) semantics ----
__global__ void sem_kernel(const float* a, const float* b, float* mul, float* add, float* dv, int* cvt, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mul[i] = a[i] * b[i]; add[i] = a[i] + b[i]; dv[i] = a[i] / b[i]; cvt[i] = (int)(a[i] * 268435456.0f); }
}

// ---- (d) copy ----
__global__ void copy4(const float4* __restrict in, float4* __restrict out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

static uint32_t rng = 0x12345678u;
static uint32_t xs() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; }

template <typename F> float time_ms(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < reps; r++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz memclk %d kHz L2 %d\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.l2CacheSize);
    float* dout; CK(hipMalloc(&dout, 1 << 26));
    const int iters = 4096, blocks = 256 * 8, threads = 256;
    {
        auto run = [&](const char* name, float ms, double ops_per_thread_iter, int ilp) {
            double ops = (double)blocks * threads * iters * ilp * ops_per_thread_iter;
            printf("%-24s ilp %d: %.3f ms  %.2f T lane-ops/s\n", name, ilp, ms, ops / ms / 1e9);
        };
        run("scalar mul+add", time_ms([&] { valu_scalar<1><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 2, 1);
        run("scalar mul+add", time_ms([&] { valu_scalar<4><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 2, 4);
        run("scalar mul+add", time_ms([&] { valu_scalar<8><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 2, 8);
        run("packed mul+add (x2 elem)", time_ms([&] { valu_packed<1><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 4, 1);
        run("packed mul+add (x2 elem)", time_ms([&] { valu_packed<4><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 4, 4);
        run("packed mul+add (x2 elem)", time_ms([&] { valu_packed<8><<<blocks, threads>>>(dout, 0.999f, 1e-3f, iters); }, 5), 4, 8);
        run("i24 mad+shift+and", time_ms([&] { valu_i24<4><<<blocks, threads>>>((int*)dout, 12345, iters); }, 5), 3, 4);
    }
    // semantics
    {
        const int n = 1 << 20;
        std::vector<float> a(n), b(n), mul(n), add(n), dv(n); std::vector<int> cv(n);
        for (int i = 0; i < n; i++) {
            uint32_t ua, ub; int mode = i & 7;
            if (mode < 3) { // products landing around the min-normal boundary
                ua = (xs() & 0x007fffffu) | ((uint32_t)(1 + (xs() % 40)) << 23);   // tiny normal 2^-126..2^-87
                ub = (xs() & 0x007fffffu) | ((uint32_t)(127 - (xs() % 42)) << 23); // 2^-41 .. 1
            } else if (mode == 3) { ua = xs() & 0x807fffffu; ub = xs(); }           // denormal inputs
            else if (mode == 4) { ua = 0x00800000u + (xs() & 7); ub = 0x3f7ffff0u + (xs() & 0x1f); } // right at boundary
            else { ua = xs(); ub = xs(); }
            if (i & 8) ua ^= 0x80000000u;
            memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4);
        }
        float *da, *db, *dm, *dad, *dd; int* dc;
        CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&db, n * 4)); CK(hipMalloc(&dm, n * 4)); CK(hipMalloc(&dad, n * 4)); CK(hipMalloc(&dd, n * 4)); CK(hipMalloc(&dc, n * 4));
        CK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
        sem_kernel<<<n / 256, 256>>>(da, db, dm, dad, dd, dc, n);
        CK(hipMemcpy(mul.data(), dm, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(add.data(), dad, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dv.data(), dd, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cv.data(), dc, n * 4, hipMemcpyDeviceToHost));
        unsigned csr = _mm_getcsr(); _mm_setcsr(csr | 0x8040u); // FTZ | DAZ
        long bad_mul = 0, bad_add = 0, bad_div = 0, bad_cvt = 0, nanskip = 0; int shown = 0;
        for (int i = 0; i < n; i++) {
            volatile float va = a[i], vb = b[i];
            float hm = va * vb, ha = va + vb, hd = va / vb;
            uint32_t g, h;
            auto cmp = [&](float gv, float hv, long& cnt, const char* what) {
                memcpy(&g, &gv, 4); memcpy(&h, &hv, 4);
                if (g != h) { if (gv != gv && hv != hv) { nanskip++; return; } cnt++;
                    if (shown < 24) { uint32_t ua, ub; memcpy(&ua, &a[i], 4); memcpy(&ub, &b[i], 4);
                        printf("  %s mismatch a=%08x b=%08x gpu=%08x cpu=%08x\n", what, ua, ub, g, h); shown++; } }
            };
            cmp(mul[i], hm, bad_mul, "mul"); cmp(add[i], ha, bad_add, "add"); cmp(dv[i], hd, bad_div, "div");
            float t = va * 268435456.0f; long long w = (long long)t; int sat = (t != t) ? 0 : (w > 2147483647LL ? 2147483647 : (w < -2147483648LL ? (int)0x80000000 : (int)w));
            if (fabsf(t) < 1e18f && sat != cv[i]) { bad_cvt++; if (shown < 24) { printf("  cvt mismatch t=%g gpu=%d sat=%d\n", t, cv[i], sat); shown++; } }
        }
        _mm_setcsr(csr);
        printf("semantics vs x86(FTZ|DAZ): n=%d mul_bad=%ld add_bad=%ld div_bad=%ld cvt_bad(vs saturating)=%ld nan_payload_diffs=%ld\n", n, bad_mul, bad_add, bad_div, bad_cvt, nanskip);
    }
    // copy bandwidth
    {
        size_t bytes = (size_t)2 << 30; float4 *i4, *o4; CK(hipMalloc(&i4, bytes)); CK(hipMalloc(&o4, bytes));
        CK(hipMemset(i4, 1, bytes));
        for (int g : {2048, 4096, 16384}) {
            float ms = time_ms([&] { copy4<<<g, 256>>>(i4, o4, bytes / 16); }, 5);
            printf("copy float4 grid %d: %.3f ms  %.2f TB/s (r+w)\n", g, ms, 2.0 * bytes / ms / 1e9);
        }
    }
    return 0;
}
