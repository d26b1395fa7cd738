// probe13 — what does a LONE wave on a SIMD pay per instruction kind?  (the latency layout's systolic step, dspi_chain_skew.inc, is one such
// wave per SIMD: config 2's step measured ~245 cycles for ~38 instructions, config 2b's ~219 for ~30 — more than probe2's ~5 cycles per
// instruction explains).  One wave per SIMD (256 workgroups of 256 threads), a loop of ITERS iterations around a body of 32 independent packed
// multiply-adds plus the thing measured; cycles from s_memtime inside the wave (lane 0 of wave 0 of block 0), ns from HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o probe13 probe13.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define F(a) "v_pk_fma_f32 v[" #a ":" #a "+1], v[" #a ":" #a "+1], v[100:101], v[102:103]\n\t"
#define IND8 F(10) F(12) F(14) F(16) F(18) F(20) F(22) F(24)
#define IND16 IND8 F(26) F(28) F(30) F(32) F(34) F(36) F(38) F(40)
#define IND32 IND16 IND16
#define D F(10)
#define DEP8 D D D D D D D D
#define DEP32 DEP8 DEP8 DEP8 DEP8
#define CM "v_cndmask_b32_e64 v10, v10, v11, s[20:21]\n\t"
#define CM8 CM CM CM CM CM CM CM CM
#define BR(n) "s_branch 1f\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v50, 0\n\t1:\n\t"
#define NT "s_cmp_eq_u32 s22, 77\n\ts_cbranch_scc1 1f\n\t1:\n\t"
#define ST "s_mov_b64 exec, s[20:21]\n\tds_write_b64 v60, v[10:11]\n\ts_mov_b64 exec, -1\n\t"
#define STP "ds_write_b64 v60, v[10:11]\n\t"
#define RD "ds_read_b64 v[62:63], v60 offset:512\n\t"
#define W0 "s_waitcnt lgkmcnt(0)\n\t"
#define DPP2 "s_nop 1\n\tv_mov_b32_dpp v52, v10 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v53, v11 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
// the recurrence of the all-biquad step: neighbour's output -> y -> select -> neighbour
#define HOP "s_nop 1\n\tv_mov_b32_dpp v52, v10 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v53, v11 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
            "v_pk_fma_f32 v[54:55], v[100:101], v[52:53], v[102:103]\n\t" \
            "v_cndmask_b32_e64 v10, v52, v54, s[20:21]\n\tv_cndmask_b32_e64 v11, v53, v55, s[20:21]\n\t"
#define SA "s_add_i32 s22, s22, 1\n\t"
#define SA8 SA SA SA SA SA SA SA SA
#define IS8(x) F(10) x F(12) x F(14) x F(16) x F(18) x F(20) x F(22) x F(24) x
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41", \
             "v50","v52","v53","v54","v55","v60","v62","v63","v100","v101","v102","v103","s20","s21","s22","memory","scc"

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = 0.f;
    asm volatile("v_mov_b32 v100, 0x3f7fff00\n\tv_mov_b32 v101, 0x3f7fff00\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
                 "s_mov_b64 s[20:21], 0x1111\n\ts_mov_b32 s22, 0\n\t"
                 "v_lshlrev_b32 v60, 3, %0\n\t"
                 "v_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\t" :: "v"(threadIdx.x) : CLOB);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(IND32 ::: CLOB);
        if (MODE == 1) asm volatile(DEP32 ::: CLOB);
        if (MODE == 2) asm volatile(CM8 CM8 CM8 CM8 ::: CLOB);
        if (MODE == 3) asm volatile(IND32 BR(0) ::: CLOB);
        if (MODE == 4) asm volatile(IND8 BR(0) IND8 BR(0) IND8 BR(0) IND8 BR(0) ::: CLOB);
        if (MODE == 5) asm volatile(IND32 STP W0 ::: CLOB);
        if (MODE == 6) asm volatile(STP IND16 W0 IND16 ::: CLOB);
        if (MODE == 7) asm volatile(STP IND32 W0 ::: CLOB);
        if (MODE == 8) asm volatile(IND32 RD W0 ::: CLOB);
        if (MODE == 9) asm volatile(IND32 ST ::: CLOB);
        if (MODE == 10) asm volatile(IND32 DPP2 ::: CLOB);
        if (MODE == 11) asm volatile(HOP HOP HOP HOP HOP HOP HOP HOP ::: CLOB);
        if (MODE == 12) asm volatile(IND8 NT IND8 NT IND8 NT IND8 NT ::: CLOB);
        if (MODE == 13) asm volatile(IS8(SA) IS8(SA) IND16 ::: CLOB);
        if (MODE == 14) asm volatile(IND32 SA8 SA8 ::: CLOB);
        if (MODE == 15) asm volatile(RD IND16 W0 IND16 ::: CLOB);
        if (MODE == 16) asm volatile(RD IND32 W0 ::: CLOB);
        if (MODE == 17) asm volatile(IND32 ST W0 ::: CLOB);
        if (MODE == 18) asm volatile(IND32 "s_nop 7\n\ts_nop 7\n\t" ::: CLOB);
        if (MODE == 20) asm volatile(IND32 STP ::: CLOB);
        if (MODE == 21) asm volatile(IND16 ST IND16 ST ::: CLOB);
        if (MODE == 22) asm volatile(IND16 STP IND16 STP ::: CLOB);
        if (MODE == 23) asm volatile(IND32 "s_mov_b64 exec, s[20:21]\n\tds_write2_b64 v60, v[10:11], v[12:13] offset0:0 offset1:1\n\ts_mov_b64 exec, -1\n\t" ::: CLOB);
        if (MODE == 19) asm volatile(IND16 "v_max_f32 v50, |v10|, v50\n\tv_max_f32 v50, |v11|, v50\n\tv_add_u32 v52, 8, v60\n\tv_mov_b32 v53, v11\n\t"
                                     "v_max_f32 v50, |v12|, v50\n\tv_max_f32 v50, |v13|, v50\n\tv_add_u32 v52, 8, v60\n\tv_mov_b32 v53, v12\n\t"
                                     "v_max_f32 v50, |v14|, v50\n\tv_max_f32 v50, |v15|, v50\n\tv_add_u32 v52, 8, v60\n\tv_mov_b32 v53, v13\n\t"
                                     "v_max_f32 v50, |v16|, v50\n\tv_max_f32 v50, |v17|, v50\n\tv_add_u32 v52, 8, v60\n\tv_mov_b32 v53, v14\n\t" ::: CLOB);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r;
    asm volatile("v_add_f32 %0, v10, v11\n\tv_add_f32 %0, %0, v50" : "=v"(r) :: CLOB);
    if (r == 12345.f) out[8 + threadIdx.x] = (uint64_t)lds[(threadIdx.x * 7) & 4095];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(uint64_t *dout, const char *what, double base_cyc, double base_ns) {
    const int iters = 200000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, dout, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f; uint64_t cyc = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, dout, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost)); }
    }
    const double ns = best * 1e6 / iters, c = (double)cyc / iters;
    printf("| %2d | %-96s | %7.1f | %7.1f | %+7.1f | %+7.1f |\n", MODE, what, ns, c, ns - base_ns, c - base_cyc);
}

int main() {
    uint64_t *dout; CK(hipMalloc(&dout, 1 << 16));
    // the baseline first, measured by itself
    const int iters = 200000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, dout, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); uint64_t cyc; CK(hipMemcpy(&cyc, dout, 8, hipMemcpyDeviceToHost));
    const double bn = ms * 1e6 / iters, bc = (double)cyc / iters;
    printf("one wave per SIMD, %d iterations; s_memtime ticks per ns: %.3f\n", iters, bc / bn);
    printf("| mode | body of one iteration | ns | s_memtime ticks | ns over 32 independent | ticks over |\n|---|---|---|---|---|---|\n");
    run<0>(dout, "32 independent v_pk_fma_f32", bc, bn);
    run<1>(dout, "32 DEPENDENT v_pk_fma_f32", bc, bn);
    run<2>(dout, "32 dependent v_cndmask_b32_e64 (SGPR-pair mask)", bc, bn);
    run<3>(dout, "32 independent + 1 taken s_branch (over two instructions)", bc, bn);
    run<4>(dout, "32 independent + 4 taken s_branch", bc, bn);
    run<12>(dout, "32 independent + 4 x (s_cmp + s_cbranch NOT taken)", bc, bn);
    run<13>(dout, "32 independent with 16 s_add_i32 interleaved one by one", bc, bn);
    run<14>(dout, "32 independent, then 16 s_add_i32 in a row", bc, bn);
    run<18>(dout, "32 independent + s_nop 7 x 2 (16 idle issue slots)", bc, bn);
    run<19>(dout, "16 independent + 16 plain VALU (v_max |x|, v_add_u32, v_mov)", bc, bn);
    run<5>(dout, "32 independent + ds_write_b64 + s_waitcnt lgkmcnt(0) at once", bc, bn);
    run<6>(dout, "ds_write_b64, 16 independent, s_waitcnt lgkmcnt(0), 16 independent", bc, bn);
    run<7>(dout, "ds_write_b64, 32 independent, s_waitcnt lgkmcnt(0)", bc, bn);
    run<8>(dout, "32 independent + ds_read_b64 + s_waitcnt lgkmcnt(0) at once", bc, bn);
    run<15>(dout, "ds_read_b64, 16 independent, s_waitcnt lgkmcnt(0), 16 independent", bc, bn);
    run<16>(dout, "ds_read_b64, 32 independent, s_waitcnt lgkmcnt(0)", bc, bn);
    run<9>(dout, "32 independent + EXEC-masked ds_write_b64 (s_mov exec / write / s_mov exec), no wait", bc, bn);
    run<17>(dout, "32 independent + EXEC-masked ds_write_b64 + s_waitcnt lgkmcnt(0)", bc, bn);
    run<20>(dout, "32 independent + PLAIN ds_write_b64 (all 64 lanes, 8-byte stride: conflict-free), no wait", bc, bn);
    run<21>(dout, "2 x (16 independent + EXEC-masked ds_write_b64), no wait", bc, bn);
    run<22>(dout, "2 x (16 independent + plain ds_write_b64), no wait", bc, bn);
    run<23>(dout, "32 independent + ONE EXEC-masked ds_write2_b64 (two frames in one instruction), no wait", bc, bn);
    run<10>(dout, "32 independent + s_nop 1 + 2 v_mov_b32_dpp row_shr:1", bc, bn);
    run<11>(dout, "8 x the all-biquad recurrence (s_nop 1, 2 DPP, v_pk_fma, 2 v_cndmask): 48 instructions, all dependent", bc, bn);
    return 0;
}
