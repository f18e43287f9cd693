// Development probe (not part of the product): sustained issue rate of single VALU instructions on gfx950
// with 8 waves per SIMD -- what a VALU-bound kernel such as Integrate actually pays per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.001f;
    v2f pb = {b, b}, pc = {c, c};
    unsigned u0 = threadIdx.x, u1 = u0 + 7, u2 = u0 + 9, u3 = u0 + 11;
    unsigned long long w0 = u0, w1 = u1;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {   // v_fma_f32 x16
            REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 1) {   // v_pk_fma_f32 x16
            REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if constexpr (OP == 2) {   // v_pk_mul_f32
            REP4(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if constexpr (OP == 3) {   // v_pk_add_f32
            REP4(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));)
        } else if constexpr (OP == 4) {   // v_rcp_f32
            REP4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if constexpr (OP == 5) {   // v_mad_u32_u24
            REP4(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 3u), "v"(u1));)
        } else if constexpr (OP == 6) {   // v_mad_u64_u32
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1"
                              : "+v"(w0), "+v"(w1) : "v"(u0), "v"(u1) : "vcc");)
        } else if constexpr (OP == 7) {   // v_mul_lo_u32
            REP4(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1 | 1u));)
        } else if constexpr (OP == 8) {   // v_cmp_lt_f32 -> vcc
            REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0"
                              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if constexpr (OP == 9) {   // v_cmp_lt_f32 -> sgpr pair (VOP3)
            REP4(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0"
                              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (OP == 10) {  // v_cndmask_b32 (vcc)
            REP4(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if constexpr (OP == 11) {  // v_cvt_rpi_i32_f32
            REP4(asm volatile("v_cvt_rpi_i32_f32 %0, %4\n v_cvt_rpi_i32_f32 %1, %5\n v_cvt_rpi_i32_f32 %2, %4\n v_cvt_rpi_i32_f32 %3, %5"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));)
        } else if constexpr (OP == 12) {  // v_lshl_add_u64
            REP4(asm volatile("v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %0\n v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %0"
                              : "+v"(w0), "+v"(w1));)
        } else if constexpr (OP == 13) {  // v_mov_b32
            REP4(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if constexpr (OP == 14) {  // v_mul_f32
            REP4(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (OP == 15) {  // v_div_scale_f32
            REP4(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if constexpr (OP == 16) {  // v_div_fixup_f32
            REP4(asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 17) {  // v_lshlrev_b32
            REP4(asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if constexpr (OP == 18) {  // s_nop 0
            REP16(asm volatile("s_nop 0");)
        } else if constexpr (OP == 19) {  // s_and_b64
            REP16(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]" : : : "s20", "s21", "scc");)
        } else if constexpr (OP == 20) {  // v_min3_f32
            REP4(asm volatile("v_min3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_min3_f32 %2, %2, %4, %5\n v_min3_f32 %3, %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 21) {  // v_fma_f32 with SGPR operand + v_pk_fma with op_sel (as the compiler emits)
            REP4(asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[1,0,1]"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if constexpr (OP == 22) {  // v_add_f32
            REP4(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if constexpr (OP == 23) {  // v_cmp_gt_u32 + s_and (the test chain)
            REP4(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_and_b64 s[20:21], s[20:21], vcc\n v_cmp_gt_u32 vcc, %2, %3\n s_and_b64 s[20:21], s[20:21], vcc"
                              : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc", "s20", "s21", "scc");)
        } else if constexpr (OP == 24) {  // v_cndmask_b32_e64 with an SGPR-pair mask
            REP4(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (OP == 25) {  // v_cmp then v_cndmask (the usual pair)
            REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if constexpr (OP == 26) {  // v_and_b32
            REP4(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0 | 0xffffu));)
        } else if constexpr (OP == 27) {  // v_add_u32
            REP4(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));)
        } else if constexpr (OP == 28) {  // v_max_f32
            REP4(asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if constexpr (OP == 29) {  // v_fmac_f32
            REP4(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 30) {  // v_cvt_f32_i32
            REP4(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %4\n v_cvt_f32_i32 %3, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1));)
        } else if constexpr (OP == 31) {  // v_lshl_add_u32
            REP4(asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));)
        } else if constexpr (OP == 32) {  // v_fma_f32 with one SGPR source
            REP4(asm volatile("v_fma_f32 %0, %0, s20, %4\n v_fma_f32 %1, %1, s20, %4\n v_fma_f32 %2, %2, s20, %4\n v_fma_f32 %3, %3, s20, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if constexpr (OP == 33) {  // v_sub_f32 (VOP2) and v_mul with SGPR
            REP4(asm volatile("v_sub_f32 %0, %0, %4\n v_mul_f32 %1, s20, %1\n v_sub_f32 %2, %2, %4\n v_mul_f32 %3, s20, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if constexpr (OP == 34) {  // v_cndmask_b32 writing registers nobody reads next (no dependent chain)
            REP4(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                 (float)(u0 + u1 + u2 + u3) + (float)(w0 + w1);
}

template <int OP>
double run(float *out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const int blocks = cus * 8;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    const int iters = 4000;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_mad_u32_u24", "v_mad_u64_u32",
                           "v_mul_lo_u32", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32 sgpr", "v_cndmask_b32", "v_cvt_rpi_i32_f32", "v_lshl_add_u64",
                           "v_mov_b32", "v_mul_f32", "v_div_scale_f32", "v_div_fixup_f32", "v_lshlrev_b32", "s_nop 0", "s_and_b64",
                           "v_min3_f32", "v_pk_fma_f32 op_sel", "v_add_f32", "v_cmp_gt_u32+s_and (pair)", "v_cndmask_b32_e64 sgpr", "v_cmp+v_cndmask (pair)",
                           "v_and_b32", "v_add_u32", "v_max_f32", "v_fmac_f32", "v_cvt_f32_i32", "v_lshl_add_u32", "v_fma_f32 sgpr src",
                           "v_sub_f32 / v_mul_f32 sgpr", "v_cndmask_b32 no chain"};
    printf("device %s, %d CUs, %.2f GHz nominal; 8 waves/SIMD, %d x 16 instructions per wave\n", prop.name, cus, ghz, iters);
    fflush(stdout);
    double base = 0;
    auto report = [&](int i, double ms) {
        if (i == 0) base = ms;
        const double cyc = ms * 1e-3 * ghz * 1e9 / (8.0 * iters * 16.0);   // per SIMD: 8 waves x iters x 16 instructions
        printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (nominal clock)  x%.2f of v_fma_f32\n", names[i], ms, cyc, ms / base);
        fflush(stdout);
    };
    report(0, run<0>(out, blocks, iters));   report(1, run<1>(out, blocks, iters));   report(2, run<2>(out, blocks, iters));
    report(3, run<3>(out, blocks, iters));   report(4, run<4>(out, blocks, iters));   report(5, run<5>(out, blocks, iters));
    report(6, run<6>(out, blocks, iters));   report(7, run<7>(out, blocks, iters));   report(8, run<8>(out, blocks, iters));
    report(9, run<9>(out, blocks, iters));   report(10, run<10>(out, blocks, iters)); report(11, run<11>(out, blocks, iters));
    report(12, run<12>(out, blocks, iters)); report(13, run<13>(out, blocks, iters)); report(14, run<14>(out, blocks, iters));
    report(15, run<15>(out, blocks, iters)); report(16, run<16>(out, blocks, iters)); report(17, run<17>(out, blocks, iters));
    report(18, run<18>(out, blocks, iters)); report(19, run<19>(out, blocks, iters)); report(20, run<20>(out, blocks, iters));
    report(21, run<21>(out, blocks, iters)); report(22, run<22>(out, blocks, iters)); report(23, run<23>(out, blocks, iters));
    report(24, run<24>(out, blocks, iters)); report(25, run<25>(out, blocks, iters)); report(26, run<26>(out, blocks, iters));
    report(27, run<27>(out, blocks, iters)); report(28, run<28>(out, blocks, iters)); report(29, run<29>(out, blocks, iters));
    report(30, run<30>(out, blocks, iters)); report(31, run<31>(out, blocks, iters)); report(32, run<32>(out, blocks, iters));
    report(33, run<33>(out, blocks, iters)); report(34, run<34>(out, blocks, iters));
    report(0, run<0>(out, blocks, iters));
    return 0;
}
