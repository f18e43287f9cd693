// Development probe (not part of the product): is the scalar unit of a gfx950 CU a resource of its own beside the VALUs?
// The fused Integrate kernels issue 0.4 - 0.6 SALU instructions (mask logic of the compares, exec save / restore around the
// gathers, branches) per VALU instruction.  This probe times wavefronts that run
//   OP 0: 16 VALU (v_fma_f32) per iteration
//   OP 1: 16 SALU (s_and_b64 / s_or_b64 on mask pairs) per iteration
//   OP 2: 16 VALU + 8 SALU interleaved        OP 3: 16 VALU + 16 SALU interleaved
//   OP 4: 16 VALU + 16 s_nop                  OP 5: 16 v_cmp -> sgpr pair + 16 s_and_b64 on the results
// at 1, 2, 4 and 8 wavefronts per SIMD (workgroups of 256 threads = one wavefront per SIMD, 256 x k workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o salu_rate salu_rate.hip && ./salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x

template <int OP>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const float b = seed * 0.5f + 1.0f, c = seed * 0.25f + 0.001f;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {
            REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 1) {
            REP4(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n"
                              "s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]"
                              : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if constexpr (OP == 2) {
            REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %1, %1, %4, %5\n"
                              "v_fma_f32 %2, %2, %4, %5\n s_or_b64 s[22:23], s[22:23], s[24:25]\n v_fma_f32 %3, %3, %4, %5"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "scc");)
        } else if constexpr (OP == 3) {
            REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_and_b64 s[20:21], s[20:21], s[22:23]\n v_fma_f32 %1, %1, %4, %5\n s_or_b64 s[22:23], s[22:23], s[24:25]\n"
                              "v_fma_f32 %2, %2, %4, %5\n s_and_b64 s[24:25], s[24:25], s[26:27]\n v_fma_f32 %3, %3, %4, %5\n s_or_b64 s[26:27], s[26:27], s[20:21]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if constexpr (OP == 4) {
            REP4(asm volatile("v_fma_f32 %0, %0, %4, %5\n s_nop 0\n v_fma_f32 %1, %1, %4, %5\n s_nop 0\n"
                              "v_fma_f32 %2, %2, %4, %5\n s_nop 0\n v_fma_f32 %3, %3, %4, %5\n s_nop 0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if constexpr (OP == 5) {
            REP4(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n s_and_b64 s[28:29], s[20:21], s[22:23]\n s_or_b64 s[30:31], s[28:29], s[20:21]\n"
                              "v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0\n s_and_b64 s[28:29], s[24:25], s[26:27]\n s_or_b64 s[30:31], s[28:29], s[30:31]"
                              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "scc");)
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int OP>
static void run(const char *name, int valu, int salu)
{
    const int iters = 20000;
    float *out;
    hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    const double ghz = pr.clockRate * 1e-6;
    for (int k : {1, 2, 4, 8}) {
        probe<OP><<<256 * k, 256>>>(out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<OP><<<256 * k, 256>>>(out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * ghz * 1e9;              // cycles of the launch
        const double per_simd_iter = cyc / ((double)iters * k); // cycles per iteration of one wavefront, per SIMD
        std::printf("%-34s %d waves/SIMD: %7.3f ms  %6.2f cycles per iteration per SIMD", name, k, ms, per_simd_iter);
        if (valu) std::printf("  = %.2f per VALU", per_simd_iter / valu);
        if (salu) std::printf("  = %.2f per SALU (%.2f per CU)", per_simd_iter / salu, per_simd_iter / salu / 4.0);
        std::printf("\n");
    }
    hipFree(out);
}

int main()
{
    run<0>("16 v_fma", 16, 0);
    run<1>("16 s_and/s_or", 0, 16);
    run<2>("16 v_fma + 8 salu", 16, 8);
    run<3>("16 v_fma + 16 salu", 16, 16);
    run<4>("16 v_fma + 16 s_nop", 16, 16);
    run<5>("16 v_cmp->sgpr + 16 salu", 16, 16);
    return 0;
}
