// f64 VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the operations of
// the fused finalisation (division sequence, conversions, fma).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed)
{
    double a0 = threadIdx.x * 1.000001 + seed, a1 = a0 + 1.5, a2 = a0 * 0.75, a3 = a0 + 3.25;
    double a4 = a0 * 1.125, a5 = a0 + 7.0, a6 = a0 * 0.5, a7 = a0 + 9.0;
    double b = 1.0000001 + seed * 1e-9, c = 1e-9 * seed;
    unsigned u0 = threadIdx.x + 5, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, u4 = u0 * 9, u5 = u0 * 11, u6 = u0 * 13, u7 = u0 * 17;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 1) {
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if constexpr (OP == 2) {
            REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 3) {
            REP8(asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %9\n v_cvt_f64_u32 %2, %10\n v_cvt_f64_u32 %3, %11\n"
                              "v_cvt_f64_u32 %4, %12\n v_cvt_f64_u32 %5, %13\n v_cvt_f64_u32 %6, %14\n v_cvt_f64_u32 %7, %15\n"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                              : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));)
        } else if constexpr (OP == 4) {
            REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                              "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (OP == 5) {
            REP8(asm volatile("v_div_scale_f64 %0, vcc, %0, %8, %0\n v_div_scale_f64 %1, vcc, %1, %8, %1\n v_div_scale_f64 %2, vcc, %2, %8, %2\n v_div_scale_f64 %3, vcc, %3, %8, %3\n"
                              "v_div_scale_f64 %4, vcc, %4, %8, %4\n v_div_scale_f64 %5, vcc, %5, %8, %5\n v_div_scale_f64 %6, vcc, %6, %8, %6\n v_div_scale_f64 %7, vcc, %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        } else if constexpr (OP == 6) {
            REP8(asm volatile("v_div_fmas_f64 %0, %0, %8, %9\n v_div_fmas_f64 %1, %1, %8, %9\n v_div_fmas_f64 %2, %2, %8, %9\n v_div_fmas_f64 %3, %3, %8, %9\n"
                              "v_div_fmas_f64 %4, %4, %8, %9\n v_div_fmas_f64 %5, %5, %8, %9\n v_div_fmas_f64 %6, %6, %8, %9\n v_div_fmas_f64 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if constexpr (OP == 7) {
            REP8(asm volatile("v_div_fixup_f64 %0, %0, %8, %9\n v_div_fixup_f64 %1, %1, %8, %9\n v_div_fixup_f64 %2, %2, %8, %9\n v_div_fixup_f64 %3, %3, %8, %9\n"
                              "v_div_fixup_f64 %4, %4, %8, %9\n v_div_fixup_f64 %5, %5, %8, %9\n v_div_fixup_f64 %6, %6, %8, %9\n v_div_fixup_f64 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 8) {
            REP8(asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n"
                              "v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0 & 1));)
        } else if constexpr (OP == 9) {
            REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n"
                              "v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (OP == 10) {
            REP8(asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n"
                              "v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15\n"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                              : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;
    double *out;
    hipMalloc(&out, blocks * 256 * 8);
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per = ms * 1e6 / ((double)iters * 64 * waves_per_simd);
    printf("%-20s waves/SIMD=%d  %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %6.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per, ns_per * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w);
        run<1>("v_add_f64", w);
        run<2>("v_mul_f64", w);
        run<3>("v_cvt_f64_u32", w);
        run<10>("v_cvt_f64_i32", w);
        run<4>("v_rcp_f64", w);
        run<5>("v_div_scale_f64", w);
        run<6>("v_div_fmas_f64", w);
        run<7>("v_div_fixup_f64", w);
        run<8>("v_ldexp_f64", w);
        run<9>("v_sqrt_f64", w);
    }
    return 0;
}
