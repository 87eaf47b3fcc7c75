// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the ops the
// pair kernel uses.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x1234567, a2 = a0 + 77, a3 = a0 * 3;
    unsigned a4 = a0 ^ 0xdeadbeef, a5 = a0 + 1234567, a6 = a0 * 7, a7 = ~a0;
    unsigned b = seed * 3 + threadIdx.x, c = seed * 5 + 1;
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {        // v_and_b32 (VOP2)
            REP8(asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"
                              "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 1) { // v_bcnt_u32_b32 (VOP3)
            REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                              "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 2) { // v_bitop3_b32 (VOP3, 3 VGPR sources)
            REP8(asm volatile("v_bitop3_b32 %0, %8, %9, %0 bitop3:0xea\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0xea\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0xea\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0xea\n"
                              "v_bitop3_b32 %4, %8, %9, %4 bitop3:0xea\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0xea\n v_bitop3_b32 %6, %8, %9, %6 bitop3:0xea\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0xea\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 3) { // v_and_or_b32 (VOP3)
            REP8(asm volatile("v_and_or_b32 %0, %8, %9, %0\n v_and_or_b32 %1, %8, %9, %1\n v_and_or_b32 %2, %8, %9, %2\n v_and_or_b32 %3, %8, %9, %3\n"
                              "v_and_or_b32 %4, %8, %9, %4\n v_and_or_b32 %5, %8, %9, %5\n v_and_or_b32 %6, %8, %9, %6\n v_and_or_b32 %7, %8, %9, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 4) { // v_xor_b32 VOP2 with SGPR operand
            REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                              "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));)
        } else if constexpr (OP == 5) { // v_bcnt with SGPR src0
            REP8(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                              "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));)
        } else if constexpr (OP == 6) { // v_bitop3 with one SGPR source
            REP8(asm volatile("v_bitop3_b32 %0, %8, %9, %0 bitop3:0xea\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0xea\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0xea\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0xea\n"
                              "v_bitop3_b32 %4, %8, %9, %4 bitop3:0xea\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0xea\n v_bitop3_b32 %6, %8, %9, %6 bitop3:0xea\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0xea\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed), "v"(c));)
        } else if constexpr (OP == 7) { // v_add_u32 (VOP2)
            REP8(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                              "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 8) { // v_add3_u32 (VOP3)
            REP8(asm volatile("v_add3_u32 %0, %8, %9, %0\n v_add3_u32 %1, %8, %9, %1\n v_add3_u32 %2, %8, %9, %2\n v_add3_u32 %3, %8, %9, %3\n"
                              "v_add3_u32 %4, %8, %9, %4\n v_add3_u32 %5, %8, %9, %5\n v_add3_u32 %6, %8, %9, %6\n v_add3_u32 %7, %8, %9, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 9) { // v_fma_f32 (VOP3) reference full-rate
            REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 10) { // v_and_b32 VOP3 encoding (e64)
            REP8(asm volatile("v_and_b32_e64 %0, %8, %0\n v_and_b32_e64 %1, %8, %1\n v_and_b32_e64 %2, %8, %2\n v_and_b32_e64 %3, %8, %3\n"
                              "v_and_b32_e64 %4, %8, %4\n v_and_b32_e64 %5, %8, %5\n v_and_b32_e64 %6, %8, %6\n v_and_b32_e64 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (OP == 11) { // v_pk_add_u16 (packed)
            REP8(asm volatile("v_pk_add_u16 %0, %8, %0\n v_pk_add_u16 %1, %8, %1\n v_pk_add_u16 %2, %8, %2\n v_pk_add_u16 %3, %8, %3\n"
                              "v_pk_add_u16 %4, %8, %4\n v_pk_add_u16 %5, %8, %5\n v_pk_add_u16 %6, %8, %6\n v_pk_add_u16 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
double run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves/block = 1 wave per SIMD) x waves_per_simd
    unsigned *out;
    hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * waves_per_simd;  // wave-instructions per SIMD
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", name,
           waves_per_simd, ms, ns_per, ns_per * 2.4);
    hipFree(out);
    return ns_per;
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_and_b32 (VOP2)", w);
        run<10>("v_and_b32_e64 (VOP3)", w);
        run<7>("v_add_u32 (VOP2)", w);
        run<4>("v_xor_b32 sgpr,vgpr", w);
        run<9>("v_fma_f32 (VOP3)", w);
        run<1>("v_bcnt_u32_b32 v,v", w);
        run<5>("v_bcnt_u32_b32 s,v", w);
        run<2>("v_bitop3_b32 v,v,v", w);
        run<6>("v_bitop3_b32 s,v,v", w);
        run<3>("v_and_or_b32 v,v,v", w);
        run<8>("v_add3_u32 v,v,v", w);
        run<11>("v_pk_add_u16", w);
    }
    return 0;
}
