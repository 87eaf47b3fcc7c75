// Does a long straight-line VOP3 loop body run slower than a short one (instruction fetch)?
// And does the mix the pair kernel uses (v_and / v_bitop3 / v_bcnt, raw step) reach its issue roof?
#include <hip/hip_runtime.h>
#include <cstdio>

#define R2(X) X X
#define R4(X) R2(R2(X))
#define R8(X) R2(R4(X))
#define R16(X) R2(R8(X))
#define R32(X) R2(R16(X))

// one "raw step": 1 and + 4 bitop3 + 2 bcnt, all VGPR
#define STEP(A, B, T)                                                                  \
    "v_and_b32 " T ", %16, %17\n v_bitop3_b32 " T ", %18, %19, " T " bitop3:0xea\n"     \
    "v_bitop3_b32 " T ", %20, %21, " T " bitop3:0xea\n v_bitop3_b32 " T ", %22, %23, " T " bitop3:0xea\n" \
    "v_bcnt_u32_b32 " A ", " T ", " A "\n v_bitop3_b32 " T ", " T ", %24, %25 bitop3:0x80\n v_bcnt_u32_b32 " B ", " T ", " B "\n"

#define BODY                                                                           \
    asm volatile(STEP("%0", "%1", "%26") STEP("%2", "%3", "%27") STEP("%4", "%5", "%26") STEP("%6", "%7", "%27")   \
                 STEP("%8", "%9", "%26") STEP("%10", "%11", "%27") STEP("%12", "%13", "%26") STEP("%14", "%15", "%27") \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(q[0]), "v"(t[0]), "v"(q[1]), "v"(t[1]), "v"(q[2]), "v"(t[2]), "v"(q[3]), "v"(t[3]), "v"(q[4]), "v"(t[4]), \
                   "v"(tmp0), "v"(tmp1));

template <int LONG>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a[16], q[5], t[5];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x + i;
    for (int i = 0; i < 5; ++i) { q[i] = seed * (i + 3) + threadIdx.x; t[i] = seed * (i + 11) ^ threadIdx.x; }
    unsigned tmp0 = 0, tmp1 = 0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (LONG) { R32(BODY) }   // 32 x 56 instr = 1792 instr ~ 14 KB
        else { BODY }                        // 56 instr ~ 440 B
    }
    unsigned r = 0;
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int LONG>
void run(int w)
{
    const int blocks = 256 * w;
    unsigned *out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const int iters = LONG ? 200 : 6400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<LONG><<<blocks, 256>>>(out, 2, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<LONG><<<blocks, 256>>>(out, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)iters * (LONG ? 32 : 1) * 8 * w;  // raw steps (7 instr) per SIMD
    const double ns = ms * 1e6 / steps_per_simd;
    printf("%-6s body, %d waves/SIMD: %8.3f ms, %6.2f ns per raw step per SIMD (roof 18 cyc = %.2f ns @2.4GHz)\n",
           LONG ? "long" : "short", w, ms, ns, 18 / 2.4);
    (void)hipFree(out);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8}) { run<0>(w); run<1>(w); }
    return 0;
}
