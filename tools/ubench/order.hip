// Does the ORDER of the raw step's instructions matter?  Same 28 ops per 4 words:
//   A: interleaved per word (and, 3x bitop3, bcnt, bitop3, bcnt) x 4         (what hipcc emits)
//   B: all 20 boolean ops of 4 words first, then the 8 bcnt back to back
//   C: boolean ops of word w+1 interleaved one-for-one with bcnt of word w
#include <hip/hip_runtime.h>
#include <cstdio>

#define R2(X) X X
#define R4(X) R2(R2(X))
#define R8(X) R2(R4(X))

#define BOOL(S, M) \
    "v_and_b32 " S ", %10, %11\n v_bitop3_b32 " S ", %12, %13, " S " bitop3:0xea\n" \
    "v_bitop3_b32 " S ", %14, %15, " S " bitop3:0xea\n v_bitop3_b32 " S ", %16, %17, " S " bitop3:0xea\n" \
    "v_bitop3_b32 " M ", " S ", %18, %19 bitop3:0x80\n"
#define CNT(A, B, S, M) "v_bcnt_u32_b32 " A ", " S ", " A "\n v_bcnt_u32_b32 " B ", " M ", " B "\n"

template <int ORDER>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x, a1 = 1;
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    unsigned q[5], t[5];
    for (int i = 0; i < 5; ++i) { q[i] = seed * (i + 3) + threadIdx.x; t[i] = seed * (i + 11) ^ threadIdx.x; }
    for (int i = 0; i < iters; ++i) {
        if constexpr (ORDER == 0) {
            R8(asm volatile(BOOL("%2","%6") CNT("%0","%1","%2","%6") BOOL("%3","%7") CNT("%0","%1","%3","%7")
                            BOOL("%4","%8") CNT("%0","%1","%4","%8") BOOL("%5","%9") CNT("%0","%1","%5","%9")
                : "+v"(a0), "+v"(a1), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                : "v"(q[0]), "v"(t[0]), "v"(q[1]), "v"(t[1]), "v"(q[2]), "v"(t[2]), "v"(q[3]), "v"(t[3]), "v"(q[4]), "v"(t[4]));)
        } else if constexpr (ORDER == 1) {
            R8(asm volatile(BOOL("%2","%6") BOOL("%3","%7") BOOL("%4","%8") BOOL("%5","%9")
                            CNT("%0","%1","%2","%6") CNT("%0","%1","%3","%7") CNT("%0","%1","%4","%8") CNT("%0","%1","%5","%9")
                : "+v"(a0), "+v"(a1), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                : "v"(q[0]), "v"(t[0]), "v"(q[1]), "v"(t[1]), "v"(q[2]), "v"(t[2]), "v"(q[3]), "v"(t[3]), "v"(q[4]), "v"(t[4]));)
        } else {
            // software-skewed: bcnt of the previous word between the boolean ops of the next
            R8(asm volatile(
                "v_and_b32 %2, %10, %11\n v_bcnt_u32_b32 %0, %5, %0\n v_bitop3_b32 %2, %12, %13, %2 bitop3:0xea\n v_bitop3_b32 %2, %14, %15, %2 bitop3:0xea\n"
                "v_bcnt_u32_b32 %1, %9, %1\n v_bitop3_b32 %2, %16, %17, %2 bitop3:0xea\n v_bitop3_b32 %6, %2, %18, %19 bitop3:0x80\n"
                "v_and_b32 %3, %10, %11\n v_bcnt_u32_b32 %0, %2, %0\n v_bitop3_b32 %3, %12, %13, %3 bitop3:0xea\n v_bitop3_b32 %3, %14, %15, %3 bitop3:0xea\n"
                "v_bcnt_u32_b32 %1, %6, %1\n v_bitop3_b32 %3, %16, %17, %3 bitop3:0xea\n v_bitop3_b32 %7, %3, %18, %19 bitop3:0x80\n"
                "v_and_b32 %4, %10, %11\n v_bcnt_u32_b32 %0, %3, %0\n v_bitop3_b32 %4, %12, %13, %4 bitop3:0xea\n v_bitop3_b32 %4, %14, %15, %4 bitop3:0xea\n"
                "v_bcnt_u32_b32 %1, %7, %1\n v_bitop3_b32 %4, %16, %17, %4 bitop3:0xea\n v_bitop3_b32 %8, %4, %18, %19 bitop3:0x80\n"
                "v_and_b32 %5, %10, %11\n v_bcnt_u32_b32 %0, %4, %0\n v_bitop3_b32 %5, %12, %13, %5 bitop3:0xea\n v_bitop3_b32 %5, %14, %15, %5 bitop3:0xea\n"
                "v_bcnt_u32_b32 %1, %8, %1\n v_bitop3_b32 %5, %16, %17, %5 bitop3:0xea\n v_bitop3_b32 %9, %5, %18, %19 bitop3:0x80\n"
                : "+v"(a0), "+v"(a1), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                : "v"(q[0]), "v"(t[0]), "v"(q[1]), "v"(t[1]), "v"(q[2]), "v"(t[2]), "v"(q[3]), "v"(t[3]), "v"(q[4]), "v"(t[4]));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ s0 ^ s1 ^ s2 ^ s3 ^ m0 ^ m1 ^ m2 ^ m3;
}

template <int ORDER>
void run(const char *name, int w)
{
    const int blocks = 256 * w;
    unsigned *out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const int iters = 1500;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<ORDER><<<blocks, 256>>>(out, 2, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<ORDER><<<blocks, 256>>>(out, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)iters * 8 * 4 * w;  // raw steps per SIMD
    printf("%-26s %d waves/SIMD: %8.3f ms  %6.2f ns per raw step per SIMD\n", name, w, ms, ms * 1e6 / steps);
    (void)hipFree(out);
}

int main()
{
    for (int w : {2, 3, 4, 8}) {
        run<0>("A interleaved per word", w);
        run<1>("B bools then 8 bcnt", w);
        run<2>("C skewed one word", w);
    }
    return 0;
}
