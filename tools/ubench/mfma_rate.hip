// Issue rate of the matrix-core instructions corr_mfma_kernel could use, on registers only: one wave per SIMD (grid =
// CUs x 4 blocks of 64 threads), four independent accumulators, kLoop x 4 instructions per wave.
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i_acc __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kLoop = 4096;

template <int KIND>
__global__ void rate(int *out, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed + 4, seed + 5, seed + 6, seed + 7};
    if constexpr (KIND == 0) {   // v_mfma_i32_32x32x32_i8
        v16i acc[4] = {};
        for (int k = 0; k < kLoop; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
        out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if constexpr (KIND == 1) {   // v_mfma_i32_16x16x64_i8
        v4i_acc acc[4] = {};
        for (int k = 0; k < kLoop; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[t], 0, 0, 0);
        out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {   // v_mfma_f32_32x32x16_bf16
        v8s ab = {(short)seed, 1, 2, 3, 4, 5, 6, 7}, bb = {7, 6, 5, 4, 3, 2, 1, (short)seed};
        v16f acc[4] = {};
        for (int k = 0; k < kLoop; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
        out[blockIdx.x * 64 + threadIdx.x] = (int)(acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3]);
    }
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int *out = nullptr;
    hipMalloc(&out, (size_t)cus * 4 * 64 * sizeof(int));
    const char *names[3] = {"v_mfma_i32_32x32x32_i8", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_32x32x16_bf16"};
    for (int kind = 0; kind < 3; ++kind) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (kind == 0)
                hipLaunchKernelGGL(rate<0>, dim3(cus * 4), dim3(64), 0, 0, out, rep);
            else if (kind == 1)
                hipLaunchKernelGGL(rate<1>, dim3(cus * 4), dim3(64), 0, 0, out, rep);
            else
                hipLaunchKernelGGL(rate<2>, dim3(cus * 4), dim3(64), 0, 0, out, rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2)
                std::printf("%-28s %8.3f ms for %d per wave, one wave per SIMD: %.1f ns = %.0f cycles at 2.4 GHz each\n", names[kind], ms,
                            kLoop * 4, ms * 1e6 / (kLoop * 4), ms * 1e6 / (kLoop * 4) * 2.4);
        }
    }
    return 0;
}
