// read_rate.hip — how fast a row-major n x len byte matrix (the pack's input: 50,000 x 30,000) can be read, by the
// shape of the access.  Every variant reads each byte once and writes 16 bytes per (record, 128-byte chunk), like the
// pack's slots.
//   stream   thread i reads 16 bytes at 16 i (grid-stride): the plain streaming ceiling
//   lane_row the pack's r01-r03 shape: lane = record, a block = 256 records x ONE chunk; a lane reads its own 128-byte line
//   r8c8     wave = 8 records x 8 chunks (a row's 1 KiB contiguous across 8 lanes), block = 32 records x 8 chunks
//   r4c16    wave = 4 records x 16 chunks (2 KiB contiguous), block = 16 records x 16 chunks
//   r1c64    wave = 1 record x 64 chunks (8 KiB contiguous), block = 4 records x 64 chunks
//   rowstrU  wave = 1 record x 1 KiB per load instruction (a lane = 16 bytes), U such pieces per wave; results by record
// Build: make -C tools/ubench read_rate ; run on the GPU box: tools/ubench/read_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

__device__ __forceinline__ uint4 fold(const uint4 *p)
{
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        v[k] = p[k];
    uint4 a = v[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        a.x ^= v[k].x; a.y += v[k].y; a.z ^= v[k].z; a.w += v[k].w;
    }
    return a;
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4 *in, size_t n16, uint4 *out)
{
    uint4 a = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = in[i];
        a.x ^= v.x; a.y += v.y; a.z ^= v.z; a.w += v.w;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
}

// RW records x CW chunks per wave (RW * CW = 64); a block = 4 waves stacked along the records
template <int RW, int CW>
__global__ __launch_bounds__(256) void tile_kernel(const uint8_t *in, size_t stride, uint32_t n, uint32_t nchunks, uint32_t npad,
                                                   uint4 *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * CW + lane % CW;
    const uint32_t s = (blockIdx.y * 4u + wave) * RW + lane / CW;
    if (s >= n || c >= nchunks)
        return;
    out[(size_t)c * npad + s] = fold(reinterpret_cast<const uint4 *>(in + (size_t)s * stride + (size_t)c * 128));
}

// lane = record as in lane_row, but a thread takes K consecutive chunks of its record one after the other: a block reads
// 256 records x K x 128 contiguous bytes of each
template <int K>
__global__ __launch_bounds__(256) void lane_rowk_kernel(const uint8_t *in, size_t stride, uint32_t n, uint32_t nchunks, uint32_t npad,
                                                        uint4 *out)
{
    const uint32_t s = blockIdx.y * 256u + threadIdx.x;
    if (s >= n)
        return;
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const uint32_t c = blockIdx.x * K + k;
        if (c >= nchunks)
            break;
        out[(size_t)c * npad + s] = fold(reinterpret_cast<const uint4 *>(in + (size_t)s * stride + (size_t)c * 128));
    }
}

// a wave = one record x 1 KiB (eight chunks), a lane = 16 bytes: every load instruction covers whole lines; the eight
// 16-byte results of the wave leave as one 128-byte store (results laid out by record: out[record][chunk])
template <int UNROLL>
__global__ __launch_bounds__(256) void rowstream_kernel(const uint8_t *in, size_t stride, uint32_t n, uint32_t nchunks, uint4 *out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s = blockIdx.y * 4u + wave;
    if (s >= n)
        return;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const uint32_t g = blockIdx.x * UNROLL + u;      // group of eight chunks
        if (g * 8u >= nchunks)
            break;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g * 1024u + 16u * lane + 16u <= nchunks * 128u)
            v = *reinterpret_cast<const uint4 *>(in + (size_t)s * stride + (size_t)g * 1024 + 16u * lane);
        uint4 a = v;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a.x ^= __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z ^= __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
        }
        const uint4 mine = make_uint4(__shfl(a.x, (lane & 7u) * 8u), __shfl(a.y, (lane & 7u) * 8u), __shfl(a.z, (lane & 7u) * 8u),
                                      __shfl(a.w, (lane & 7u) * 8u));
        if (lane < 8u && g * 8u + lane < nchunks)
            out[(size_t)s * nchunks + g * 8u + lane] = mine;
    }
}

int main()
{
    const uint32_t n = 50000, len = 30000, nchunks = len / 128, npad = 50048;   // 234 whole chunks
    uint8_t *in = nullptr;
    uint4 *out = nullptr;
    CHECK(hipMalloc((void **)&in, (size_t)n * len));
    CHECK(hipMemset(in, 0x88, (size_t)n * len));
    CHECK(hipMalloc((void **)&out, (size_t)(nchunks + 1) * npad * sizeof(uint4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double bytes = (double)n * nchunks * 128;
    auto report = [&](const char *name, float ms) {
        std::printf("%-9s %8.3f ms  %6.2f TB/s read (+ %.2f GB of 16-byte results)\n", name, ms, bytes / ms / 1e9,
                    (double)n * nchunks * 16 / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        float ms = 0;
        const size_t n16 = (size_t)n * len / 16;
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 5; ++k)
            hipLaunchKernelGGL(stream_kernel, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(in), n16, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep)
            std::printf("%-9s %8.3f ms  %6.2f TB/s read\n", "stream", ms / 5, (double)n * len / (ms / 5) / 1e9);
#define RUN(NAME, RW, CW)                                                                                                       \
    CHECK(hipEventRecord(e0));                                                                                                  \
    for (int k = 0; k < 5; ++k)                                                                                                 \
        hipLaunchKernelGGL((tile_kernel<RW, CW>), dim3((nchunks + CW - 1) / CW, (n + 4 * RW - 1) / (4 * RW)), dim3(256), 0, 0, in, \
                           (size_t)len, n, nchunks, npad, out);                                                                 \
    CHECK(hipEventRecord(e1));                                                                                                  \
    CHECK(hipEventSynchronize(e1));                                                                                             \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                                    \
    if (rep)                                                                                                                    \
        report(NAME, ms / 5);
#define RUNS(NAME, U)                                                                                                           \
    CHECK(hipEventRecord(e0));                                                                                                  \
    for (int k = 0; k < 5; ++k)                                                                                                 \
        hipLaunchKernelGGL((rowstream_kernel<U>), dim3((nchunks + 8 * U - 1) / (8 * U), (n + 3) / 4), dim3(256), 0, 0, in,          \
                           (size_t)len, n, nchunks, out);                                                                       \
    CHECK(hipEventRecord(e1));                                                                                                  \
    CHECK(hipEventSynchronize(e1));                                                                                             \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                                    \
    if (rep)                                                                                                                    \
        report(NAME, ms / 5);
#define RUNK(NAME, K)                                                                                                           \
    CHECK(hipEventRecord(e0));                                                                                                  \
    for (int k = 0; k < 5; ++k)                                                                                                 \
        hipLaunchKernelGGL((lane_rowk_kernel<K>), dim3((nchunks + K - 1) / K, (n + 255) / 256), dim3(256), 0, 0, in, (size_t)len, n, \
                           nchunks, npad, out);                                                                                 \
    CHECK(hipEventRecord(e1));                                                                                                  \
    CHECK(hipEventSynchronize(e1));                                                                                             \
    CHECK(hipEventElapsedTime(&ms, e0, e1));                                                                                    \
    if (rep)                                                                                                                    \
        report(NAME, ms / 5);
        RUNK("lanerow2", 2)
        RUNK("lanerow4", 4)
        RUNK("lanerow8", 8)
        RUNS("rowstr1", 1)
        RUNS("rowstr4", 4)
        RUNS("rowstr8", 8)
        RUN("lane_row", 64, 1)
        RUN("r8c8", 8, 8)
        RUN("r4c16", 4, 16)
        RUN("r1c64", 1, 64)
    }
    return 0;
}
