// HBM write-rate microbenchmark for gfx950: what a kernel that ONLY writes f64 results can reach, for the store
// patterns the consensus pair kernel could use.  Build: hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate
//   ./store_rate [n_records=50000] [rows per tile=32] [panel-major tile order=1]
// Patterns (all write n(n-1)/2 doubles = the canonical i<j triangle, or the same byte count linearly):
//   0 linear, 16 B per lane, 16-byte aligned, grid-stride
//   1 linear, 16 B per lane, base + 8 B (8-byte aligned only, like a triangle row start)
//   2 linear, nontemporal 16 B
//   3 triangle tiles as consensus_pair_kernel writes them: block = ROWS rows x one panel of 2,048 columns,
//     256 threads, thread t writes columns 2t+512j (j = 0..3) of a row, rows one after the other
//   4 like 3 with 8 B per lane (columns t + 256 j)
//   5 like 3 but the 4 waves take different rows (wave w: rows w, w+4, ...; lane l writes columns 2l + 128 j, j = 0..15)
//   6 like 3, nontemporal
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct __attribute__((packed, aligned(8))) D2 { double a, b; };

template <int MODE>
__global__ __launch_bounds__(256) void linear(double *out, size_t n2)
{
    // n2 = number of 16-byte pairs
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        D2 v{(double)i, 1.0};
        if constexpr (MODE == 2) {
            __builtin_nontemporal_store(v.a, out + 2 * i);
            __builtin_nontemporal_store(v.b, out + 2 * i + 1);
        } else {
            *reinterpret_cast<D2 *>(out + 2 * i) = v;
        }
    }
}

struct Tile { uint32_t i0, i1, panel; };

__device__ __forceinline__ uint64_t tri_row_start(uint32_t n, uint32_t q) { return (uint64_t)q * (2ull * n - q - 1) / 2; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void tiles(double *out, const Tile *tl, uint32_t n)
{
    const Tile t = tl[blockIdx.x];
    const uint32_t panel0 = t.panel * 2048u, pcols = min(2048u, n - panel0);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if constexpr (MODE == 5) {
        for (uint32_t q = t.i0 + wave; q < t.i1; q += 4) {
            const uint64_t row_at = tri_row_start(n, q) - (uint64_t)(q + 1);
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const uint32_t k = 2 * lane + 128 * j, c = panel0 + k;
                if (k >= pcols) break;
                const bool l0 = c > q, l1 = k + 1 < pcols && c + 1 > q;
                if (l0 && l1) { D2 v{(double)c, (double)q}; *reinterpret_cast<D2 *>(out + row_at + c) = v; }
                else if (l0) out[row_at + c] = (double)c;
                else if (l1) out[row_at + c + 1] = (double)q;
            }
        }
        return;
    }
    if constexpr (MODE == 8 || MODE == 9) {
        // wave per row; lanes mapped by ABSOLUTE output index: 16-byte aligned pairs (8) / 128-byte aligned wave segments (9)
        for (uint32_t q = t.i0 + wave; q < t.i1; q += 4) {
            const uint64_t row_at = tri_row_start(n, q) - (uint64_t)(q + 1);
            const uint64_t first = row_at + panel0;
            const uint64_t base = MODE == 8 ? (first & ~1ull) : (first & ~15ull);
            const uint64_t lo = row_at + max(panel0, q + 1), hi = row_at + panel0 + pcols;   // live absolute range
            for (uint64_t a = base + 2 * lane; a < hi; a += 128) {
                const bool l0 = a >= lo, l1 = a + 1 >= lo && a + 1 < hi;
                if (l0 && l1) { D2 v{(double)a, (double)q}; *reinterpret_cast<D2 *>(out + a) = v; }
                else if (l0) out[a] = (double)a;
                else if (l1) out[a + 1] = (double)q;
            }
        }
        return;
    }
    if constexpr (MODE == 10 || MODE == 11) {
        // all 4 waves on the same row, each wave one contiguous QUARTER of the panel (4 consecutive 1-KB stores);
        // 11: the quarters are cut at 128-byte boundaries of the absolute output address
        for (uint32_t q = t.i0; q < t.i1; ++q) {
            const uint64_t row_at = tri_row_start(n, q) - (uint64_t)(q + 1);
            const uint64_t first = row_at + panel0;
            const uint64_t base = MODE == 10 ? first : (first & ~15ull);
            const uint64_t lo = row_at + max(panel0, q + 1), hi = row_at + panel0 + pcols;
            const uint64_t w0 = base + 512ull * wave, w1 = wave == 3 ? hi : w0 + 512;
            for (uint64_t a = w0 + 2 * lane; a < w1; a += 128) {
                const bool l0 = a >= lo && a < hi, l1 = a + 1 >= lo && a + 1 < hi;
                if (l0 && l1) { D2 v{(double)a, (double)q}; *reinterpret_cast<D2 *>(out + a) = v; }
                else if (l0) out[a] = (double)a;
                else if (l1) out[a + 1] = (double)q;
            }
        }
        return;
    }
    if constexpr (MODE == 7) {
        // rows one after the other over the whole block (like 3) but 16-byte aligned pairs by absolute index
        for (uint32_t q = t.i0; q < t.i1; ++q) {
            const uint64_t row_at = tri_row_start(n, q) - (uint64_t)(q + 1);
            const uint64_t first = row_at + panel0, base = first & ~1ull;
            const uint64_t lo = row_at + max(panel0, q + 1), hi = row_at + panel0 + pcols;
            for (uint64_t a = base + 2 * tid; a < hi; a += 512) {
                const bool l0 = a >= lo, l1 = a + 1 >= lo && a + 1 < hi;
                if (l0 && l1) { D2 v{(double)a, (double)q}; *reinterpret_cast<D2 *>(out + a) = v; }
                else if (l0) out[a] = (double)a;
                else if (l1) out[a + 1] = (double)q;
            }
        }
        return;
    }
    for (uint32_t q = t.i0; q < t.i1; ++q) {
        const uint64_t row_at = tri_row_start(n, q) - (uint64_t)(q + 1);
        if constexpr (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t k = tid + 256 * j, c = panel0 + k;
                if (k < pcols && c > q) out[row_at + c] = (double)c;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t k = 2 * tid + 512 * j, c = panel0 + k;
                if (k >= pcols) continue;
                const bool l0 = c > q, l1 = k + 1 < pcols && c + 1 > q;
                if (l0 && l1) {
                    D2 v{(double)c, (double)q};
                    if constexpr (MODE == 6) {
                        __builtin_nontemporal_store(v.a, out + row_at + c);
                        __builtin_nontemporal_store(v.b, out + row_at + c + 1);
                    } else
                        *reinterpret_cast<D2 *>(out + row_at + c) = v;
                } else if (l0) out[row_at + c] = (double)c;
                else if (l1) out[row_at + c + 1] = (double)q;
            }
        }
    }
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 50000;
    const uint32_t rows = argc > 2 ? atoi(argv[2]) : 32;
    const uint64_t pairs = (uint64_t)n * (n - 1) / 2;
    double *out;
    CK(hipMalloc(&out, (pairs + 4) * 8));
    // tile list: panel-major like the product's schedule (all row blocks of a panel adjacent)
    std::vector<Tile> tl;
    const uint32_t npanels = (n + 2047) / 2048;
    const bool panel_major = argc > 3 ? atoi(argv[3]) != 0 : true;   // the product's order
    if (panel_major) {
        for (uint32_t p = 0; p < npanels; ++p) {
            const uint32_t last = std::min<uint32_t>(n, (p + 1) * 2048u) - 1;
            for (uint32_t i0 = 0; i0 < last; i0 += rows)
                tl.push_back({i0, std::min(last, i0 + rows), p});
        }
    } else {
        for (uint32_t i0 = 0; i0 < n - 1; i0 += rows)
            for (uint32_t p = i0 / 2048; p < npanels; ++p)
                tl.push_back({i0, (uint32_t)std::min<uint64_t>(i0 + rows, n - 1), p});
    }
    Tile *dtl;
    CK(hipMalloc(&dtl, tl.size() * sizeof(Tile)));
    CK(hipMemcpy(dtl, tl.data(), tl.size() * sizeof(Tile), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# n = %u, %llu pairs = %.2f GB, %zu tiles of %u rows\n", n, (unsigned long long)pairs, pairs * 8e-9, tl.size(), rows);
    for (int off : {1, 2, 4, 8}) {   // linear, base shifted by `off` doubles
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            linear<1><<<256 * 16, 256>>>(out + off, (pairs - 16) / 2);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("linear +%d doubles  %8.3f ms  %6.2f TB/s\n", off, best, pairs * 8e-9 / best);
    }
    for (int mode = 0; mode <= 11; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            switch (mode) {
            case 0: linear<0><<<256 * 16, 256>>>(out, pairs / 2); break;
            case 1: linear<1><<<256 * 16, 256>>>(out + 1, pairs / 2); break;
            case 2: linear<2><<<256 * 16, 256>>>(out, pairs / 2); break;
            case 3: tiles<3><<<tl.size(), 256>>>(out, dtl, n); break;
            case 4: tiles<4><<<tl.size(), 256>>>(out, dtl, n); break;
            case 5: tiles<5><<<tl.size(), 256>>>(out, dtl, n); break;
            case 6: tiles<6><<<tl.size(), 256>>>(out, dtl, n); break;
            case 7: tiles<7><<<tl.size(), 256>>>(out, dtl, n); break;
            case 8: tiles<8><<<tl.size(), 256>>>(out, dtl, n); break;
            case 9: tiles<9><<<tl.size(), 256>>>(out, dtl, n); break;
            case 10: tiles<10><<<tl.size(), 256>>>(out, dtl, n); break;
            case 11: tiles<11><<<tl.size(), 256>>>(out, dtl, n); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("mode %d  %8.3f ms  %6.2f TB/s\n", mode, best, pairs * 8e-9 / best);
    }
    return 0;
}
