// What bounds the row gather of the per-slot convolution at 128 channels (512-byte rows)?  The same 46 538 x 9 row gathers
// (214 MB) under different lane -> byte mappings of the 16-byte loads, into registers or straight into LDS:
//   A  lane (i, q) reads quarter q of row i front to back: every wave-instruction touches 64 different 128-byte lines (16 B of each)
//   B  the four lanes of a row read 64 adjacent bytes: 16 lines per instruction, 64 B of each
//   C  32 lanes read one whole row: 2 rows (8 lines, all 128 B of each) per instruction
//   D  mapping C through global_load_lds_dwordx4 (no registers)
// Index patterns: random rows of the table, or rows within +-W of the query row (a spatially sorted lattice).
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_layout_probe gather_layout_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int V = 128, E = 9;

template <int MODE>
__global__ void __launch_bounds__(768) k_gather(const int* __restrict__ nbr, const float* __restrict__ values, int m, float* __restrict__ out) {
    __shared__ floatx4 s_buf[12 * 512];  // D: one 8 KB landing zone per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 192 + wave * 16;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < E; ++e) {
        if (MODE == 0 || MODE == 1) {
            const int i = lane & 15, q = lane >> 4;
            const int row = m0 + i;
            int nb = row < m ? nbr[(size_t)row * E + e] : -1;
            const float* src = values + (size_t)(nb >= 0 ? nb : 0) * V;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const floatx4 v = *reinterpret_cast<const floatx4*>(MODE == 0 ? src + q * 32 + k * 4 : src + k * 16 + q * 4);
                acc += v;
            }
        } else {
            // lane l of instruction k reads bytes (l & 31) * 16 of row 2k + (l >> 5)
            floatx4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = m0 + 2 * k + (lane >> 5);
                int nb = row < m ? nbr[(size_t)row * E + e] : -1;
                const float* src = values + (size_t)(nb >= 0 ? nb : 0) * V + (lane & 31) * 4;
                if (MODE == 2) {
                    v[k] = *reinterpret_cast<const floatx4*>(src);
                } else {
                    floatx4* dst = s_buf + wave * 512 + k * 64;  // wave-uniform base; the hardware adds lane * 16
                    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
            }
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += v[k];
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += s_buf[wave * 512 + k * 64 + lane];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

int main() {
    const int m = 46538;
    std::mt19937 rng(1);
    float* d_vals; int* d_nbr; float* d_out;
    CK(hipMalloc(&d_vals, (size_t)m * V * 4));
    CK(hipMalloc(&d_nbr, (size_t)m * E * 4));
    CK(hipMalloc(&d_out, 4096));
    std::vector<float> hv((size_t)m * V);
    for (auto& x : hv) x = (float)(rng() & 1023) / 1024.f;
    CK(hipMemcpy(d_vals, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (m + 191) / 192;
    for (int pattern = 0; pattern < 4; ++pattern) {
        std::vector<int> hn((size_t)m * E);
        const int W = pattern == 1 ? 2000 : (pattern == 2 ? 200 : 0);
        for (int r = 0; r < m; ++r)
            for (int e = 0; e < E; ++e) {
                int t;
                if (pattern == 0) t = rng() % m;
                else if (pattern == 3) t = r;
                else t = std::min(m - 1, std::max(0, r + (int)(rng() % (2 * W + 1)) - W));
                hn[(size_t)r * E + e] = t;
            }
        CK(hipMemcpy(d_nbr, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
        const char* pname[] = {"random rows", "rows within +-2000", "rows within +-200", "own row"};
        for (int mode = 0; mode < 4; ++mode) {
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(768), 0, 0, d_nbr, d_vals, m, d_out); break;
                    case 1: hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(768), 0, 0, d_nbr, d_vals, m, d_out); break;
                    case 2: hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(768), 0, 0, d_nbr, d_vals, m, d_out); break;
                    default: hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(768), 0, 0, d_nbr, d_vals, m, d_out); break;
                }
            };
            for (int i = 0; i < 5; ++i) launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms / 50 * 1e3;
            printf("%-20s mode %c  %7.1f us  %6.2f TB/s  %5.1f B/cycle/CU at 2.4 GHz\n", pname[pattern], "ABCD"[mode], us,
                   (double)m * E * V * 4 / us / 1e6, (double)m * E * V * 4 / (us * 1e-6) / 256 / 2.4e9);
        }
    }
    return 0;
}
