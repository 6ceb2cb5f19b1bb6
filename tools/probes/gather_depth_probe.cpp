// Is the coalesced LDS-DMA row gather (mode D of gather_layout_probe.cpp: 7 TB/s at 46 538 x 9 rows of 512 B) bound by latency x bytes
// in flight, or by a rate?  The same gathers with W waves per workgroup and DEPTH slots of 8 KB per wave in flight.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_depth_probe gather_depth_probe.cpp ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int V = 128, E = 9;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned int lds_dst) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int W, int DEPTH, int ROWS>  // ROWS rows per wave and slot (16: 8 KB, 8 pieces)
__global__ void __launch_bounds__(64 * W) k_gather(const int* __restrict__ nbr, const float* __restrict__ values, int m, float* __restrict__ out) {
    __shared__ floatx4 s_buf[W * DEPTH * ROWS * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = (blockIdx.x * W + wave) * ROWS;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)(char*)s_buf) + wave * DEPTH * ROWS * 512;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int e) {
#pragma unroll
        for (int k = 0; k < ROWS / 2; ++k) {
            const int row = m0 + 2 * k + (lane >> 5);
            int nb = row < m ? nbr[(size_t)row * E + e] : 0;
            glds16(values + (size_t)nb * V + (lane & 31) * 4, lds0 + (e % DEPTH) * ROWS * 512 + k * 1024);
        }
    };
    for (int e = 0; e < DEPTH - 1; ++e) issue(e);
    for (int e = 0; e < E; ++e) {
        if (e + DEPTH - 1 < E) issue(e + DEPTH - 1);
        // wait until slot e has landed: at most (DEPTH - 1) * ROWS / 2 younger pieces may stay in flight
        const int younger = (E - 1 - e < DEPTH - 1 ? E - 1 - e : DEPTH - 1) * (ROWS / 2);
        if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < ROWS / 2; ++k) acc += s_buf[(wave * DEPTH + (e % DEPTH)) * ROWS * 32 + k * 64 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x] = acc[0];
}

int main() {
    const int m = 46538;
    std::mt19937 rng(1);
    float* d_vals; int* d_nbr; float* d_out;
    CK(hipMalloc(&d_vals, (size_t)m * V * 4)); CK(hipMalloc(&d_nbr, (size_t)m * E * 4)); CK(hipMalloc(&d_out, 65536));
    std::vector<float> hv((size_t)m * V, 0.5f);
    CK(hipMemcpy(d_vals, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> hn((size_t)m * E);
    for (auto& x : hn) x = rng() % m;
    CK(hipMemcpy(d_nbr, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kernel, int waves, int rows) {
        const int grid = (m + waves * rows - 1) / (waves * rows);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), 0, 0, d_nbr, d_vals, m, d_out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), 0, 0, d_nbr, d_vals, m, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms / 50 * 1e3;
        printf("%-44s grid %4d  %7.1f us  %6.2f TB/s\n", name, grid, us, (double)m * E * V * 4 / us / 1e6);
    };
    run("12 waves x 16 rows, 1 slot in flight", k_gather<12, 1, 16>, 12, 16);
    run("8 waves x 16 rows, 1 slot in flight", k_gather<8, 1, 16>, 8, 16);
    run("8 waves x 16 rows, 2 slots in flight", k_gather<8, 2, 16>, 8, 16);
    run("4 waves x 16 rows, 1 slot in flight", k_gather<4, 1, 16>, 4, 16);
    run("4 waves x 16 rows, 2 slots in flight", k_gather<4, 2, 16>, 4, 16);
    run("4 waves x 16 rows, 3 slots in flight", k_gather<4, 3, 16>, 4, 16);
    run("12 waves x 8 rows, 1 slot in flight", k_gather<12, 1, 8>, 12, 8);
    run("12 waves x 8 rows, 2 slots in flight", k_gather<12, 2, 8>, 12, 8);
    run("12 waves x 8 rows, 3 slots in flight", k_gather<12, 3, 8>, 12, 8);
    run("16 waves x 8 rows, 2 slots in flight", k_gather<16, 2, 8>, 16, 8);
    return 0;
}
