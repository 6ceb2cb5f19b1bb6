// Does ds_read_b64_tr_b16 stay correct when workgroups of another kernel hammer LDS on the same CU?  (GPU box: hipcc
// --offload-arch=gfx950 -O2 tools/probes/tr_read_stress.cpp -o /tmp/trs && /tmp/trs)
// Kernel A: every wave writes row-major [64 rows][RS] 16-bit tiles (value = f(row, col, round)), barrier, reads them back through
// the transpose read exactly as k_conv_backward_fused_b3 does, checks every element; kernel B (another stream): plain LDS traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int RS = 40;

template <int PAD_BYTES>
__global__ void __launch_bounds__(256) k_a(unsigned int* bad, int rounds) {
    __shared__ __attribute__((aligned(16))) unsigned short s[2][64 * RS + PAD_BYTES / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    unsigned int errors = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned short* sg = s[r & 1];
        // lane (row = wave*16 + i, q) writes 8 elements (cols 8q..8q+7) as one 16-byte store
        u32x4 v;
        const int row = wave * 16 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int c0 = 8 * q + 2 * j, c1 = c0 + 1;
            const unsigned int e0 = (unsigned int)((row * 37 + c0 * 101 + r * 7 + blockIdx.x) & 0xFFFF), e1 = (unsigned int)((row * 37 + c1 * 101 + r * 7 + blockIdx.x) & 0xFFFF);
            v[j] = e0 | (e1 << 16);
        }
        *reinterpret_cast<u32x4*>(sg + row * RS + q * 8) = v;
        __syncthreads();
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row0 = 32 * st + 8 * q + 4 * h;
                    const unsigned short* base = sg + (row0 + (i >> 2)) * RS + ft * 16 + (i & 3) * 4;
                    const short4v got = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4v __attribute__((address_space(3)))*)(base));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = row0 + j, cc = ft * 16 + i;
                        const unsigned short want = (unsigned short)((rr * 37 + cc * 101 + r * 7 + blockIdx.x) & 0xFFFF);
                        errors += ((unsigned short)got[j] != want);
                    }
                }
    }
    if (errors) atomicAdd(bad, errors);
}
__global__ void __launch_bounds__(256) k_b(unsigned int* sink, int rounds) {
    __shared__ unsigned int s[48 * 1024 / 4];
    unsigned int acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) s[i] = 0xB0B0B0B0u + r;
        __syncthreads();
        for (int i = threadIdx.x; i < 48 * 1024 / 4; i += 256) acc += s[i];
        __syncthreads();
    }
    if (acc == 12345) *sink = acc;
}
template <int PAD>
static int run(bool with_b, hipStream_t sa, hipStream_t sb, unsigned int* d_bad, unsigned int* d_sink) {
    CK(hipMemset(d_bad, 0, 4));
    for (int it = 0; it < 20; ++it) {
        hipLaunchKernelGGL(k_a<PAD>, dim3(600), dim3(256), 0, sa, d_bad, 400);
        if (with_b) hipLaunchKernelGGL(k_b, dim3(2000), dim3(256), 0, sb, d_sink, 30);
    }
    CK(hipDeviceSynchronize());
    unsigned int h = 0;
    CK(hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost));
    printf("transpose reads, A LDS %6d B, %s: %u wrong elements\n", (int)(2 * (64 * RS * 2 + PAD)), with_b ? "beside kernel B" : "alone", h);
    return 0;
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    unsigned int *d_bad, *d_sink;
    CK(hipMalloc(&d_bad, 4));
    CK(hipMalloc(&d_sink, 4));
    run<0>(false, sa, sb, d_bad, d_sink);
    run<0>(true, sa, sb, d_bad, d_sink);
    run<38000>(false, sa, sb, d_bad, d_sink);
    run<38000>(true, sa, sb, d_bad, d_sink);
    return 0;
}
