// Are workgroups with more than 64 KB of LDS isolated from workgroups of OTHER kernels running concurrently?  (GPU box:
// hipcc --offload-arch=gfx950 -O2 tools/probes/lds_coresidency_probe.cpp -o /tmp/lds && /tmp/lds)
// Kernel A (BYTES of static LDS, 256 threads, few VGPRs): fills its LDS with a per-workgroup pattern, spins, verifies, repeats; counts
// mismatches.  Kernel B (48 KB of LDS) on another stream keeps writing ITS pattern into its LDS.  If the dispatcher accounted A's
// allocation wrongly, B's workgroups would be placed over A's LDS and A would see foreign words.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BYTES>
__global__ void __launch_bounds__(256) k_a(unsigned int* bad, int rounds) {
    __shared__ unsigned int s[BYTES / 4];
    unsigned int errors = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned int pat = 0xA0000000u | (blockIdx.x << 8) | (r & 255);
        for (int i = threadIdx.x; i < BYTES / 4; i += 256) s[i] = pat ^ (unsigned int)i;
        __syncthreads();
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 300) {}  // 3 us
        __syncthreads();
        for (int i = threadIdx.x; i < BYTES / 4; i += 256) errors += (s[i] != (pat ^ (unsigned int)i));
        __syncthreads();
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
template <int BYTES>
static int run(hipStream_t sa, hipStream_t sb, unsigned int* d_bad, unsigned int* d_sink) {
    CK(hipMemset(d_bad, 0, 4));
    for (int it = 0; it < 20; ++it) {
        hipLaunchKernelGGL(k_a<BYTES>, dim3(300), dim3(256), 0, sa, d_bad, 200);
        hipLaunchKernelGGL(k_b, dim3(2000), dim3(256), 0, sb, d_sink, 30);
    }
    CK(hipDeviceSynchronize());
    unsigned int h = 0;
    CK(hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost));
    printf("A with %6d bytes of LDS beside B (48 KB): %u corrupted words\n", BYTES, h);
    return 0;
}
int main() {
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    unsigned int *d_bad, *d_sink;
    CK(hipMalloc(&d_bad, 4));
    CK(hipMalloc(&d_sink, 4));
    run<60 * 1024>(sa, sb, d_bad, d_sink);
    run<86016>(sa, sb, d_bad, d_sink);
    run<116736>(sa, sb, d_bad, d_sink);
    run<147456>(sa, sb, d_bad, d_sink);
    return 0;
}
