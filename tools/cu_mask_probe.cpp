// Which bits of a hipExtStreamCreateWithCUMask mask belong to which XCD?  (GPU box: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.cpp -o /tmp/p && /tmp/p)
// Launches a census kernel (XCC id per workgroup, s_getreg_b32 HW_REG_XCC_ID) on streams restricted to (a) the first 32 mask
// bits, (b) every 8th bit from bit x, and prints the histogram of XCC ids the workgroups ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_census(int* xcc, int spin) {
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID = 20, bits [3:0]
        const unsigned v = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        xcc[blockIdx.x] = (int)v;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin) {}
    }
}

static int run(const char* what, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", what, hipGetErrorString(e)); return 0; }
    const int nb = 4096;
    int* d;
    CK(hipMalloc(&d, nb * sizeof(int)));
    CK(hipMemsetAsync(d, 0xff, nb * sizeof(int), st));
    hipLaunchKernelGGL(k_census, dim3(nb), dim3(64), 0, st, d, 200);
    CK(hipStreamSynchronize(st));
    std::vector<int> h(nb);
    CK(hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost));
    int hist[16] = {0};
    for (int v : h) hist[v & 15]++;
    printf("%-28s:", what);
    for (int i = 0; i < 8; ++i) printf(" %5d", hist[i]);
    printf("\n");
    CK(hipFree(d));
    CK(hipStreamDestroy(st));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    std::vector<uint32_t> m(8, 0);
    m[0] = 0xffffffffu;
    run("bits 0..31", m);
    m.assign(8, 0);
    m[1] = 0xffffffffu;
    run("bits 32..63", m);
    for (int x = 0; x < 8; ++x) {
        m.assign(8, 0);
        for (int b = x; b < 256; b += 8) m[b / 32] |= 1u << (b % 32);
        char name[64];
        snprintf(name, sizeof name, "bits %d + 8k", x);
        run(name, m);
    }
    m.assign(8, 0xffffffffu);
    run("all 256", m);
    return 0;
}
