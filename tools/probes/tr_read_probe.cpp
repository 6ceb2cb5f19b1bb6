// What does ds_read_b64_tr_b16 deliver?  (GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.cpp -o /tmp/tr && /tmp/tr)
// LDS holds lds[x] = x.  Every lane passes the address of 4 contiguous 16-bit elements; within a 16-lane group the lanes' 4x16
// element block is transposed: printed is what each lane receives for two address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out, int row_stride, int grp_stride) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    short* p = lds + (l >> 4) * grp_stride + ((l & 15) >> 2) * row_stride + (l & 3) * 4;
    out[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)p);
}
int main() {
    s4* d;
    hipMalloc(&d, 64 * sizeof(s4));
    for (int pass = 0; pass < 2; ++pass) {
        const int rs = pass ? 40 : 16, gs = pass ? 4 * 40 : 64;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs, gs);
        s4 h[64];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("row stride %d, group stride %d (lane: 4 values)\n", rs, gs);
        for (int l = 0; l < 64; ++l) printf("%2d: %5d %5d %5d %5d%s", l, h[l][0], h[l][1], h[l][2], h[l][3], (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
