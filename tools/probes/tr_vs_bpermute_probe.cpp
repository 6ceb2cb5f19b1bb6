// Does ds_read_b64_tr_b16 (LDS transpose read) in one wave disturb ds_bpermute_b32 (__shfl_*) in ANOTHER wave of the same CU?
// (GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_vs_bpermute_probe.cpp -o /tmp/p && /tmp/p)
//
// Background: with the bf16x3 fused convolution backward launched with one or two sub-tiles per workgroup — so that
// workgroups of other kernels fit beside it on a CU — a concurrently running segment reduce (k_csr_reduce_segments, whose segmented
// suffix reduction is four __shfl_down per step, one per float4 component) produced rows with ONE wrong component per float4, only
// for vertices heavy enough to use the shuffles.  This probe isolates the pair: kernel `aggressor` loops over transpose reads (or,
// as the control, over ordinary 8-byte LDS reads), kernel `victim` loops over runtime-delta __shfl_down and checks every result.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef short short4v __attribute__((ext_vector_type(4)));

template <bool TR, int LDS_KB>
__global__ void __launch_bounds__(256) aggressor(int iters, unsigned int* sink) {
    __shared__ __attribute__((aligned(16))) unsigned short s[LDS_KB * 512];
    for (int k = threadIdx.x; k < LDS_KB * 512; k += 256) s[k] = (unsigned short)(k * 7 + 1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    unsigned int acc = 0;
    constexpr int TILES = (LDS_KB * 512 - 3 * 2560 - 200) / 640;
    for (int it = 0; it < iters; ++it) {
        unsigned short* tile = s + ((it * 4 + wave) % TILES) * 640;
        if constexpr (TR) {
            // staging as in the fused backward: three 16-byte row pieces per lane, one barrier, then 12 transpose reads in flight
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            unsigned short* dst = tile + (lane & 15) * 40 + q * 8;
            const u32x4 pv = {acc, acc + 1u, acc + 2u, (unsigned int)it};
            *reinterpret_cast<u32x4*>(dst) = pv;
            *reinterpret_cast<u32x4*>(dst + 2560) = pv;
            *reinterpret_cast<u32x4*>(dst + 5120) = pv;
            __syncthreads();
            short4v r[12];
#pragma unroll
            for (int k = 0; k < 12; ++k)
                r[k] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (short4v __attribute__((address_space(3)))*)(tile + (k % 3) * 2560 + ((k / 3) * 4 + (i >> 2)) * 40 + (i & 3) * 4 + (q & 1) * 16));
            // and feed them to the bf16 matrix cores as the B operand (two reads = one 8-element fragment), as the filter gradient does
            typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
            typedef float floatx4 __attribute__((ext_vector_type(4)));
            typedef short short8v __attribute__((ext_vector_type(8)));
            floatx4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 12; k += 2) {
                const short8v b8 = {r[k][0], r[k][1], r[k][2], r[k][3], r[k + 1][0], r[k + 1][1], r[k + 1][2], r[k + 1][3]};
                const bf16x8 bb = __builtin_bit_cast(bf16x8, b8);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, bb, c, 0, 0, 0);
            }
            acc += (unsigned int)__builtin_bit_cast(unsigned int, c[0]) & 0xffu;
        } else {
            const unsigned short* base = tile + (8 * q + (i >> 2)) * 40 + (i & 3) * 4;
            unsigned long long a[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) a[k] = *reinterpret_cast<const unsigned long long __attribute__((address_space(3)))*>(
                (const unsigned long long __attribute__((address_space(3)))*)(base + k * 160));
#pragma unroll
            for (int k = 0; k < 6; ++k) acc += (unsigned int)a[k] + (unsigned int)(a[k] >> 32);
            __syncthreads();
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// every lane holds f(lane, it); after __shfl_down(x, delta) lane l must hold f(l + delta, it) (or its own value past the end)
__global__ void __launch_bounds__(256) victim(int iters, int delta, unsigned int* errors, unsigned int* first) {
    const int lane = threadIdx.x & 63;
    unsigned int bad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned int mine[6], got[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) mine[k] = (unsigned int)(lane * 2654435761u) ^ (unsigned int)(it * 40503u + blockIdx.x + k * 0x9e3779b9u);
#pragma unroll
        for (int k = 0; k < 6; ++k) got[k] = (unsigned int)__shfl_down((int)mine[k], delta, 64);  // six ds_bpermute_b32 in flight, as in the reduce
        const int src = lane + delta < 64 ? lane + delta : lane;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned int want = (unsigned int)(src * 2654435761u) ^ (unsigned int)(it * 40503u + blockIdx.x + k * 0x9e3779b9u);
            if (got[k] != want) {
                if (!bad) { first[0] = got[k]; first[1] = want; first[2] = (unsigned int)lane; first[3] = (unsigned int)(it * 8 + k); }
                ++bad;
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <bool TR, int LDS_KB>
static int run(const char* what, bool with_aggressor) {
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    unsigned int *err, *first, *sink;
    CK(hipMalloc(&err, 4));
    CK(hipMalloc(&first, 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(err, 0, 4));
    CK(hipMemset(first, 0, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sv));
    for (int rep = 0; rep < 20; ++rep) {
        if (with_aggressor) hipLaunchKernelGGL((aggressor<TR, LDS_KB>), dim3(512), dim3(256), 0, sa, 40000, sink);
        for (int v = 0; v < 8; ++v) hipLaunchKernelGGL(victim, dim3(2048), dim3(256), 0, sv, 2000, 8 << (v & 1), err, first);
    }
    CK(hipEventRecord(e1, sv));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned int h = 0, f[4];
    CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
    printf("%-64s: %10u wrong shuffles of %.2e (%.0f ms)", what, h, 20.0 * 8 * 2048 * 256 * 2000 * 6, ms);
    if (h) printf("   e.g. got %08x want %08x lane %u it*8+k %u", f[0], f[1], f[2], f[3]);
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s\n", p.name);
    run<true, 80>("victim alone", false);
    run<false, 80>("victim beside ordinary 8-byte LDS reads (80 KB workgroups)", true);
    run<true, 80>("victim beside ds_read_b64_tr_b16 (80 KB workgroups)", true);
    run<true, 32>("victim beside ds_read_b64_tr_b16 (32 KB workgroups)", true);
    run<true, 80>("victim alone again", false);
    return 0;
}
