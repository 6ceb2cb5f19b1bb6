// Is packed fp32 arithmetic (v_pk_fma_f32) of one wave disturbed by matrix instructions of ANOTHER wave on the same SIMD?
// (GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/pk_fma_vs_mfma_probe.cpp -o /tmp/p && /tmp/p)
//
// Background (tools/probes/pair_probe.py): a segment reduce (k_csr_reduce_segments: v_pk_fma_f32 accumulation)
// running beside a convolution kernel on v_mfma_f32_16x16x32_bf16 returns rows in which the LOW element of a packed pair
// (component 0 or 2 of a float4) is wrong; beside the same convolution on v_mfma_f32_16x16x4_f32 it never does.
// Victim: every lane accumulates small integers with packed fp32 FMAs (exact in fp32) and compares with integer arithmetic.
// Aggressors: loops of one matrix instruction each, in workgroups that hold 80 KB of LDS (one per CU, one wave per SIMD), so that
// victim waves are placed beside them.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));

enum { BF16_K32 = 0, F32_K4 = 1, F16_K32 = 2, BF16_K16 = 3, VALU_ONLY = 4, SDWA = 5, SPLIT_MFMA = 6 };

template <int KIND>
__global__ void __launch_bounds__(256) aggressor(int iters, float* sink) {
    __shared__ float pad[80 * 256];  // 80 KB: one workgroup per CU
    pad[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const float seed = pad[(threadIdx.x * 7) & 255];
    floatx4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    bf16x8 ab;
    halfx8 ah;
    short4v a4;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(seed + j); ah[j] = (_Float16)(seed + j); }
    for (int j = 0; j < 4; ++j) a4[j] = (short)(threadIdx.x + j);
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == BF16_K32) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c1, 0, 0, 0);
        } else if constexpr (KIND == F32_K4) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, c1, 0, 0, 0);
        } else if constexpr (KIND == F16_K32) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ah, c1, 0, 0, 0);
        } else if constexpr (KIND == BF16_K16) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, c1, 0, 0, 0);
        } else if constexpr (KIND == SDWA) {  // the packing instruction of the bf16 split: (h0 >> 16) | h1 as one SDWA or
            unsigned int r0, r1;
            const unsigned int u0 = __float_as_uint(c0[0]) + it, u1 = __float_as_uint(c1[0]) ^ it;
            asm volatile("v_or_b32_sdwa %0, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                         "v_or_b32_sdwa %1, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD"
                         : "=&v"(r0), "=&v"(r1) : "v"(u0), "v"(u1));
            c0[0] = __uint_as_float(r0 & 0x3fffffffu);
            c1[0] = __uint_as_float(r1 & 0x3fffffffu);
        } else if constexpr (KIND == SPLIT_MFMA) {  // split + SDWA packing + bf16 matrix instruction, as in the convolution kernels
            unsigned int r0;
            const unsigned int u0 = __float_as_uint(c0[0]) + it, u1 = __float_as_uint(c1[0]) ^ it;
            asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r0) : "v"(u0), "v"(u1 & 0xffff0000u));
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 pk = {r0, r0 + 1u, r0 ^ 0x10001u, u0 & 0x7f7f7f7fu};
            const bf16x8 av = __builtin_bit_cast(bf16x8, pk);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, ab, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, av, c1, 0, 0, 0);
            c0[0] = 0.5f; c1[0] = 0.25f;
        } else {
            c0 = c0 * seed + c1;
            c1 = c1 * seed + c0;
        }
    }
    if (c0[0] + c1[1] == 12345.678f) sink[0] = c0[0];
}

enum { ALL_FORMS = 0, NO_SEL = 1, BCAST_LO = 2, BCAST_HI = 3, PK_MUL_ADD = 4 };
template <int FORM>
__global__ void __launch_bounds__(256) victim(int iters, unsigned int* errors, unsigned int* first) {
    const int lane = threadIdx.x & 63;
    unsigned int bad = 0;
    for (int blk = 0; blk < iters; ++blk) {
        float2v acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
#pragma unroll 8
        for (int it = 0; it < 64; ++it) {
            const int t = blk * 64 + it;
            const int w = (t + lane) & 7;
            const int x0 = (t * 3 + lane) & 15, x1 = (t * 5 + lane) & 15, x2 = (t * 7 + lane) & 15, x3 = (t * 11 + lane) & 15;
            const float2v xa = {(float)x0, (float)x1}, xb = {(float)x2, (float)x3};
            const int w2 = (t * 13 + lane) & 7;
            const float2v wv = {(float)w, (float)w2};  // one weight pair, broadcast by op_sel as the compiler does in the reduce
            if constexpr (FORM == ALL_FORMS) {
                asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %0, %3, %4, %0 op_sel:[0,1,0]\n"
                             "v_pk_fma_f32 %1, %2, %4, %1 op_sel:[0,1,0]"
                             : "+v"(acc0), "+v"(acc1) : "v"(xa), "v"(xb), "v"(wv));
                i0 += x0 * w + x2 * w2; i1 += x1 * w + x3 * w2; i2 += x2 * w + x0 * w2; i3 += x3 * w + x1 * w2;
            } else if constexpr (FORM == NO_SEL) {
                asm volatile("v_pk_fma_f32 %0, %2, %4, %0\n"
                             "v_pk_fma_f32 %1, %3, %4, %1"
                             : "+v"(acc0), "+v"(acc1) : "v"(xa), "v"(xb), "v"(wv));
                i0 += x0 * w; i1 += x1 * w2; i2 += x2 * w; i3 += x3 * w2;
            } else if constexpr (FORM == BCAST_LO) {
                asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"
                             : "+v"(acc0), "+v"(acc1) : "v"(xa), "v"(xb), "v"(wv));
                i0 += x0 * w; i1 += x1 * w; i2 += x2 * w; i3 += x3 * w;
            } else if constexpr (FORM == BCAST_HI) {
                asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0]\n"
                             "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0]"
                             : "+v"(acc0), "+v"(acc1) : "v"(xa), "v"(xb), "v"(wv));
                i0 += x0 * w2; i1 += x1 * w2; i2 += x2 * w2; i3 += x3 * w2;
            } else {
                float2v t0, t1;
                asm volatile("v_pk_mul_f32 %2, %4, %6 op_sel_hi:[1,0]\n"
                             "v_pk_mul_f32 %3, %5, %6 op_sel:[0,1]\n"
                             "v_pk_add_f32 %0, %0, %2\n"
                             "v_pk_add_f32 %1, %1, %3"
                             : "+v"(acc0), "+v"(acc1), "=&v"(t0), "=&v"(t1) : "v"(xa), "v"(xb), "v"(wv));
                i0 += x0 * w; i1 += x1 * w; i2 += x2 * w2; i3 += x3 * w2;
            }
        }
        const bool ok = acc0[0] == (float)i0 && acc0[1] == (float)i1 && acc1[0] == (float)i2 && acc1[1] == (float)i3;
        if (!ok) {
            if (!bad) {
                first[0] = (acc0[0] != (float)i0) | (acc0[1] != (float)i1) << 1 | (acc1[0] != (float)i2) << 2 | (acc1[1] != (float)i3) << 3;
                first[1] = (unsigned int)lane;
                first[2] = (unsigned int)blk;
                first[3] = __float_as_uint(acc0[0] != (float)i0 ? acc0[0] : acc1[0]);
            }
            ++bad;
            atomicOr(errors + 1, (acc0[0] != (float)i0) | (acc0[1] != (float)i1) << 1 | (acc1[0] != (float)i2) << 2 | (acc1[1] != (float)i3) << 3);
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int KIND, int FORM = ALL_FORMS>
static int run(const char* what, bool with_aggressor) {
    hipStream_t sa, sv;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    unsigned int *err, *first;
    float* sink;
    CK(hipMalloc(&err, 8));
    CK(hipMalloc(&first, 16));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(err, 0, 8));
    CK(hipMemset(first, 0, 16));
    for (int rep = 0; rep < 20; ++rep) {
        if (with_aggressor) hipLaunchKernelGGL((aggressor<KIND>), dim3(512), dim3(256), 0, sa, 200000, sink);
        for (int v = 0; v < 8; ++v) hipLaunchKernelGGL((victim<FORM>), dim3(2048), dim3(256), 0, sv, 100, err, first);
    }
    CK(hipDeviceSynchronize());
    unsigned int hh[2] = {0, 0}, f[4];
    CK(hipMemcpy(hh, err, 8, hipMemcpyDeviceToHost));
    const unsigned int h = hh[0];
    CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
    printf("%-58s: %9u wrong 64-step blocks of %.1e", what, h, 20.0 * 8 * 2048 * 256 * 100);
    if (h) printf("   wrong components (all) %x; first: mask %x lane %u block %u value %08x", hh[1], f[0], f[1], f[2], f[3]);
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s\n", p.name);
    run<BF16_K32>("packed fp32 FMAs alone", false);
    run<VALU_ONLY>("beside plain VALU work", true);
    run<F32_K4>("beside v_mfma_f32_16x16x4_f32", true);
    run<BF16_K16>("beside v_mfma_f32_16x16x16_bf16", true);
    run<BF16_K32>("beside v_mfma_f32_16x16x32_bf16", true);
    run<F16_K32>("beside v_mfma_f32_16x16x32_f16", true);
    run<SDWA>("beside v_or_b32_sdwa", true);
    run<SPLIT_MFMA>("beside SDWA packing + v_mfma_f32_16x16x32_bf16", true);
    printf("forms of the packed instruction, each beside v_mfma_f32_16x16x32_bf16:\n");
    run<BF16_K32, NO_SEL>("  v_pk_fma_f32 without op_sel", true);
    run<BF16_K32, BCAST_LO>("  v_pk_fma_f32 op_sel_hi:[1,0,1] (src1 low broadcast)", true);
    run<BF16_K32, BCAST_HI>("  v_pk_fma_f32 op_sel:[0,1,0] (src1 high broadcast)", true);
    run<BF16_K32, PK_MUL_ADD>("  v_pk_mul_f32 with broadcasts + v_pk_add_f32", true);
    run<BF16_K32>("packed fp32 FMAs alone again", false);
    return 0;
}
