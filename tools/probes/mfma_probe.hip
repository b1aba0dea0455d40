// Probe: sustained rate of v_mfma_f32_32x32x16_f16 / 16x16x32_f16 chains as the attention and GEMM kernels issue them
// (accumulators in VGPRs or AGPRs by build flag), alone and with v_exp / v_add fillers, at 1-3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form=1] tools/probes/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 32x32x16 two chains; 1: + 2 v_exp per MFMA; 2: + 2 exp + 4 add per MFMA; 3: 16x16x32 eight chains
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + threadIdx.x * 0.001f + j); b[j] = (_Float16)(seed * 0.5f + j); }
    float e0 = seed, e1 = seed + 1.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (MODE != 3) {
        f32x16 c0 = {0}, c1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                if (MODE >= 1) { e0 = __builtin_amdgcn_exp2f(e0 * 0.5f); e1 = __builtin_amdgcn_exp2f(e1 * 0.25f); }
                if (MODE >= 2) { s0 += e0; s1 += e1; s2 += e0 * 2.f; s3 += e1 * 3.f; }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                if (MODE >= 1) { e0 = __builtin_amdgcn_exp2f(e0 * 0.5f); e1 = __builtin_amdgcn_exp2f(e1 * 0.25f); }
                if (MODE >= 2) { s0 += e0; s1 += e1; s2 += e0 * 2.f; s3 += e1 * 3.f; }
            }
        }
        float r = 0; for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
        out[blockIdx.x * 256 + threadIdx.x] = r + e0 + e1 + s0 + s1 + s2 + s3;
    } else {
        f32x4 c[8]; for (int k = 0; k < 8; ++k) c[k] = (f32x4){0, 0, 0, 0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[k], 0, 0, 0);
        float r = 0; for (int k = 0; k < 8; ++k) for (int j = 0; j < 4; ++j) r += c[k][j];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

template <int MODE> void run(const char* name, int wgs_per_cu) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 8;                    // per wave 8 MFMAs per iteration
    const double flop = mfmas * (MODE == 3 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2);
    printf("%-34s %d WG/CU: %7.3f ms  %7.1f TFLOP/s  %6.1f ns per MFMA per SIMD\n", name, wgs_per_cu, ms, flop / ms / 1e9,
           ms * 1e6 / (mfmas / 1024.0));
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 3; ++w) {
        run<0>("32x32x16 f16, 2 chains", w);
        run<1>("32x32x16 + 2 v_exp per MFMA", w);
        run<2>("32x32x16 + 2 v_exp + 4 add per MFMA", w);
        run<3>("16x16x32 f16, 8 chains", w);
    }
    return 0;
}
