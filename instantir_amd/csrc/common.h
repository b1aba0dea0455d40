// Shared device helpers for the gfx950 kernels (CDNA4: wave64, MFMA, 160 KiB LDS/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// element-type traits of the kernels that exist in an fp16 and a bf16 build (the VAE runs bf16: SDXL's VAE overflows fp16)
template <typename E> struct ET;
template <> struct ET<f16> {
    using x4 = f16x4; using x8 = f16x8;
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct ET<bf16> {
    using x4 = bf16x4; using x8 = bf16x8;
    static __device__ __forceinline__ f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
#define IIR_DT_F16 0
#define IIR_DT_BF16 1

// 8 halves -> 8 fp8-E4M3 (OCP) bytes: round to nearest even, saturating, element order kept (v_cvt_scalef32_pk_fp8_f16, scale 1).
// The SAME conversion the fp8-weight GEMM applies to its fp16 activation fragments, so a producer that stores fp8 with this
// helper hands the all-fp8 GEMM exactly the operand bytes the fp16-activation form would have formed in registers.
__device__ __forceinline__ long iir_fp8x8(f16x8 v) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){v[0], v[1]}, 1.0f, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){v[2], v[3]}, 1.0f, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, (f16x2){v[4], v[5]}, 1.0f, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, (f16x2){v[6], v[7]}, 1.0f, true);
    return (long)(unsigned)__builtin_bit_cast(int, lo) | ((long)__builtin_bit_cast(int, hi) << 32);
}
__device__ __forceinline__ int iir_fp8x4(f16x4 v) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 lo = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){v[0], v[1]}, 1.0f, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){v[2], v[3]}, 1.0f, true);
    return __builtin_bit_cast(int, lo);
}

#define IIR_OK 0
#define IIR_EINVAL (-1)
#define IIR_ELAUNCH (-2)

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  LDS destination is
// wave-uniform base + lane*16; the global source address is per lane.
//
// Round 3: issued through inline asm, invisible to hipcc.  With the builtin in a loop, hipcc's wait insertion (ROCm 7.2) treats
// the pending LDS-DMA as an LDS event of unknown order and degrades EVERY `s_waitcnt lgkmcnt(N)` in that loop to `lgkmcnt(0)`:
// fragment reads issued a sub-step ahead (to run under the MFMAs) were waited for at once, one exposed LDS round trip per MFMA
// group in every GEMM / conv / attention main loop (tools/probes: the same loop gets `lgkmcnt(10)/(9)/(8)/(7)` without the
// builtin, one `lgkmcnt(0)` with it).  The kernels order the DMA themselves anyway -- counted `s_waitcnt vmcnt(N)` + barrier in
// asm before any read of a staged buffer -- so nothing relied on the compiler knowing about it.  M0 (the LDS destination base)
// is written and restored inside the statement (cdna_hip_programming.md 5.7: M0 is compiler-reserved).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef IIR_GLDS_BUILTIN
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
#else
    // (the low 32 bits of a generic pointer into LDS are the LDS byte address: no address-space cast with its null check)
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#endif
}

// x * sigmoid(x) with one v_exp and one v_rcp (1 ulp) instead of the IEEE division sequence: the GroupNorm + SiLU pass
// evaluates it for every element of every resnet input.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-erf GELU (F.gelu default, the GEGLU gate of module/min_sdxl.py:502-528).  erf by Abramowitz-Stegun 7.1.26
// (|error| < 1.5e-7, branch-free: one v_rcp, one v_exp, five FMAs) instead of libm's piecewise erff: the GEGLU epilogue
// evaluates it 10240 times per 128x160 tile.  Against torch's fp64 GELU the fp32 result is within 5e-7 absolute.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float z = x * 0.70710678118654752f, a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
    const float erf_abs = fmaf(-p * t, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Per-launch timing (include/instantir_hip.h: iir_timing_arm): when a start/stop event pair is armed on this thread,
// the next MFMA-kernel launch goes through hipExtLaunchKernelGGL, which stamps the events with the dispatch's own
// begin / end timestamps -- the kernel's duration as the profiler sees it, with no event-record barrier in between.
extern thread_local hipEvent_t iir_armed_start, iir_armed_stop;

template <typename K, typename... Args>
static inline void iir_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
    if (iir_armed_start) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, iir_armed_start, iir_armed_stop, 0, args...);
        iir_armed_start = iir_armed_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
    }
}

static inline int iir_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IIR_OK : IIR_ELAUNCH;
}
