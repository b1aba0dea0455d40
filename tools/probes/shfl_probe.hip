// Probe for DESIGN.md section 5.8: the wave-butterfly GroupNorm finalize that gave run-to-run different statistics beside a
// concurrent conv.  Variants of the same reduction over identical input, each launched REP times on one stream while a second
// stream runs the library's conv kernel; outputs compared bit for bit with the first launch.
//   V0  __shfl_xor butterfly, Chan merge with its early `return` when the incoming count is 0   (the form that failed)
//   V1  same butterfly, branch-free merge
//   V2  butterfly through LDS instead of ds_bpermute, Chan merge with the early return
//   V3  V0 with unconditional (clamped) loads instead of one predicated load per basic block
//   V4  V0 with an explicit `s_waitcnt vmcnt(0)` behind the predicated loads
//   V6  V0 with a sentinel (1e6) instead of 0 as the value of a lane whose predicate is false (all predicates are true here)
//   V7  V0 with v_rcp_f32 in the merges instead of the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup)
//   V5  V0 with the four load-address register pairs kept live to the end of the kernel (the allocator cannot reuse them)
//   hipcc --offload-arch=gfx950 -O3 -I include tools/probes/shfl_probe.hip -o tools/probes/bin/shfl_probe -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "instantir_hip.h"

__device__ __forceinline__ void merge_ret(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb <= 0.f) return;
    const float tot = n + nb, d = mb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}
__device__ __forceinline__ void merge_nobr(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float tot = n + nb, d = mb - mean, inv = 1.0f / fmaxf(tot, 1e-30f);
    mean += d * (nb * inv);
    m2 += m2b + d * d * (n * nb * inv);
    n = tot;
}

__device__ __forceinline__ void merge_rcp(float& n, float& mean, float& m2, float nb, float mb, float m2b) {   // no IEEE division sequence
    if (nb <= 0.f) return;
    const float tot = n + nb, d = mb - mean, inv = __builtin_amdgcn_rcpf(tot);
    mean += d * (nb * inv);
    m2 += m2b + d * d * (n * nb * inv);
    n = tot;
}

template <int V>
__global__ __launch_bounds__(256) void fin(const float* part, int nslab, int G, float cnt, float* stat, int RG) {
    __shared__ float xs[3][256];
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= RG) return;
    const int r = w / G, gi = w - r * G;
    const float* base = part + ((long)r * nslab * G + gi) * 2;
    float2 a[4];
    const float2* ptr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sl = lane + k * 64;
        ptr[k] = (const float2*)(base + (long)sl * G * 2);
        if (V == 3) a[k] = *(const float2*)(base + (long)min(sl, nslab - 1) * G * 2);          // unconditional, clamped
        else if (V == 6) a[k] = sl < nslab ? *ptr[k] : make_float2(1.0e6f, 0.f);      // sentinel: a lane that skips its load shows as a huge mean
        else a[k] = sl < nslab ? *ptr[k] : make_float2(0.f, 0.f);
    }
    if (V == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // explicit wait behind the predicated loads
    float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (lane + k * 64 < nslab) { if (V == 1) merge_nobr(n, mean, m2, cnt, a[k].x, a[k].y); else if (V == 7) merge_rcp(n, mean, m2, cnt, a[k].x, a[k].y); else merge_ret(n, mean, m2, cnt, a[k].x, a[k].y); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float nb, mb, qb;
        if (V == 2) {
            float* x0 = xs[0] + (threadIdx.x & ~63), *x1 = xs[1] + (threadIdx.x & ~63), *x2 = xs[2] + (threadIdx.x & ~63);
            x0[lane] = n; x1[lane] = mean; x2[lane] = m2;
            __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
            nb = x0[lane ^ o]; mb = x1[lane ^ o]; qb = x2[lane ^ o];
            __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
        } else { nb = __shfl_xor(n, o, 64); mb = __shfl_xor(mean, o, 64); qb = __shfl_xor(m2, o, 64); }
        const bool lo = (lane & o) == 0;
        float n0 = lo ? n : nb, me0 = lo ? mean : mb, q0 = lo ? m2 : qb;
        const float n1 = lo ? nb : n, me1 = lo ? mb : mean, q1 = lo ? qb : m2;
        if (V == 1) merge_nobr(n0, me0, q0, n1, me1, q1); else if (V == 7) merge_rcp(n0, me0, q0, n1, me1, q1); else merge_ret(n0, me0, q0, n1, me1, q1);
        n = n0; mean = me0; m2 = q0;
    }
    if (V == 5 && n == -12345.f) {     // never true: keeps the four address register pairs alive to the end of the kernel
        unsigned long acc = 0;
        for (int k = 0; k < 4; ++k) acc += (unsigned long)ptr[k];
        stat[0] = (float)acc;
    }
    if (lane == 0) { stat[((long)r * G + gi) * 2] = mean; stat[((long)r * G + gi) * 2 + 1] = rsqrtf(m2 / n + 1e-5f); }
}

// neutral noise: no LDS, no LDS-DMA, no MFMA -- VGPRS live values keep the register footprint at ~VGPRS
template <int VGPRS>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    float v[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// direct check of the lane exchange: every lane knows what its partner holds, so a wrong ds_bpermute result is counted on the GPU
__global__ __launch_bounds__(256) void bperm_check(unsigned* mismatches, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float mine = (float)(lane * 131 + it * 7 + o), theirs = (float)((lane ^ o) * 131 + it * 7 + o);
            const float got = __shfl_xor(mine, o, 64);
            bad += got != theirs;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// direct check of the predicated loads: tab[i] = i, every lane knows what it must receive
__global__ __launch_bounds__(256) void load_check(const float* tab, int n, int stride, unsigned* mismatches, int iters) {
    const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int off = (it * 13 + w * 7) & 63;
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sl = lane + k * 64;
            a[k] = sl < n ? tab[(long)sl * stride + off] : -1.f;          // the same form as V0's loads: one predicated load per basic block
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) bad += a[k] != (float)((lane + k * 64) * stride + off);
    }
    if (bad) atomicAdd(mismatches, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "instantir_amd/libinstantir_hip.so";
    const int REP = argc > 2 ? atoi(argv[2]) : 40;
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    auto conv = (int (*)(const iir_conv_desc*, void*))dlsym(h, "iir_conv2d_nhwc_f16");
    auto gemm = (int (*)(const iir_gemm_desc*, void*))dlsym(h, "iir_gemm_f16");
    const int R = 2, G = 32, nslab = 256, RG = R * G;
    std::vector<float> hp((size_t)R * nslab * G * 2);
    srand(1);
    for (size_t i = 0; i < hp.size(); i += 2) { hp[i] = 0.05f * ((rand() % 2001) / 1000.f - 1.f); hp[i + 1] = 300.f + (rand() % 1000) * 0.1f; }
    float *part, *stat; CK(hipMalloc(&part, hp.size() * 4)); CK(hipMalloc(&stat, RG * 2 * 4));
    CK(hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    // noise operands: conv 2x64x64x640 -> 640 (3x3) and a 4096x2560x1280 GEMM, contents irrelevant (zeros)
    void *cx, *cw, *cy, *zp, *ga, *gw, *gc;
    CK(hipMalloc(&cx, 2L * 64 * 64 * 640 * 2)); CK(hipMalloc(&cw, 640L * 9 * 640 * 2)); CK(hipMalloc(&cy, 2L * 64 * 64 * 640 * 2)); CK(hipMalloc(&zp, 4096));
    CK(hipMalloc(&ga, 4096L * 1280 * 2)); CK(hipMalloc(&gw, 2560L * 1280 * 2)); CK(hipMalloc(&gc, 4096L * 2560 * 2));
    CK(hipMemset(cx, 0, 2L * 64 * 64 * 640 * 2)); CK(hipMemset(cw, 0, 640L * 9 * 640 * 2)); CK(hipMemset(zp, 0, 4096)); CK(hipMemset(ga, 0, 4096L * 1280 * 2)); CK(hipMemset(gw, 0, 2560L * 1280 * 2));
    hipStream_t sa, sb;
    const bool prio = getenv("PROBE_PRIO") != nullptr;       // reduction stream at the highest priority, noise at the lowest
    int plo = 0, phi = 0; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    if (prio) { CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, phi)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, plo)); }
    else { CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    float* spin_out; CK(hipMalloc(&spin_out, 1024 * 256 * 4));
    iir_conv_desc cd; memset(&cd, 0, sizeof cd);
    cd.X = cx; cd.ldx = 640; cd.R = 2; cd.H = 64; cd.Wd = 64; cd.Cin = 640; cd.Wt = cw; cd.Y = cy; cd.ldy = 640; cd.Cout = 640; cd.ksize = 3; cd.stride = 1; cd.zero_page = zp;
    iir_gemm_desc gd; memset(&gd, 0, sizeof gd);
    gd.A = ga; gd.lda = 1280; gd.W = gw; gd.C = gc; gd.ldc = 2560; gd.M = 4096; gd.N = 2560; gd.K = 1280;
    std::vector<float> first(RG * 2), cur(RG * 2);
    if (getenv("PROBE_CUMASK")) {
        // ONE discriminating experiment.  The failing form (V0) beside its trigger (the library's 64x160 GEMM),
        // with the two streams (a) unrestricted, (b) on DISJOINT halves of the chip's CUs, (c) on the SAME half.  Clean when disjoint
        // => the trigger disturbs CU-local state of a co-resident wave; failing when disjoint => the memory side.
        // The reference result is taken with NO neighbour; every launch beside the neighbour is compared with it.
        uint32_t lo[8], hi[8], all[8];
        for (int i = 0; i < 8; ++i) { lo[i] = i < 4 ? 0xFFFFFFFFu : 0u; hi[i] = i < 4 ? 0u : 0xFFFFFFFFu; all[i] = 0xFFFFFFFFu; }
        struct Pl { const char* name; const uint32_t* victim; const uint32_t* trigger; } pls[] = {
            {"both streams on all CUs", all, all}, {"victim CUs 0-127, trigger CUs 128-255 (disjoint)", lo, hi},
            {"victim and trigger both on CUs 0-127", lo, lo}, {"victim CUs 128-255, trigger CUs 0-127 (disjoint)", hi, lo}};
        hipLaunchKernelGGL(fin<0>, dim3((RG + 3) / 4), dim3(256), 0, 0, part, nslab, G, 320.f, stat, RG);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(first.data(), stat, RG * 2 * 4, hipMemcpyDeviceToHost));
        for (auto& pl : pls) {
            hipStream_t sv, st;
            CK(hipExtStreamCreateWithCUMask(&sv, 8, pl.victim));
            CK(hipExtStreamCreateWithCUMask(&st, 8, pl.trigger));
            for (int with = 1; with >= 0; --with) {
                int nd = 0; float maxd = 0.f;
                for (int it = 0; it < REP; ++it) {
                    for (int k = 0; k < 3 && with; ++k) { gd.tile = 25; if (gemm(&gd, st)) { printf("gemm launch failed\n"); return 1; } }
                    CK(hipMemsetAsync(stat, 0, RG * 2 * 4, sv));
                    hipLaunchKernelGGL(fin<0>, dim3((RG + 3) / 4), dim3(256), 0, sv, part, nslab, G, 320.f, stat, RG);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(cur.data(), stat, RG * 2 * 4, hipMemcpyDeviceToHost));
                    bool d = false;
                    for (int i = 0; i < RG * 2; ++i) if (cur[i] != first[i]) { d = true; const float e = fabsf(cur[i] - first[i]); if (e > maxd) maxd = e; }
                    nd += d;
                }
                printf("%-52s %s: %2d of %d launches differ from the quiet-chip result (max |diff| %.3g)\n", pl.name,
                       with ? "beside the 64x160 GEMM" : "alone                 ", nd, REP, maxd);
            }
            CK(hipStreamDestroy(sv)); CK(hipStreamDestroy(st));
        }
        return 0;
    }
    // noise = (kind, tile): which kernel of the library keeps the other stream busy
    struct Nz { const char* name; int kind, tile; } nzs[] = {{"none", 0, 0}, {"gemm 128x160", 1, 24}, {"gemm 64x160", 1, 25},
                                                             {"spin 32 vgpr", 4, 32}, {"spin 96 vgpr", 4, 96}, {"spin 160 vgpr", 4, 160}};
    unsigned* mm; CK(hipMalloc(&mm, 4));
    float* tab; { std::vector<float> ht(256 * 64 + 64); for (size_t i = 0; i < ht.size(); ++i) ht[i] = (float)i; CK(hipMalloc(&tab, ht.size() * 4)); CK(hipMemcpy(tab, ht.data(), ht.size() * 4, hipMemcpyHostToDevice)); }
    for (auto& z : nzs) {            // the exchange on its own, 16 workgroups x 2000 iterations x 6 exchanges per launch, 20 launches
        CK(hipMemset(mm, 0, 4));
        for (int it = 0; it < 20; ++it) {
            for (int k = 0; k < 3; ++k) {
                gd.tile = z.tile;
                if (z.kind == 1 && gemm(&gd, sb)) { printf("gemm launch failed\n"); return 1; }
            }
            hipLaunchKernelGGL(bperm_check, dim3(16), dim3(256), 0, sa, mm, 2000);
            CK(hipDeviceSynchronize());
        }
        unsigned hm = 0; CK(hipMemcpy(&hm, mm, 4, hipMemcpyDeviceToHost));
        if (z.kind <= 1) printf("noise %-13s ds_bpermute self-check: %u wrong exchanges of %ld\n", z.name, hm, 20L * 16 * 256 * 2000 * 6);
        CK(hipMemset(mm, 0, 4));
        for (int it = 0; it < 20; ++it) {
            for (int k = 0; k < 3; ++k) {
                gd.tile = z.tile;
                if (z.kind == 1 && gemm(&gd, sb)) { printf("gemm launch failed\n"); return 1; }
            }
            hipLaunchKernelGGL(load_check, dim3(16), dim3(256), 0, sa, tab, 256, 64, mm, 2000);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(&hm, mm, 4, hipMemcpyDeviceToHost));
        if (z.kind <= 1) printf("noise %-13s predicated-load self-check: %u wrong values of %ld\n", z.name, hm, 20L * 16 * 256 * 2000 * 4);
    }
    for (auto& z : nzs)
        for (int v = 0; v < 8; ++v) {
            int nd = 0; float maxd = 0.f;
            for (int it = 0; it < REP; ++it) {
                for (int k = 0; k < 3; ++k) {
                    gd.tile = z.tile; cd.tile = z.tile; cd.ksize = z.kind == 3 ? 1 : 3;
                    if (z.kind == 1 && gemm(&gd, sb)) { printf("gemm launch failed\n"); return 1; }
                    if ((z.kind == 2 || z.kind == 3) && conv(&cd, sb)) { printf("conv launch failed\n"); return 1; }
                    if (z.kind == 4 && z.tile == 32) hipLaunchKernelGGL(spin<32>, dim3(1024), dim3(256), 0, sb, spin_out, 400);
                    if (z.kind == 4 && z.tile == 96) hipLaunchKernelGGL(spin<96>, dim3(1024), dim3(256), 0, sb, spin_out, 140);
                    if (z.kind == 4 && z.tile == 160) hipLaunchKernelGGL(spin<160>, dim3(1024), dim3(256), 0, sb, spin_out, 80);
                }
                CK(hipMemsetAsync(stat, 0, RG * 2 * 4, sa));
                if (v == 0) hipLaunchKernelGGL(fin<0>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 1) hipLaunchKernelGGL(fin<1>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 2) hipLaunchKernelGGL(fin<2>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 3) hipLaunchKernelGGL(fin<3>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 4) hipLaunchKernelGGL(fin<4>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 5) hipLaunchKernelGGL(fin<5>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 6) hipLaunchKernelGGL(fin<6>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                if (v == 7) hipLaunchKernelGGL(fin<7>, dim3((RG + 3) / 4), dim3(256), 0, sa, part, nslab, G, 320.f, stat, RG);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), stat, RG * 2 * 4, hipMemcpyDeviceToHost));
                if (it == 0) first = cur;
                else { bool d = false; for (int i = 0; i < RG * 2; ++i) if (cur[i] != first[i]) { d = true; float e = fabsf(cur[i] - first[i]); if (e > maxd) maxd = e; } nd += d; }
            }
            printf("noise %-13s variant V%d: %2d of %d launches differ from the first (max |diff| %.3g)\n", z.name, v, nd, REP - 1, maxd);
        }
    return 0;
}
