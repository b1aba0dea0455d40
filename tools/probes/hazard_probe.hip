// Hazard hunt for DESIGN.md section 5.8: loads hand-edited code objects of the failing reduction (variant V0 of shfl_probe.hip,
// `hipcc -S`, s_nop pads inserted at one class of places per object, assembled with clang + ld.lld) and runs each beside the
// library's 64x160 GEMM on a second stream, comparing the output of every launch with the first.
//   hipcc --offload-arch=gfx950 -O3 -I include tools/probes/hazard_probe.hip -o tools/probes/bin/hazard_probe -ldl
//   tools/probes/bin/hazard_probe instantir_amd/libinstantir_hip.so 30 tools/probes/bin/hz_*.co
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "instantir_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    auto gemm = (int (*)(const iir_gemm_desc*, void*))dlsym(h, "iir_gemm_f16");
    const int REP = atoi(argv[2]);
    const int R = 2, G = 32, nslab = 256, RG = R * G;
    std::vector<float> hp((size_t)R * nslab * G * 2);
    srand(1);
    for (size_t i = 0; i < hp.size(); i += 2) { hp[i] = 0.05f * ((rand() % 2001) / 1000.f - 1.f); hp[i + 1] = 300.f + (rand() % 1000) * 0.1f; }
    float *part, *stat; CK(hipMalloc(&part, hp.size() * 4)); CK(hipMalloc(&stat, RG * 2 * 4));
    CK(hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    void *ga, *gw, *gc;
    CK(hipMalloc(&ga, 4096L * 1280 * 2)); CK(hipMalloc(&gw, 2560L * 1280 * 2)); CK(hipMalloc(&gc, 4096L * 2560 * 2));
    CK(hipMemset(ga, 0, 4096L * 1280 * 2)); CK(hipMemset(gw, 0, 2560L * 1280 * 2));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    iir_gemm_desc gd; memset(&gd, 0, sizeof gd);
    gd.A = ga; gd.lda = 1280; gd.W = gw; gd.C = gc; gd.ldc = 2560; gd.M = 4096; gd.N = 2560; gd.K = 1280; gd.tile = 25;     // 64x160, two stages
    std::vector<float> first(RG * 2), cur(RG * 2);
    for (int a = 3; a < argc; ++a) {
        hipModule_t mod; hipFunction_t fn;
        CK(hipModuleLoad(&mod, argv[a]));
        CK(hipModuleGetFunction(&fn, mod, "_Z3finILi0EEvPKfiifPfi"));
        for (int nz = 0; nz < 2; ++nz) {
            int nd = 0; float maxd = 0.f;
            for (int it = 0; it < REP; ++it) {
                if (nz) for (int k = 0; k < 3; ++k) if (gemm(&gd, sb)) { printf("gemm launch failed\n"); return 1; }
                CK(hipMemsetAsync(stat, 0, RG * 2 * 4, sa));
                int ns = nslab, g_ = G, rg = RG; float cnt = 320.f;
                void* params[] = {&part, &ns, &g_, &cnt, &stat, &rg};
                CK(hipModuleLaunchKernel(fn, (RG + 3) / 4, 1, 1, 256, 1, 1, 0, sa, params, nullptr));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(cur.data(), stat, RG * 2 * 4, hipMemcpyDeviceToHost));
                if (it == 0) first = cur;
                else { bool d = false; for (int i = 0; i < RG * 2; ++i) if (cur[i] != first[i]) { d = true; float e = fabsf(cur[i] - first[i]); if (e > maxd) maxd = e; } nd += d; }
            }
            printf("%-40s noise %-12s: %2d of %d launches differ from the first (max |diff| %.3g)\n", argv[a], nz ? "gemm 64x160" : "none", nd, REP - 1, maxd);
        }
        CK(hipModuleUnload(mod));
    }
    return 0;
}
