// L2 -> LDS fill-rate probe: how many bytes per second can ONE CU pull from L2-resident operand panels
// into LDS, by staging form and wave count?  Emulates the operand traffic of the 2048 x 1280 x K projection GEMM (64 x 160
// tile, one workgroup per CU, 256 workgroups): per K tile 64 rows of A and 160 rows of W, 128 bytes each = 28 KiB, no MFMA,
// no fragment reads.  A 5 MB + W 3.3 MB are L2 / Infinity-Cache resident after the first pass.
//   mode 0: LDS-DMA (global_load_lds_dwordx4), ring of ST stages, counted vmcnt + one barrier per K tile (the shipped protocol)
//   mode 1: global_load_dwordx4 -> registers -> ds_write_b128, one tile of registers in flight, barrier per tile
//   mode 2: LDS-DMA, no barrier: every wave streams its own pieces with a counted vmcnt throttle (raw issue / return rate)
//   mode 3: global_load_dwordx4 -> registers, results discarded (no LDS write): the L2 -> register rate
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/fill_probe tools/probes/fill_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
typedef _Float16 f16;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 160, ROWS = BM + BN;   // 224 rows of 128 B per K tile = 28 pieces of 1 KiB

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NW, int ST, int TPB = 1, int LPWO = 0>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const f16* __restrict__ A, const f16* __restrict__ W, int K, int reps, float* sink, int xcd_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile of this workgroup: XCD x = blockIdx & 7.  xcd_rows = 1: XCD x owns row tiles [4x, 4x+4) and all 8 column tiles;
    // xcd_rows = 0: XCD x owns column tile x and all 32 row tiles
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tm = xcd_rows ? x * 4 + (l & 3) : l;
    const int tn = xcd_rows ? (l >> 2) : x;
    constexpr int PIECES = ROWS / 8;                       // 28 pieces of 8 rows
    constexpr int PPW = (PIECES + NW - 1) / NW;            // pieces per wave per tile
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const f16* src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int p = i * NW + wave;
        if (p >= PIECES) p = PIECES - 1;                   // surplus waves re-fetch the last piece (keeps the counts uniform)
        const int r = p * 8 + srow;
        src[i] = r < BM ? A + (long)(tm * BM + r) * K + schunk * 8 : W + (long)(tn * BN + (r - BM)) * K + schunk * 8;
    }
    const int nk = K / 64;
    const int total = nk * reps;
    float acc = 0.f;
    if constexpr (MODE == 0) {
        auto stage = [&](int t, int buf) {
            const int k0 = (t % nk) * 64;
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                int p = i * NW + wave;
                if (p >= PIECES) p = PIECES - 1;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src[i] + k0), (LDS_AS void*)(smem + buf * ROWS * 128 + p * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int s = 0; s < ST - 1; ++s) stage(s, s);
        int cur = 0;
        for (int t = 0; t < total; ++t) {
            wait_vm<(ST - 2) * PPW>();
            __builtin_amdgcn_s_barrier();
            if (t + ST - 1 < total) stage(t + ST - 1, (cur + ST - 1) % ST);
            else stage(t, (cur + ST - 1) % ST);            // keep the in-flight count constant to the end
            cur = cur + 1 == ST ? 0 : cur + 1;
        }
        wait_vm<0>();
    } else if constexpr (MODE == 2) {
        for (int t = 0; t < total; ++t) {
            const int k0 = (t % nk) * 64, buf = t % ST;
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                int p = i * NW + wave;
                if (p >= PIECES) p = PIECES - 1;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src[i] + k0), (LDS_AS void*)(smem + buf * ROWS * 128 + p * 1024), 16, 0, 0);
            }
            wait_vm<(ST - 1) * PPW>();
        }
        wait_vm<0>();
    } else if constexpr (MODE == 1) {
        f32x4 r0[PPW], r1[PPW];
        auto ld = [&](int t, f32x4 (&r)[PPW]) {
            const int k0 = (t % nk) * 64;
#pragma unroll
            for (int i = 0; i < PPW; ++i) r[i] = *(const f32x4*)(src[i] + k0);
        };
        auto st = [&](int buf, const f32x4 (&r)[PPW]) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                int p = i * NW + wave;
                if (p >= PIECES) p = PIECES - 1;
                *(f32x4*)(smem + buf * ROWS * 128 + p * 1024 + lane * 16) = r[i];
            }
        };
        ld(0, r0);
        for (int t = 0; t < total; t += 2) {
            ld(t + 1, r1);
            st(0, r0);
            __syncthreads();
            ld(t + 2, r0);
            st(1, r1);
            __syncthreads();
        }
        acc += r0[0][0];
    } else if constexpr (MODE == 3) {
        f32x4 r0[PPW], r1[PPW];
        auto ld = [&](int t, f32x4 (&r)[PPW]) {
            const int k0 = (t % nk) * 64;
#pragma unroll
            for (int i = 0; i < PPW; ++i) r[i] = *(const f32x4*)(src[i] + k0);
        };
        ld(0, r0);
        for (int t = 0; t < total; t += 2) {
            ld(t + 1, r1);
#pragma unroll
            for (int i = 0; i < PPW; ++i) acc += r0[i][0] + r0[i][3];
            ld(t + 2, r0);
#pragma unroll
            for (int i = 0; i < PPW; ++i) acc += r1[i][0] + r1[i][3];
        }
    }
    if constexpr (MODE >= 4) {
        // waves [0, 4): loaders (ring of ST stages, counted vmcnt, one barrier per tile); waves [4, 8): compute stand-ins
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f16v __attribute__((ext_vector_type(16)));
        constexpr int LPW = LPWO ? LPWO : PIECES / 4;     // 7 pieces per loader wave (LPWO: fewer -- the conv-halo question: what is a smaller fill worth?)
        constexpr int GB = TPB * ROWS * 128; // bytes of one ring slot (TPB tiles)
        const int groups = total / TPB;
        if (wave < 4) {
            __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 2048 * K * 2, 0x00020000);
            __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 1280 * K * 2, 0x00020000);
            const int voff = srow * K * 2 + schunk * 16;          // per lane, the same for every piece and tile
            if constexpr (MODE == 10) __builtin_amdgcn_s_setprio(3);
            auto stage = [&](int gidx, int slot) {
#pragma unroll
                for (int u = 0; u < TPB; ++u) {
                    const int k0 = ((gidx * TPB + u) % nk) * 64;
#pragma unroll
                    for (int i = 0; i < LPW; ++i) {
                        const int p = i * 4 + wave, r = p * 8 + srow;
                        if constexpr (MODE == 9 || MODE == 10) {
                            // scalar piece offset: the loader's loop has no vector ALU instruction at all
                            const int r0 = p * 8;
                            if (r0 < BM) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)(smem + slot * GB + u * ROWS * 128 + p * 1024), 16, voff, ((tm * BM + r0) * K + k0) * 2, 0, 0);
                            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (LDS_AS void*)(smem + slot * GB + u * ROWS * 128 + p * 1024), 16, voff, ((tn * BN + r0 - BM) * K + k0) * 2, 0, 0);
                        } else {
                            const f16* sp = r < BM ? A + (long)(tm * BM + r) * K + schunk * 8 : W + (long)(tn * BN + (r - BM)) * K + schunk * 8;
                            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sp + k0), (LDS_AS void*)(smem + slot * GB + u * ROWS * 128 + p * 1024), 16, 0, 0);
                        }
                    }
                }
            };
#pragma unroll
            for (int s2 = 0; s2 < ST - 1; ++s2) stage(s2, s2);
            int cur = 0;
            for (int t = 0; t < groups; ++t) {
                wait_vm<(ST - 2) * LPW * TPB>();
                __builtin_amdgcn_s_barrier();
                stage(t + ST - 1, (cur + ST - 1) % ST);
                cur = cur + 1 == ST ? 0 : cur + 1;
            }
            wait_vm<0>();
        } else {
            h8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
            f4 c4[10];
            f16v c16[5];
            for (int i = 0; i < 10; ++i) c4[i] = (f4){0, 0, 0, 0};
            for (int i = 0; i < 5; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
            int cur = 0;
            for (int t = 0; t < groups; ++t) {
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int u = 0; u < TPB; ++u) {
                    if constexpr (MODE == 6 || MODE == 7 || MODE == 9 || MODE == 10) {
                        // the GEMM's conflict-free fragment reads: row frow of a 16-row group, chunk (4 s + fq) ^ (frow & 7); 7 groups x 2 k-steps
                        const int frow = lane & 15, fq = lane >> 4;
                        const char* base = smem + cur * GB + u * ROWS * 128 + frow * 128;
#pragma unroll
                        for (int i = 0; i < 14; ++i) { h8 v = *(const h8*)(base + (i >> 1) * 2048 + ((((i & 1) * 4 + fq) ^ (frow & 7)) * 16)); a[i & 7] += v[i & 7]; }
                    }
                    if constexpr (MODE == 4 || MODE == 7 || MODE == 9 || MODE == 10) {
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int i = 0; i < 10; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[i], 0, 0, 0);
                    }
                    if constexpr (MODE == 5) {
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int i = 0; i < 5; ++i) c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[i], 0, 0, 0);
                    }
                }
                cur = cur + 1 == ST ? 0 : cur + 1;
            }
            for (int i = 0; i < 10; ++i) acc += c4[i][0];
            for (int i = 0; i < 5; ++i) acc += c16[i][0];
            acc += (float)a[0];
        }
    }
    __syncthreads();
    acc += ((const float*)smem)[tid];
    if (acc == 123.456f) sink[0] = acc;
}


// ---- "B direct" probe: the weight operand never touches LDS.  4 waves laid out 1 (M) x 4 (N) over the 64 x 160 tile: wave w owns
// the 16-column blocks {w, w+4, w+8 (waves 0,1 only)}; its weight fragments come straight from memory into registers (16 rows x
// 64 B per instruction, the MFMA B-operand layout), three K tiles deep; the activation tile (64 rows x 128 B) goes through the
// LDS-DMA ring as before and every wave reads all of it.  Per K tile and CU: 8 KiB DMA + 20 KiB register loads through the
// vector memory path, 32 KiB of fragment reads, 24 / 16 MFMA 16x16x32 per wave.  FULL = 0: no MFMA (loads + reads only).
template <int FULL>
__global__ __launch_bounds__(256, 2) void bd_kernel(const f16* __restrict__ A, const f16* __restrict__ W, int K, int reps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int tm = x * 4 + (l & 3), tn = l >> 2;
    const int nk = K / 64, total = nk * reps;
    constexpr int ST = 3;
    const int nblk = wave < 2 ? 3 : 2;
    // A staging: 8 pieces of 8 rows, 2 per wave
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const f16* asrc[2];
    for (int i = 0; i < 2; ++i) asrc[i] = A + (long)(tm * 64 + (i * 4 + wave) * 8 + srow) * K + schunk * 8;
    auto stage_a = [&](int t, int buf) {
        const int k0 = (t % nk) * 64;
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(asrc[i] + k0), (LDS_AS void*)(smem + buf * 8192 + (i * 4 + wave) * 1024), 16, 0, 0);
    };
    const int frow = lane & 15, fq = lane >> 4;
    const f16* bsrc[3];
    // FULL & 2: fragment-major weights (packed once at load time): [N / 16][K / 32][64 lanes][8 halves], a whole B fragment is one
    // contiguous KiB in lane order (fully coalesced); otherwise row-major [N][K]: 16 rows x 64 B per instruction
    for (int j = 0; j < 3; ++j)
        bsrc[j] = (FULL & 2) ? W + (long)(tn * 10 + min(wave + 4 * j, 9)) * (K / 32) * 512 + lane * 8
                             : W + (long)(tn * 160 + min(wave + 4 * j, 9) * 16 + frow) * K + fq * 8;
    h8 b[3][3][2];
    // register loads in asm (the compiler's own bookkeeping turns every wait in this loop into vmcnt(0)); all waves issue three
    // blocks (waves 2, 3: the third is a clamped duplicate) so the counts are uniform
    auto load_b = [&](int t, h8 (&r)[3][2]) {
        const int k0 = (FULL & 2) ? (t % nk) * 1024 : (t % nk) * 64, k1 = (FULL & 2) ? k0 + 512 : k0 + 32;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j][0]) : "v"(bsrc[j] + k0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j][1]) : "v"(bsrc[j] + k1) : "memory");
        }
    };
    f4 acc[4][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) acc[i][j] = (f4){0, 0, 0, 0};
    h8 asum = {0, 0, 0, 0, 0, 0, 0, 0};
    stage_a(0, 0); load_b(0, b[0]); stage_a(1, 1); load_b(1, b[1]); load_b(2, b[2]);
    auto body = [&](int t, int buf, h8 (&r)[3][2]) {
        // issue order per iteration: A(i+2) [2 DMA], B(i+3) [6 loads]; before iteration t the queue holds, oldest first,
        // B(t), A(t), B(t+1), A(t+1), B(t+2): 14 younger than what this tile needs
        asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)\n\ts_barrier" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1]) :: "memory");
        stage_a(t + 2, (buf + 2) % ST);
        h8 af[4][2];
        const char* base = smem + buf * 8192 + frow * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) af[i][s2] = *(const h8*)(base + i * 2048 + (((s2 * 4 + fq) ^ (frow & 7)) * 16));
        if constexpr (FULL & 1) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < nblk)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r[j][s2], af[i][s2], acc[i][j], 0, 0, 0);
        } else {
            for (int i = 0; i < 4; ++i) for (int s2 = 0; s2 < 2; ++s2) asum += af[i][s2];
            for (int j = 0; j < 3; ++j) { asum += r[j][0]; asum += r[j][1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_b(t + 3, r);
    };
    for (int t = 0; t < total; t += 3) {
        body(t, 0, b[0]);
        body(t + 1, 1, b[1]);
        body(t + 2, 2, b[2]);
    }
    float a = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) a += acc[i][j][0];
    a += (float)asum[0] + (float)b[0][0][0][0] + (float)b[1][0][0][0] + (float)b[2][0][0][0];
    __syncthreads();
    a += ((const float*)smem)[tid];
    if (a == 123.456f) sink[0] = a;
}

template <int FULL>
void run_bd(const char* name, const f16* A, const f16* W, int K, float* sink) {
    const int reps = 42, grid = 256;
    const size_t lds = 90 * 1024;
    auto k = bd_kernel<FULL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, K, reps, sink);
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, A, W, K, reps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_tile = ms * 1e3 / iters / ((double)(K / 64) * reps);
    printf("%-60s %6.3f us per K tile\n", name, us_tile);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
}

template <int MODE, int NW, int ST, int TPB = 1, int LPWO = 0>
void run(const char* name, const f16* A, const f16* W, int K, float* sink, int xcd_rows, int grid) {
    const int reps = 40;
    const size_t lds = (size_t)(MODE == 1 ? 2 : ST) * TPB * ROWS * 128 > 90 * 1024 ? (size_t)ST * TPB * ROWS * 128 : 90 * 1024;   // >= 90 KiB: one workgroup per CU
    auto k = fill_kernel<MODE, NW, ST, TPB, LPWO>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, A, W, K, reps, sink, xcd_rows);
    hipEventRecord(e0);
    const int iters = 5;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, A, W, K, reps, sink, xcd_rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tiles = (double)(K / 64) * reps;
    const double us_tile = ms * 1e3 / iters / tiles;
    printf("%-34s grid %3d xcd_%s  %6.3f us per 28-KiB K tile  = %6.1f GB/s per CU, %5.2f TB/s chip\n", name, grid, xcd_rows ? "rows" : "cols", us_tile,
           ROWS * 128 / us_tile * 1e-3, ROWS * 128 / us_tile * 1e-3 * grid * 1e-3);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
}

int main(int argc, char** argv) {
    const int M = 2048, N = 1280, K = 1280;
    f16 *A, *W;
    float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&sink, 64);
    std::vector<f16> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (f16)((float)(rand() % 2001 - 1000) * 1e-3f);
    hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    if (argc > 1 && argv[1][0] == 'h') {   // same compute (14 reads + 20 MFMA per wave and K tile), fewer bytes filled per K tile
        run<7, 8, 3, 1, 7>("reads + mfma16, 28 KiB filled per K tile (shipped)", A, W, K, sink, 1, 256);
        run<7, 8, 3, 1, 6>("reads + mfma16, 24 KiB filled per K tile", A, W, K, sink, 1, 256);
        run<7, 8, 3, 1, 5>("reads + mfma16, 20 KiB filled per K tile", A, W, K, sink, 1, 256);
        run<7, 8, 3, 1, 4>("reads + mfma16, 16 KiB filled per K tile", A, W, K, sink, 1, 256);
        run<7, 8, 3, 1, 2>("reads + mfma16,  8 KiB filled per K tile", A, W, K, sink, 1, 256);
        run<7, 8, 3, 1, 7>("reads + mfma16, 28 KiB filled per K tile (again)", A, W, K, sink, 1, 256);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'b') {   // weight operand straight into registers (no LDS): does the K tile get cheaper than the 0.43 us of the shipped loop?
        run<7, 8, 3>("shipped form: 4 loaders + 4 waves x (14 reads + 20 mfma16)", A, W, K, sink, 1, 256);
        run_bd<0>("B direct: A by LDS-DMA + reads, B to registers, no MFMA", A, W, K, sink);
        run_bd<1>("B direct: the same with the MFMAs (24 / 16 per wave)", A, W, K, sink);
        run_bd<2>("B direct, fragment-major weights: loads + reads, no MFMA", A, W, K, sink);
        run_bd<3>("B direct, fragment-major weights: with the MFMAs", A, W, K, sink);
        run_bd<3>("B direct, fragment-major weights: again", A, W, K, sink);
        return 0;
    }
    if (argc > 1) {   // loader / compute co-issue: what slows the LDS-DMA issue of the loader-wave GEMM (880 cycles per K tile against 480 here)?
        run<0, 4, 3>("LDS-DMA 4 waves ring3 barrier (loaders alone)", A, W, K, sink, 1, 256);
        run<4, 8, 3>("4 loaders + 4 waves x 20 mfma16x16x32", A, W, K, sink, 1, 256);
        run<5, 8, 3>("4 loaders + 4 waves x 10 mfma32x32x16", A, W, K, sink, 1, 256);
        run<6, 8, 3>("4 loaders + 4 waves x 14 ds_read_b128", A, W, K, sink, 1, 256);
        run<7, 8, 3>("4 loaders + 4 waves x (14 reads + 20 mfma16)", A, W, K, sink, 1, 256);
        run<7, 8, 4>("same, ring of 4 tiles", A, W, K, sink, 1, 256);
        run<7, 8, 2, 2>("same, 2 tiles per barrier, ring 2x2", A, W, K, sink, 1, 256);
        run<9, 8, 3>("reads + mfma16, loaders by buffer_load..lds with scalar offsets", A, W, K, sink, 1, 256);
        run<10, 8, 3>("same, loaders at s_setprio 3", A, W, K, sink, 1, 256);
        run<9, 8, 2, 2>("buffer loaders, 2 tiles per barrier, ring 2x2", A, W, K, sink, 1, 256);
        run<0, 4, 3>("LDS-DMA 4 waves ring3 barrier (loaders alone)", A, W, K, sink, 1, 256);
        return 0;
    }
    for (int xr = 0; xr < 2; ++xr) {
        run<0, 4, 3>("LDS-DMA 4 waves ring3 barrier", A, W, K, sink, xr, 256);
        run<0, 8, 3>("LDS-DMA 8 waves ring3 barrier", A, W, K, sink, xr, 256);
        run<0, 14, 3>("LDS-DMA 14 waves ring3 barrier", A, W, K, sink, xr, 256);
        run<2, 4, 5>("LDS-DMA 4 waves free-run d5", A, W, K, sink, xr, 256);
        run<2, 14, 5>("LDS-DMA 14 waves free-run d5", A, W, K, sink, xr, 256);
        run<1, 4, 2>("reg + ds_write 4 waves", A, W, K, sink, xr, 256);
        run<1, 14, 2>("reg + ds_write 14 waves", A, W, K, sink, xr, 256);
        run<3, 4, 2>("reg only 4 waves", A, W, K, sink, xr, 256);
        run<3, 14, 2>("reg only 14 waves", A, W, K, sink, xr, 256);
    }
    run<2, 8, 5>("LDS-DMA 8 waves free-run d5", A, W, K, sink, 1, 64);
    run<3, 8, 2>("reg only 8 waves", A, W, K, sink, 1, 64);
    return 0;
}
