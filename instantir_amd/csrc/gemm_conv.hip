// K1/K2/K7/K8: fp16 MFMA GEMM and implicit-GEMM convolution (NHWC) with fused epilogues, gfx950.
//
// Replaces on the reference's hot path: every nn.Linear call site of the UNet /
// Aggregator / adapters (module/ip_adapter/attention_processor.py:370,377-378,402,1140,1148-1149,
// 1173-1174,1195; GEGLU FF per module/min_sdxl.py:502-528), every 3x3 / 1x1 nn.Conv2d
// (ResnetBlock2D module/min_sdxl.py:242-283, Down/Upsample2D :598-618, SFT module/aggregator.py:70-90).
//
// Math: C[m][n] = epi( sum_k A[m][k] * W[n][k] ), fp16 operands, fp32 accumulate (MFMA 16x16x32 f16).
//   GEMM : A is [M][K] row-major (lda), W is [N][K] (torch Linear layout).
//   CONV : A row m = output pixel (img, oy, ox) of an NHWC tensor, k = (ky*ks+kx)*Cin + c; the A tile
//          is gathered straight from the input image (zero page for padding) -- no im2col buffer.
//          W is [Cout][ks][ks][Cin].  stride 1/2; `ups` folds a nearest-2x upsample into the gather.
//
// Structure: BMxBNx64 block tile, 4 waves (2x2; 8 for the 256x128 tile), operands staged by global_load_lds_dwordx4
// into an ST-deep ring of XOR-swizzled LDS images (swizzle applied on the per-lane SOURCE address so the LDS write
// stays lane-linear), fragments by ds_read_b128 double-buffered across the barrier, XCD-aware tile order.
// The weight fragment is the MFMA "A" operand so each lane ends up with 4 consecutive output columns of one row;
// the finished tile is staged through the (then idle) ring and written out in whole rows (or, for a column range,
// transposed).  Optional: two-slice split-K with an in-launch reduction, LayerNorm of the A rows folded in.
#include "common.h"
#include "gemm_geo.h"
#include "../../include/instantir_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using iir::Geo;

constexpr int BK = 64;   // halfs per K tile = one 128-byte LDS row
// Issue-order pin (round 3).  The K loop is written as  [fragment reads of the NEXT half step] [MFMAs of the CURRENT half step],
// so that the LDS latency runs under the matrix instructions.  hipcc's scheduler sank the reads below the MFMAs and then waited
// `lgkmcnt(0)` right behind them -- one exposed LDS round trip (~200 cycles with four waves reading and the LDS-DMA writing) per
// half step, with one wave per SIMD nothing else to cover it.  A scheduling barrier after each read group keeps the source order.
#ifndef IIR_NO_PIN
#define IIR_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define IIR_PIN() ((void)0)
#endif
constexpr int PF_TOUCHES = 4;   // prefetch touches per lane per launch (x 128 B x threads = up to 128-256 KiB per workgroup)
#define IIR_DEFAULT_STAGES 2
#ifndef IIR_T1_MIN
#define IIR_T1_MIN 384
#endif
#ifndef IIR_T2_MIN
#define IIR_T2_MIN 256
#endif


// Chan's pairwise update of (count, mean, M2) -- the same form norm.hip uses for GroupNorm
__device__ __forceinline__ void ln_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    const float tot = n + nb, d = mb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}

constexpr int vmcnt_imm(int n) { return (n & 15) | 0x0F70 | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only

// wait until at most N of this wave's vector-memory ops (the LDS-DMA loads) are outstanding, then barrier.
// One asm statement with a memory clobber: no LDS access may be scheduled across it.
template <int N>
__device__ __forceinline__ void wait_vm_lgkm_and_barrier() {
#ifdef IIR_DBG_NO_BARRIER
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
#ifdef IIR_DBG_NO_BARRIER      // (timing experiment of DESIGN.md 5.10, results wrong: the K loop's barrier removed)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// W8 (BASELINE configs[4]: "fp8 MFMA weights"): the weight operand is fp8-E4M3 with one fp32 scale per output channel.  Its
// tile image has 64-byte rows (64 K values), 16 rows per LDS-DMA instruction, 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 2) & 3) (conflict-free 8-byte fragment reads); the activation tile stays fp16 in LDS and each A fragment is
// converted to fp8 in registers (v_cvt_scalef32_pk_fp8_f16, scale 1) -- 4 conversions per fragment, reused by all NI column
// tiles -- so the MFMA is v_mfma_f32_16x16x32_fp8_fp8.  Staged bytes per K tile drop from (BM + BN) * 128 to
// BM * 128 + BN * 64: the weight panel, which dominates the level-2 projections' fill, halves.
// LW (loader waves): the workgroup carries NW extra waves that do nothing but issue the LDS-DMA instructions of the K
// loop (and wait for them); the NW compute waves issue only fragment reads and MFMAs.  With one workgroup per CU (the
// level-2 projections: 256 tiles of 64x160) a compute wave otherwise sits ~60 cycles in the issue of every 1-KiB LDS-DMA piece,
// 7 pieces per K tile, with its SIMD's matrix pipe idle behind it (MI355X_MICROARCH.md, LDS-DMA issue cost) -- the "serial
// round trip" of DESIGN.md section 5.6.  Both kinds of wave meet at the one barrier per K tile; the protocol (tile kt+1 landed
// and tile kt's buffer drained at the barrier of iteration kt) is unchanged.  All 2*NW waves share the epilogue write-out.
// F8 (round 3): BOTH operands are fp8-E4M3 bytes.  A row of 128 K values is the same 128 bytes as a row of 64 halves, so the host
// hands the launch over with K and lda counted in 2-byte units and everything up to the fragment registers -- addresses, LDS-DMA
// staging, swizzle, conflict-free 16-byte fragment reads -- is the fp16 path unchanged; the 16 bytes a lane holds are 16 K values,
// fed to two `v_mfma_f32_16x16x32_fp8_fp8` (low and high 8 bytes: any split of the contraction index is valid as long as both
// operands use the same one).  Per K tile: the same MFMA count per K value as fp16, HALF the 128-byte lines per FLOP -- which is
// what the one-per-CU GEMMs are bound by.  Scales: wscale[n] per output channel, a_scale for the activations.
template <typename E, int BM, int BN, int ST, bool CONV, int WAVES_M = 2, bool W8 = false, bool LW = false, bool F8 = false>
__global__ __launch_bounds__(128 * WAVES_M * (LW ? 2 : 1), (WAVES_M == 2 ? 2 : 1)) void gemm_kernel(const Geo g) {
    static_assert(!(F8 && (W8 || CONV)), "the all-fp8 build covers the linear layers; W8 is the fp16-activation form");
    static_assert(!(W8 && CONV), "the fp8-weight build covers the linear layers only");
    static_assert(!(LW && W8), "loader waves: fp16 / bf16 operands only");
    using E4 = typename ET<E>::x4;
    using E8 = typename ET<E>::x8;     // (pointers stay f16-typed: both element types are 2 bytes; only conversions differ)
    constexpr int NW = WAVES_M * 2;                 // compute waves, laid out WAVES_M x 2 over the tile (= staging waves)
    constexpr int NT = 64 * NW * (LW ? 2 : 1);      // threads of the workgroup
    constexpr int WM = BM / WAVES_M, WN = BN / 2;   // wave tile
    constexpr int MI = WM / 16, NI = WN / 16;    // 16x16 MFMA tiles per wave
    constexpr int B_GROUPS = W8 ? BN / 16 : BN / 8;                   // LDS-DMA instructions that cover the weight tile
    constexpr int A_INST = BM / 8 / NW, B_INST = (B_GROUPS + NW - 1) / NW;   // glds instructions per wave per K tile
    constexpr int B_STAGE_BYTES = W8 ? BN * 64 : BN * 128;
    constexpr int RING_BYTES = ST * (BM * 128 + B_STAGE_BYTES);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* As = (f16*)smem;                       // [ST][BM][64]
    f16* Bs = As + ST * BM * BK;                // [ST][BN][64]  (W8: [ST][BN][64 bytes])
    constexpr int LOADS = A_INST + B_INST;      // LDS-DMA instructions per wave per stage (every wave issues exactly this many)
    constexpr int SCRATCH_BYTES = 256 * NW * (LW ? 2 : 1);        // 256 B per wave (prefetch touches, dummy loads)
    float2* rowstat = (float2*)(smem + RING_BYTES + SCRATCH_BYTES);          // [BM] (rstd, -rstd * mean) of this tile's rows (ln_in)
    constexpr int LNP_OFF = (BM * (2 * BN + 32) + 15) / 16 * 16;              // producer partials sit behind the finished output tile
    constexpr bool LN_OUT_FITS = LNP_OFF + BM * (BN / 8) * 8 <= RING_BYTES;
    // GroupNorm partials (round 3): per-thread column statistics of a 64-row slab meet in [row groups][BN] float2 behind the tile
    constexpr int GN_RG = NT / (BN / 8);                                     // row groups of the column-fixed write-out mapping
    constexpr bool GN_OUT_FITS = BM % 64 == 0 && LNP_OFF + GN_RG * BN * 8 <= RING_BYTES && BN <= NT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = LW && wave_all >= NW;                       // wave-uniform role
    const int wave = loader ? wave_all - NW : wave_all;             // index among the waves of its role
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order.  Workgroups b, b+8, b+16, ... share an XCD (= one private L2).  The tile grid is
    // cut into 8 rectangles (xm x 8/xm), one per XCD, chosen on the host to minimise the operand bytes each
    // L2 has to pull over the fabric; inside a rectangle tiles walk M fastest so co-resident workgroups
    // share a weight panel.  Placement only affects speed, never results.
    // split-K: the grid holds every tile twice; the second half of the grid (kz = 1) works on the upper half of K.
    // Both workgroups of a tile keep the same index mod 8, i.e. the same XCD / L2.
    const int kz = (g.splitk == 2 && blockIdx.x >= (gridDim.x >> 1)) ? 1 : 0;
    int tm, tn;
    {
        const int bid = kz ? (int)blockIdx.x - (int)(gridDim.x >> 1) : (int)blockIdx.x, xcd = bid & 7, local = bid >> 3;
        const int rx = xcd % g.xm, ry = xcd / g.xm;
        tm = rx * g.rm + local % g.rm;
        tn = ry * g.rn + local / g.rm;
        if (tm >= g.tiles_m || tn >= g.tiles_n) return;     // padding workgroup of a ragged rectangle
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane staging addresses -------------------------------------------------------------
    const int srow = lane >> 3;                       // row inside an 8-row glds instruction
    const int schunk = (lane & 7) ^ srow;             // swizzled 16-byte chunk fetched by this lane
    const f16* a_src[A_INST];
    int a_pix_y[A_INST], a_pix_x[A_INST];             // conv: output pixel coords (input domain origin)
    const f16* a_img[A_INST];
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
        int m = m0 + (i * NW + wave) * 8 + srow;
        if (m >= g.M) m = g.M - 1;                    // clamp: tail rows are computed and discarded
        if (!CONV) {
            a_src[i] = g.A + (long)m * g.lda + schunk * 8;
        } else {
            const int hw = g.Ho * g.Wo;
            const int img = m / hw, rem = m - img * hw;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_img[i] = g.A + (long)img * g.x_img_stride + schunk * 8;
            a_pix_y[i] = oy * g.stride - g.pad;
            a_pix_x[i] = ox * g.stride - g.pad;
        }
    }
    const f16* b_src[B_INST];
#pragma unroll
    for (int i = 0; i < B_INST; ++i) {
        if (!W8) {
            int n = n0 + (i * NW + wave) * 8 + srow;
            if (n >= g.N) n = g.N - 1;
            b_src[i] = g.W + (long)n * g.K + schunk * 8;
        } else {                                      // 16 rows of 64 bytes per instruction; lane = (row, 16-byte chunk)
            const int r = (i * NW + wave) * 16 + (lane >> 2);
            int n = n0 + r;
            if (n >= g.N) n = g.N - 1;
            b_src[i] = (const f16*)((const char*)g.W + (long)n * g.K + (((lane & 3) ^ ((r >> 2) & 3)) * 16));
        }
    }

    // conv gather state: stage() is called for consecutive K tiles, so the (tap, channel) walk is incremental -- per tile one
    // pointer bump per row group; the pixel / padding arithmetic runs only when the tap changes (every Cin/64 tiles).  It
    // used to be redone for every tile: ~40 VALU instructions beside 40 MFMAs.
    const f16* cv_ptr[CONV ? A_INST : 1];
    int cv_inc[CONV ? A_INST : 1];
    int cv_c0 = 0, cv_tap = 0;
    auto conv_tap = [&](int tap) {
        const int ky = tap / g.ks, kx = tap - ky * g.ks;
#pragma unroll
        for (int i = 0; i < (CONV ? A_INST : 0); ++i) {
            const int iy = a_pix_y[i] + ky, ix = a_pix_x[i] + kx;
            bool ok;
            long pix;
            if (g.ups) {   // coordinates are in the 2x-upsampled domain
                ok = (iy >= 0) & (iy < 2 * g.H) & (ix >= 0) & (ix < 2 * g.Wd);
                pix = (long)(iy >> 1) * g.Wd + (ix >> 1);
            } else {
                ok = (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.Wd);
                pix = (long)iy * g.Wd + ix;
            }
            cv_ptr[i] = ok ? a_img[i] + pix * g.lda : g.zero + (lane & 7) * 8;      // padding pixels read the zero page, in place
            cv_inc[i] = ok ? BK : 0;
        }
    };
    auto conv_seek = [&](int kt) {
        const int k0 = kt * BK;
        cv_tap = k0 / g.Cin;
        cv_c0 = k0 - cv_tap * g.Cin;
        conv_tap(cv_tap);
#pragma unroll
        for (int i = 0; i < (CONV ? A_INST : 0); ++i) cv_ptr[i] += cv_inc[i] ? cv_c0 : 0;
    };

    // (timing experiments of DESIGN.md 5.10, results wrong: -DIIR_DBG_THIN_A / _B stage 4 bytes per lane instead of 16 for the
    // activation / weight pieces after the first tile -- same instruction count and waits, a quarter of the bytes)
#ifdef IIR_DBG_THIN_A
#define GLDS_A(src, dst) do { if (kt > kt0_dbg) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src), (LDS_AS void*)(dst), 4, 0, 0); else glds16(src, dst); } while (0)
#else
#define GLDS_A(src, dst) glds16(src, dst)
#endif
#ifdef IIR_DBG_THIN_B
#define GLDS_B(src, dst) do { if (kt > kt0_dbg) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src), (LDS_AS void*)(dst), 4, 0, 0); else glds16(src, dst); } while (0)
#else
#define GLDS_B(src, dst) glds16(src, dst)
#endif
    const int kt0_dbg = 2;
    auto stage = [&](int kt, int buf) {
        f16* as = As + buf * BM * BK;
        f16* bs = Bs + buf * BN * BK;
        const int k0 = kt * BK;
        if (!CONV) {
#pragma unroll
            for (int i = 0; i < A_INST; ++i) GLDS_A(a_src[i] + k0, as + (i * NW + wave) * 8 * BK);
        } else {
#pragma unroll
            for (int i = 0; i < A_INST; ++i) {
                GLDS_A(cv_ptr[i], as + (i * NW + wave) * 8 * BK);
                cv_ptr[i] += cv_inc[i];
            }
            cv_c0 += BK;
            if (cv_c0 == g.Cin) {
                cv_c0 = 0;
                if (++cv_tap < g.ks * g.ks) conv_tap(cv_tap);
            }
        }
        if (!W8) {
#pragma unroll
            for (int i = 0; i < B_INST; ++i) {
                if (B_GROUPS % NW == 0 || i * NW + wave < B_GROUPS) GLDS_B(b_src[i] + k0, bs + (i * NW + wave) * 8 * BK);
                else     // (8-wave builds of the 160-wide tiles: 20 row groups over 8 waves) one 4-byte touch keeps the counted waits uniform
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g.W, (LDS_AS void*)(smem + RING_BYTES + wave * 256), 4, 0, 0);
            }
        } else {
            char* bs8 = (char*)Bs + buf * B_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < B_INST; ++i) {
                if (B_GROUPS % NW == 0 || i * NW + wave < B_GROUPS) {
                    glds16((const char*)b_src[i] + k0, bs8 + (i * NW + wave) * 1024);
                } else {     // a wave without a row group this round still issues ONE vector-memory op, so the counted
                             // vmcnt waits mean the same thing in every wave: a 4-byte touch of the tile's first line
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g.W, (LDS_AS void*)(smem + RING_BYTES + wave * 256), 4, 0, 0);
                }
            }
        }
    };

    // ---- fragment read offsets (bytes inside a tile image) --------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];   // per k-step, for tile row (w*W? + i*16 + frow)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int phys = (s * 4 + fq) ^ (frow & 7);
        a_off[s] = (wm * WM + frow) * 128 + phys * 16;
        if (!W8) b_off[s] = (wn * WN + frow) * 128 + phys * 16;
        else b_off[s] = (wn * WN + frow) * 64 + ((((fq >> 1) + 2 * s) ^ ((frow >> 2) & 3)) * 16) + (fq & 1) * 8;     // (WN % 16 == 0)
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop: ST-deep LDS ring, tiles kt+1 .. kt+ST-2 stay in flight across the barrier ---------
    // (one barrier per K tile; the buffer refilled after the barrier is the one every wave finished
    //  reading before it arrived there)
    const int nk_all = g.K / BK;
    const int kt0 = kz ? nk_all / 2 : 0, nk = g.splitk == 2 ? (kz ? nk_all : nk_all / 2) : nk_all;   // this workgroup's K tiles [kt0, nk)
    if (!LW || loader) {
        if (CONV) conv_seek(kt0);
#pragma unroll
        for (int s = 0; s < ST - 1; ++s)
            if (kt0 + s < nk) stage(kt0 + s, s);
    }
    // LayerNorm statistics of this tile's rows: the producer's per-column-tile partials are fetched here, behind the first K
    // tiles' LDS-DMA, as ONE batch of unconditional loads, merged without a division per group and parked in LDS until the
    // epilogue.  (Measured on the 2048 x 10240 x 1280 GEGLU projection: branchy loads + Chan merges with divisions up front
    // +12.8 us; the partials held in 16 registers across the K loop and merged after it +9.4 us.)
    constexpr int MAXP = 8;
    float2 lnp_in[MAXP];
    const int ln_lt = LW ? tid - 64 * NW : tid;          // LW builds: the loader waves carry them
    if (g.ln_in && ln_lt >= 0 && ln_lt < BM) {
        const int m = min(m0 + ln_lt, g.M - 1);
#pragma unroll
        for (int j = 0; j < MAXP; ++j)              // unconditional (clamped) loads: one straight-line batch; absent groups are masked at the merge
            lnp_in[j] = ((const float2*)g.ln_in)[(long)min(j, g.ln_parts - 1) * g.M + m];

        // equal-count groups: mean = average of the group means, M2 = sum of the group M2 + cols * sum (mean_j - mean)^2
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) sm += j < g.ln_parts ? lnp_in[j].x : 0.f;
        const float inv_p = 1.0f / (float)g.ln_parts, mean = sm * inv_p;
        float m2 = 0.f, dev = 0.f;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const float d = lnp_in[j].x - mean;
            m2 += j < g.ln_parts ? lnp_in[j].y : 0.f;
            dev += j < g.ln_parts ? d * d : 0.f;
        }
        const float var = (m2 + (float)g.ln_part_cols * dev) * inv_p / (float)g.ln_part_cols;
        const float rstd = rsqrtf(var + g.ln_eps);
        rowstat[ln_lt] = make_float2(rstd, -rstd * mean);
        }

    // Software-pipelined across the barrier: the fragments of K-step 0 of tile kt+1 are fetched from LDS while the MFMAs of
    // K-step 1 of tile kt run, and the barrier that admits tile kt+1 sits between the two MFMA groups of tile kt -- so no
    // LDS-read latency is exposed after a barrier (with one workgroup per CU nothing else would hide it).  The barrier
    // also carries lgkmcnt(0): every wave's reads of tile kt are complete, so its buffer is refilled right away
    // (tile kt+ST), one tile further ahead than a refill-then-read order allows.
    using BF = std::conditional_t<W8, long, E8>;       // weight fragment: 8 fp8 bytes or 8 halves
    auto frags = [&](int buf, int s, E8 (&af)[MI], BF (&bf)[NI]) {
        const char* as = (const char*)(As + buf * BM * BK);
        const char* bs = (const char*)Bs + buf * B_STAGE_BYTES;
#ifdef IIR_DBG_THIN_READS      // (timing experiment of DESIGN.md 5.10, results wrong: two fragment reads per half step instead of MI + NI)
        af[0] = *(const E8*)(as + a_off[s]);
        bf[0] = *(const BF*)(bs + b_off[s]);
        for (int i = 1; i < MI; ++i) af[i] = af[0];
        for (int j = 1; j < NI; ++j) bf[j] = bf[0];
        return;
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const E8*)(as + a_off[s] + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < NI; ++j) bf[j] = *(const BF*)(bs + b_off[s] + j * 16 * (W8 ? 64 : 128));
    };
    auto mma = [&](const E8 (&af)[MI], const BF (&bf)[NI]) {
#ifdef IIR_DBG_NO_MFMA       // (trigger bisection of DESIGN.md 5.8: same loads, barriers and LDS reads, no matrix instructions; results are wrong)
        for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) asm volatile("" :: "v"(af[i]), "v"(bf[j]));
        return;
#endif
        if constexpr (F8) {
            typedef long l2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const l2 a2 = __builtin_bit_cast(l2, af[i]);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const l2 b2 = __builtin_bit_cast(l2, bf[j]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], acc[i][j], 0, 0, 0);
                }
            }
        } else if constexpr (!W8) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = ET<E>::mfma16(bf[j], af[i], acc[i][j]);
        } else {
            long a8[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {          // 8 halves -> 8 fp8 (E4M3, round to nearest even, saturating), k order kept
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                s16x2 lo = {0, 0}, hi = {0, 0};
                lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){af[i][0], af[i][1]}, 1.0f, false);
                lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(lo, (f16x2){af[i][2], af[i][3]}, 1.0f, true);
                hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, (f16x2){af[i][4], af[i][5]}, 1.0f, false);
                hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(hi, (f16x2){af[i][6], af[i][7]}, 1.0f, true);
                a8[i] = (long)(unsigned)__builtin_bit_cast(int, lo) | ((long)__builtin_bit_cast(int, hi) << 32);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf[j], a8[i], acc[i][j], 0, 0, 0);
        }
    };
    auto admit = [&](int tiles_after) {     // wait until only `tiles_after` later tiles are still in flight, all LDS reads done, barrier
        if (ST >= 5 && tiles_after >= 3) wait_vm_lgkm_and_barrier<(ST >= 5 ? 3 * LOADS : 0)>();
        else if (ST >= 4 && tiles_after >= 2) wait_vm_lgkm_and_barrier<(ST >= 4 ? 2 * LOADS : 0)>();
        else if (ST >= 3 && tiles_after >= 1) wait_vm_lgkm_and_barrier<(ST >= 3 ? LOADS : 0)>();
        else wait_vm_lgkm_and_barrier<0>();
    };
    // what the epilogue needs per column (bias, LayerNorm column sums) is requested ahead of the LAST tile's MFMAs: fetched at
    // the point of use, these loads' latency was exposed once per launch, on every wave at the same moment
    f32x4 pre_c1[NI];
    E4 pre_c0[NI];
    auto preload_cols = [&]() {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nc = min(n0 + wn * WN + j * 16 + fq * 4, g.N - 4);       // clamped: lanes past the edge fetch a valid quad they never use
            pre_c1[j] = g.ln_in ? *(const f32x4*)(g.ln_colsum + nc) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g.bias) pre_c0[j] = *(const E4*)((const E*)g.bias + nc);
            else for (int t = 0; t < 4; ++t) pre_c0[j][t] = (E)0.f;
        }
    };
    E8 a0[MI], a1[MI];
    BF b0[NI], b1[NI];
    // Loop form (round 3).  Per K tile the instruction order is unchanged -- [reads of half step 1][MFMAs of half step 0]
    // [barrier: next tile landed][reads of the next tile's half step 0][MFMAs of half step 1] -- but the loop now BEGINS at the
    // barrier, and the barrier's `lgkmcnt(0)` is a compiler-visible builtin.  With fragment reads pending across the back edge
    // (the first form) hipcc's wait insertion fell back to `lgkmcnt(0)` in front of every MFMA group, i.e. it also waited for the
    // reads issued one instruction earlier for the NEXT group: one exposed LDS round trip per half step.  With nothing pending at
    // the loop head the same reads get counted waits (`lgkmcnt(7)`): the prefetched group stays in flight under the MFMAs.
    auto admit2 = [&](int tiles_after) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): every fragment read of the finished tile has returned
        if (ST >= 5 && tiles_after >= 3) wait_vm_and_barrier<(ST >= 5 ? 3 * LOADS : 0)>();
        else if (ST >= 4 && tiles_after >= 2) wait_vm_and_barrier<(ST >= 4 ? 2 * LOADS : 0)>();
        else if (ST >= 3 && tiles_after >= 1) wait_vm_and_barrier<(ST >= 3 ? LOADS : 0)>();
        else wait_vm_and_barrier<0>();
    };
    if constexpr (!LW) {
        {
            const int rem = nk - 1 - kt0;
            admit(rem < ST - 2 ? rem : ST - 2);
            if (kt0 + ST - 1 < nk) stage(kt0 + ST - 1, ST - 1);
            frags(0, 0, a0, b0);
            IIR_PIN();
            frags(0, 1, a1, b1);
            IIR_PIN();
            mma(a0, b0);
        }
        int cur = 0;
        for (int kt = kt0 + 1; kt < nk; ++kt) {                 // kt = the tile this iteration admits and starts
            const int rem = nk - 1 - kt;                        // tiles that exist after kt
            admit2(rem < ST - 2 ? rem : ST - 2);
            if (kt - 1 + ST < nk) stage(kt - 1 + ST, cur);      // tile kt-1's buffer is free: every wave finished reading it
            cur = cur + 1 == ST ? 0 : cur + 1;
            frags(cur, 0, a0, b0);
            IIR_PIN();
            mma(a1, b1);                                        // tile kt-1, half step 1
            frags(cur, 1, a1, b1);
            IIR_PIN();
            mma(a0, b0);                                        // tile kt, half step 0
        }
        preload_cols();
        mma(a1, b1);
    } else if (loader) {
        // loader waves: same barriers, same counted waits, no LDS reads and no MFMAs
        {
            const int rem = nk - 1 - kt0;
            admit(rem < ST - 2 ? rem : ST - 2);
            if (kt0 + ST - 1 < nk) stage(kt0 + ST - 1, ST - 1);
        }
        int cur = 0;
        for (int kt = kt0; kt + 1 < nk; ++kt) {
            const int rem = nk - 2 - kt;
            admit(rem < ST - 2 ? rem : ST - 2);
            if (kt + ST < nk) stage(kt + ST, cur);
            cur = cur + 1 == ST ? 0 : cur + 1;
        }
    } else {
        // compute waves: they issue no vector-memory operation in the loop, so the vmcnt part of `admit` never waits
        admit(0);
        frags(0, 0, a0, b0);
        IIR_PIN();
        frags(0, 1, a1, b1);
        IIR_PIN();
        mma(a0, b0);
        int cur = 0;
        for (int kt = kt0 + 1; kt < nk; ++kt) {
            admit2(0);
            cur = cur + 1 == ST ? 0 : cur + 1;
            frags(cur, 0, a0, b0);
            IIR_PIN();
            mma(a1, b1);
            frags(cur, 1, a1, b1);
            IIR_PIN();
            mma(a0, b0);
        }
        preload_cols();
        mma(a1, b1);
    }

    // ---- IIR_EPI_XATTN: `to_q` and the decoupled text / IP cross-attention of
    // TA_IPAttnProcessor2_0 (module/ip_adapter/attention_processor.py:1140-1192) in one launch.  The 64 x 128 tile is q for 64 rows
    // of one image and two heads; wave (wm, wn) holds rows 32 wm .. +32 of head wn in its accumulators.  Nothing is re-laid-out:
    // an accumulator lane (frow, fq) holds q[frow][16 j + 4 fq + t] -- as an MFMA operand that is "8 values of the contraction
    // index" for ANY consistent order of that index, so the K fragment is read in the same permuted order (two 8-byte pieces
    // per lane), S = q K^T lands as (row frow, keys 16 kb + 4 fq + t), and P feeds P V the same way with V^T read in matching
    // key order.  Both softmaxes (text keys, IP keys) are exact two-pass ones over <= 80 / 64 keys held in registers; each is
    // normalised before P V, so one accumulator sums text and IP outputs (`hidden_states + ip_hidden_states`, :1192, scale 1).
    // The K / V^T images (72 KiB: exactly the idle ring) are staged by LDS-DMA after the K loop; requesting them into registers at
    // kernel entry instead (182 VGPRs) measured SLOWER (22.3 vs 20.6 us per level-2 launch).
    bool xdone = false;
    if constexpr (BM == 64 && BN == 128 && ST == 3 && !CONV && !W8 && !LW && WAVES_M == 2 && std::is_same<E, f16>::value) {
    if (g.xa_on) {
        constexpr int KROWS = 144, KIMG = KROWS * 128, VROW = 288, VIMG = 64 * VROW;     // per head: K image 18 KiB, V^T image 18 KiB
        static_assert(2 * KIMG + 2 * VIMG <= RING_BYTES, "K / V images of two heads must fit the idle ring");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading the ring
        const int img = m0 / g.xa_tq, hd0 = n0 >> 6;
        char* kl = smem;
        char* vl = smem + 2 * KIMG;
        const int tk0 = g.xa_tk[0], tk1 = g.xa_tk[1];
        const int vch0 = (tk0 + 7) >> 3, vch1 = (tk1 + 7) >> 3;               // readable 16-byte chunks of a V^T row
#ifndef IIR_DBG_XA_NOSTAGE      // (timing experiments, results wrong: -DIIR_DBG_XA_NOSTAGE / _NOCOMPUTE / _NOSOFTMAX price the parts of this tail)
#pragma unroll
        for (int it = 0; it < 9; ++it) {          // K rows: [0, 80) text keys, [80, 144) IP keys; rows past Tkv repeat the last key (masked below)
            const int s = (it * 4 + wave) * 64 + lane, hh = s >= 1152 ? 1 : 0, sl = s - hh * 1152;
            const int row = sl >> 3, ch = (sl & 7) ^ (row & 7);
            const bool ip = row >= 80;
            const int key = ip ? min(row - 80, tk1 - 1) : min(row, tk0 - 1);
            const f16* src = (ip ? g.xa_k[1] + (long)img * g.xa_kb[1] + (long)key * g.xa_ldk[1]
                                 : g.xa_k[0] + (long)img * g.xa_kb[0] + (long)key * g.xa_ldk[0]) + (hd0 + hh) * 64 + ch * 8;
            glds16(src, kl + (it * 4 + wave) * 1024);
        }
#pragma unroll
        for (int it = 0; it < 9; ++it) {          // V^T rows d = 0..63 of each head: 10 chunks of text keys, 8 chunks of IP keys
            const int s = (it * 4 + wave) * 64 + lane, hh = s >= 1152 ? 1 : 0, sl = s - hh * 1152;
            const int d = sl / 18, c = sl - d * 18;
            const bool ip = c >= 10;
            const int cc = ip ? min(c - 10, vch1 - 1) : min(c, vch0 - 1);
            const long rowv = (long)((hd0 + hh) * 64 + d);
            const f16* src = (ip ? g.xa_vt[1] + rowv * g.xa_ldvt[1] + (long)img * g.xa_vb[1]
                                 : g.xa_vt[0] + rowv * g.xa_ldvt[0] + (long)img * g.xa_vb[0]) + cc * 8;
            glds16(src, vl + (it * 4 + wave) * 1024);
        }
#endif
        // q = what the plain epilogue would have stored (LayerNorm fold, bias, fp16 rounding), as MFMA operands
        typedef E E4v __attribute__((ext_vector_type(4)));
        E8 qf[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float2 rs = g.ln_in ? rowstat[wm * WM + i * 16 + frow] : make_float2(1.f, 0.f);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    qf[i][j >> 1][(j & 1) * 4 + t] = (E)fmaf(acc[i][j][t], rs.x, fmaf(rs.y, pre_c1[j][t], (float)pre_c0[j][t]));
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");         // K / V images complete (waiting for K alone here and for V^T before P V measured no better)
#ifndef IIR_DBG_XA_NOCOMPUTE
        // S = q K^T: 9 key blocks of 16 (5 text, 4 IP), contraction over d in the accumulator's order
        f32x4 sa[MI][9];
        const char* kh = kl + wn * KIMG;
#pragma unroll
        for (int kb = 0; kb < 9; ++kb) {
            const int kr = kb * 16 + frow;
            const char* krow = kh + kr * 128 + (fq & 1) * 8;
            E8 kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const E4v lo = *(const E4v*)(krow + (((4 * ks + (fq >> 1)) ^ (kr & 7)) << 4));
                const E4v hi = *(const E4v*)(krow + (((4 * ks + 2 + (fq >> 1)) ^ (kr & 7)) << 4));
                for (int t = 0; t < 4; ++t) { kf[ks][t] = lo[t]; kf[ks][4 + t] = hi[t]; }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                z = ET<E>::mfma16(kf[0], qf[i][0], z);
                sa[i][kb] = ET<E>::mfma16(kf[1], qf[i][1], z);
            }
        }
        // softmax per segment (q carries scale x log2 e: p = exp2(s - max)), normalised, as the P operand of P V
        E8 pf[MI][5];
#ifdef IIR_DBG_XA_NOSOFTMAX
        for (int i = 0; i < MI; ++i) for (int kp = 0; kp < 5; ++kp) for (int t = 0; t < 4; ++t) { pf[i][kp][t] = (E)sa[i][2 * kp][t]; pf[i][kp][4 + t] = (E)sa[i][kp < 4 ? 2 * kp + 1 : 8][t]; }
#else
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
            for (int kb = 0; kb < 9; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int key = (kb < 5 ? kb * 16 : kb * 16 - 80) + fq * 4 + t;
                    const bool ok = key < (kb < 5 ? tk0 : tk1);
                    sa[i][kb][t] = ok ? sa[i][kb][t] : -INFINITY;
                    mx[kb < 5 ? 0 : 1] = fmaxf(mx[kb < 5 ? 0 : 1], sa[i][kb][t]);
                }
#pragma unroll
            for (int sgm = 0; sgm < 2; ++sgm) {
                mx[sgm] = fmaxf(mx[sgm], __shfl_xor(mx[sgm], 16, 64));
                mx[sgm] = fmaxf(mx[sgm], __shfl_xor(mx[sgm], 32, 64));
            }
            float sm[2] = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 9; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float pv = __builtin_amdgcn_exp2f(sa[i][kb][t] - mx[kb < 5 ? 0 : 1]);     // masked keys: exp2(-inf) = 0
                    sa[i][kb][t] = pv;
                    sm[kb < 5 ? 0 : 1] += pv;
                }
#pragma unroll
            for (int sgm = 0; sgm < 2; ++sgm) {
                sm[sgm] += __shfl_xor(sm[sgm], 16, 64);
                sm[sgm] += __shfl_xor(sm[sgm], 32, 64);
                sm[sgm] = __builtin_amdgcn_rcpf(sm[sgm]);
            }
#pragma unroll
            for (int kp = 0; kp < 5; ++kp)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ka = 2 * kp, kb2 = 2 * kp + 1;
                    pf[i][kp][t] = (E)(sa[i][ka][t] * sm[ka < 5 ? 0 : 1]);
                    pf[i][kp][4 + t] = kb2 < 9 ? (E)(sa[i][kb2 < 9 ? kb2 : 8][t] * sm[kb2 < 5 ? 0 : 1]) : (E)0.f;
                }
        }
#endif
        // O = P V: contraction over keys in the order P holds them (key column 16 kb + 4 fq + t: IP keys start at column 80)
        const char* vh = vl + wn * VIMG;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const char* vrow = vh + (j * 16 + frow) * VROW + fq * 8;
#pragma unroll
            for (int i = 0; i < MI; ++i) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < 5; ++kp) {
                const E4v lo = *(const E4v*)(vrow + (2 * kp) * 32);
                const E4v hi = *(const E4v*)(vrow + (kp < 4 ? 2 * kp + 1 : 8) * 32);       // (the ninth block's partner: P is zero there)
                E8 vf;
                for (int t = 0; t < 4; ++t) { vf[t] = lo[t]; vf[4 + t] = hi[t]; }
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = ET<E>::mfma16(vf, pf[i][kp], acc[i][j]);
            }
        }
#endif
#pragma unroll
        for (int j = 0; j < NI; ++j) {            // the tile now holds finished values: phase 1 adds nothing
            pre_c1[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < 4; ++t) pre_c0[j][t] = (E)0.f;
        }
        xdone = true;
    }
    }

    // ---- epilogue, phase 1: registers -> LDS.  A lane holds row m = ..+frow and 4 consecutive columns n = ..+4*fq+{0..3};
    // stored straight to memory that is 16 rows x 32 B per instruction (16 B for the paired epilogues), which made the
    // write-out -- not the K loop -- the fixed cost of every launch.  So the finished fp16 tile (bias / row bias /
    // activation / GEGLU / SFT applied) goes through the now idle ring, and phase 2 writes whole rows, 16 B per lane,
    // adding the residual from equally coalesced loads.  (fp16 rounding before the residual add = torch's own order:
    // the Linear / Conv output is an fp16 tensor before `+ residual`.)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is done reading the ring

    // ---- split-K: whichever of a tile's two workgroups finishes LAST adds the other's fp32 partial and runs the epilogue
    // (a + b = b + a in fp32, so the result does not depend on which one that is).  Hand-off: all stores of the slab
    // drained by every wave -> workgroup barrier -> one agent-scope release -> relaxed agent-scope ticket; the reducer
    // does one agent-scope acquire, then plain loads (cdna_hip_programming.md, in-launch split-K reduction recipe).
    if (!LW && g.splitk == 2) {
        const int tile_id = tn * g.tiles_m + tm;
        float* mine = g.sk_slabs + ((long)tile_id * 2 + kz) * (BM * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) *(f32x4*)(mine + ((i * NI + j) * NT + tid) * 4) = acc[i][j];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)(smem + RING_BYTES);     // first word of the prefetch scratch (unused until phase 2)
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *flag = __hip_atomic_fetch_add(g.sk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*flag == 0) return;                                              // first to arrive: the partner finishes the tile
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g.sk_cnt[tile_id] = 0;                                            // ready for the next launch on this stream
        }
        __syncthreads();
        const float* other = g.sk_slabs + ((long)tile_id * 2 + (1 - kz)) * (BM * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] += *(const f32x4*)(other + ((i * NI + j) * NT + tid) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (g.c_f32) {      // fp32 output straight from the accumulators (a lane holds 4 consecutive columns of one row)
        float* c32 = (float*)g.C;
        if (loader) return;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * WM + i * 16 + frow;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn * WN + j * 16 + fq * 4;
                if (m < g.M && n < g.N) *(f32x4*)(c32 + (long)m * g.ldc + n) = acc[i][j] * g.out_scale;
            }
        }
        return;
    }
    const bool paired = g.epi != IIR_EPI_PLAIN;
    const int cs = (paired ? BN : 2 * BN) + 32;                          // tile row stride in bytes: odd multiple of 32 mod 256
    char* ct = smem;
    if (!loader) {
    // column-outer, row-inner: what depends on the column only (bias, fp8 scale, LayerNorm column sum) is fetched once per
    // 16-column group and lane, not once per accumulator tile (with `ln_in` the per-tile form cost the GEGLU projection 26 us)
    int lrs[MI], ms[MI];
    float2 rss[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        lrs[i] = wm * WM + i * 16 + frow;                  // row inside the tile
        ms[i] = min(m0 + lrs[i], g.M - 1);                 // tail rows: computed from clamped operands, dropped in phase 2
        rss[i] = (g.ln_in && !xdone) ? rowstat[lrs[i]] : make_float2(1.f, 0.f);
    }
    if (!paired) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int lc = wn * WN + j * 16 + fq * 4;
            const int n = n0 + lc;
            if (n >= g.N) continue;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
            if (W8 || F8) { const f32x4 ws = *(const f32x4*)(g.wscale + n); for (int t = 0; t < 4; ++t) sc[t] = ws[t] * g.a_scale; }
            for (int t = 0; t < 4; ++t) { c1[t] = pre_c1[j][t]; c0[t] = (float)pre_c0[j][t]; }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int lr = lrs[i];
                float v[4];
                for (int t = 0; t < 4; ++t) v[t] = (W8 || F8) ? acc[i][j][t] * sc[t] : acc[i][j][t];
                for (int t = 0; t < 4; ++t) v[t] = fmaf(v[t], rss[i].x, fmaf(rss[i].y, c1[t], c0[t]));     // (1, 0) without ln_in
                if (g.rowbias) { E4 b = *(const E4*)(g.rowbias + (long)(ms[i] / g.rows_per_rb) * g.ldrb + n); for (int t = 0; t < 4; ++t) v[t] += (float)b[t]; }
                if (g.act == IIR_ACT_SILU) for (int t = 0; t < 4; ++t) v[t] = silu_f(v[t]);
                else if (g.act == IIR_ACT_GELU) for (int t = 0; t < 4; ++t) v[t] = gelu_erf_f(v[t]);
                else if (g.act == IIR_ACT_QUICKGELU) for (int t = 0; t < 4; ++t) v[t] = v[t] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v[t]));
                E4 o;
                for (int t = 0; t < 4; ++t) o[t] = (E)v[t];
                *(E4*)(ct + lr * cs + lc * 2) = o;
            }
        }
    } else {
        // paired columns: in every 16-column group of the (row-permuted) weight the first 8 are the "value" rows and
        // the next 8 their partners (gate for GEGLU; beta for SFT).  A lane holds 4 consecutive columns, so value
        // lanes (fq = 0,1) fetch their partner from lane + 32 (fq + 2) with one cross-half exchange per register.
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WN + j * 16 + fq * 4;           // permuted column held by this lane
            const bool in_n = n < g.N;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
            if ((W8 || F8) && in_n) { const f32x4 ws = *(const f32x4*)(g.wscale + n); for (int t = 0; t < 4; ++t) sc[t] = ws[t] * g.a_scale; }
            for (int t = 0; t < 4; ++t) { c1[t] = pre_c1[j][t]; c0[t] = (float)pre_c0[j][t]; }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int lr = lrs[i];
                float a[4];
                for (int t = 0; t < 4; ++t) a[t] = (W8 || F8) ? acc[i][j][t] * sc[t] : acc[i][j][t];
                for (int t = 0; t < 4; ++t) a[t] = fmaf(a[t], rss[i].x, fmaf(rss[i].y, c1[t], c0[t]));
                float b[4];
                for (int t = 0; t < 4; ++t) b[t] = __shfl_xor(a[t], 32, 64);     // all lanes take part in the exchange
                if (g.epi == IIR_EPI_GEGLU) {
                    // value * gelu(gate): the erf evaluation (14 VALU + v_rcp + v_exp per element) is the cost of this epilogue
                    // -- kbench 2048x10240x1280: 865 TFLOP/s plain, 735 with value lanes alone doing all four columns -- so
                    // both lanes of a pair work: the value lane finishes columns 0,1 of the quad, its gate lane columns 2,3.
                    if (!in_n) continue;
                    const bool gate = fq >= 2;
                    const int lco = (wn * WN + j * 16) / 2 + (fq & 1) * 4 + (gate ? 2 : 0);
                    const float v0 = gate ? b[2] : a[0], v1 = gate ? b[3] : a[1], g0 = gate ? a[2] : b[0], g1 = gate ? a[3] : b[1];
                    typedef E E2 __attribute__((ext_vector_type(2)));
                    E2 o2 = {(E)(v0 * gelu_erf_f(g0)), (E)(v1 * gelu_erf_f(g1))};
                    *(E2*)(ct + lr * cs + lco * 2) = o2;
                    continue;
                }
                if (fq >= 2 || !in_n) continue;
                const int lco = (wn * WN + j * 16) / 2 + fq * 4;        // output column inside the tile
                long mr = ms[i];
                if (CONV && g.res_img_rows) { const int hw = g.Ho * g.Wo, img = ms[i] / hw; mr = (long)img * g.res_img_rows + (ms[i] - img * hw); }
                E4 o;
                {   // IIR_EPI_SFT: h * (gamma + 1) + beta, h from `res`
                    E4 h = *(const E4*)(g.res + mr * g.ldr + n0 / 2 + lco);
                    for (int t = 0; t < 4; ++t) o[t] = (E)((float)h[t] * (a[t] + 1.0f) + b[t]);
                }
                *(E4*)(ct + lr * cs + lco * 2) = o;
            }
        }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tile complete

    // ---- phase 2: LDS -> memory, one 16-byte chunk of a row per lane: residual loads, add, then the weight prefetch
    // for the launches that follow (see iir_gemm_desc.prefetch: 4-byte LDS-DMA touches, clamped into the range; no VGPR
    // destination, the data lands in a scratch KiB behind the ring and is never read), then the stores.
    auto touch_next_weights = [&]() {
#ifdef IIR_DBG_NO_TOUCH      // (trigger bisection of DESIGN.md 5.8: build without the prefetch touches)
        return;
#endif
        const int per = (g.pf_lines + (int)gridDim.x - 1) / (int)gridDim.x;
        const long l0 = (long)blockIdx.x * per, last = g.pf_lines - 1;
        char* scratch = smem + RING_BYTES + wave_all * 256;
#pragma unroll
        for (int i = 0; i < PF_TOUCHES; ++i) {
            long l = l0 + min(tid + i * NT, per - 1);
            if (l > last) l = last;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(g.pf + l * 128), (LDS_AS void*)scratch, 4, 0, 0);
        }
    };
    auto write_out = [&](auto cpr_tag) {
        constexpr int CPR = decltype(cpr_tag)::value;                    // 16-byte chunks per tile row
        constexpr int TOTAL = BM * CPR, CH = (TOTAL + NT - 1) / NT;
        const int no_tile = paired ? n0 / 2 : n0, No = paired ? g.N / 2 : g.N;
        const bool use_res = g.res && !paired;
        const bool remap = CONV && (g.y_img_rows | g.res_img_rows);
        // whole tile inside the matrix, rows 16-byte addressable: straight-line code, no predicates on the loads
        const bool tr_tile = g.Ct && no_tile >= g.tr_from;                 // this whole tile lies in the transposed column range
        const bool tr_mixed = g.Ct && !tr_tile && no_tile + CPR * 8 > g.tr_from;
        const bool inside = m0 + BM <= g.M && no_tile + CPR * 8 <= No && !remap;
        const bool fast = !tr_tile && !tr_mixed && g.c_vec && (!use_res || g.r_vec) && inside;
        if (tr_tile && inside && g.ct_vec && !use_res) {
            // transposed write-out (the V third of a fused q|k|v projection becomes the V^T image the attention kernel reads):
            // a lane gathers 8 consecutive rows of one tile column from LDS and stores them as 16 contiguous bytes of Ct.
            constexpr int RC = BM / 8, TOTAL_T = (CPR * 8) * RC, CH_T = (TOTAL_T + NT - 1) / NT;
            touch_next_weights();
#pragma unroll
            for (int k = 0; k < CH_T; ++k) {
                const int idx = tid + k * NT;
                if (TOTAL_T % NT != 0 && idx >= TOTAL_T) break;
                const int c = idx / RC, rc = idx - c * RC;
                E8 o;
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] = (E)((float)*(const E*)(ct + (rc * 8 + t) * cs + c * 2) * g.out_scale);
                *(E8*)(g.Ct + (long)(no_tile + c - g.tr_from) * g.ldct + m0 + rc * 8) = o;
            }
            return;
        }
        // (hipcc drains every LDS-DMA in flight at the next use of an ordinary load's result, so the touches are issued
        //  after the residuals have been consumed and ahead of the stores, which need no wait)
        if constexpr (GN_OUT_FITS && CPR == BN / 8) {
        if (fast && g.gn_out && !paired) {
            // Write-out with GroupNorm statistics: thread = (row group rg, 16-byte column chunk cc), so
            // all rows a thread stores share its 8 channels and their sums stay in registers (taken about the thread's first
            // sample: no E[x^2] - mean^2 cancellation); per 64-row slab the RG row groups of a column are merged through LDS in
            // a fixed order by one thread per column -> (mean, M2) of the 64 stored (rounded) values of that channel.  The
            // finalize kernel merges slabs x channels of an (image, group).  No atomics: bit-reproducible.
            constexpr int RG = NT / CPR, RPT = (64 + RG - 1) / RG;            // rows per thread and slab
            const int cc = tid % CPR, rg = tid / CPR;
            const bool active = rg < RG;
            float2* part = (float2*)(smem + LNP_OFF);                         // [RG][BN]
            touch_next_weights();
#pragma unroll 1
            for (int half = 0; half < BM / 64; ++half) {
                E8 rr[RPT];
                if (use_res) {
#pragma unroll
                    for (int j = 0; j < RPT; ++j) {
                        const int r = half * 64 + min(rg + j * RG, 63);
                        rr[j] = *(const E8*)(g.res + (long)(m0 + r) * g.ldr + no_tile + min(cc, CPR - 1) * 8);
                    }
                }
                float k0[8], s8[8], q8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { k0[t] = 0.f; s8[t] = 0.f; q8[t] = 0.f; }
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const int rl = rg + j * RG;
                    if (active && rl < 64) {
                        const int r = half * 64 + rl;
                        const E8 v = *(const E8*)(ct + r * cs + cc * 16);
                        E8 o;
                        for (int t = 0; t < 8; ++t) o[t] = (E)(((float)v[t] + (use_res ? (float)rr[j][t] : 0.f)) * g.out_scale);
                        iir::store16(g.C, ((long)(m0 + r) * g.ldc + no_tile + cc * 8) * 2, o, g.st_wt != 0);
                        for (int t = 0; t < 8; ++t) {
                            const float f = (float)o[t];
                            if (j == 0) k0[t] = f;
                            const float d = f - k0[t];
                            s8[t] += d; q8[t] = fmaf(d, d, q8[t]);
                        }
                        ++cnt;
                    }
                }
                if (active) {
                    const float inv = 1.0f / (float)cnt;                      // (cnt >= 1: rg < RG <= 64)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        part[rg * BN + cc * 8 + t] = make_float2(k0[t] + s8[t] * inv, fmaxf(q8[t] - s8[t] * s8[t] * inv, 0.f));
                }
                __syncthreads();
                if (tid < BN) {                                               // one thread per tile column: merge the row groups in order
                    float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll 4
                    for (int q = 0; q < RG; ++q) {
                        const float2 pq = part[q * BN + tid];
                        const float nb = (float)((64 - q + RG - 1) / RG);     // rows row group q holds in a slab
                        const float tot = n + nb, d = pq.x - mean;
                        mean += d * (nb / tot);
                        m2 += pq.y + d * d * (n * nb / tot);
                        n = tot;
                    }
                    ((float2*)g.gn_out)[(long)(m0 / 64 + half) * g.N + n0 + tid] = make_float2(mean, m2);
                }
                __syncthreads();
            }
            return;
        }
        }
        if (fast) {
            E8 o[CH];
            if (use_res) {
                E8 rr[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = min(tid + k * NT, TOTAL - 1), r = c / CPR, cc = c - r * CPR;
                    rr[k] = *(const E8*)(g.res + (long)(m0 + r) * g.ldr + no_tile + cc * 8);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = min(tid + k * NT, TOTAL - 1), r = c / CPR, cc = c - r * CPR;
                    const E8 v = *(const E8*)(ct + r * cs + cc * 16);
                    for (int t = 0; t < 8; ++t) o[k][t] = (E)(((float)v[t] + (float)rr[k][t]) * g.out_scale);
                }
            } else {
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const int c = min(tid + k * NT, TOTAL - 1), r = c / CPR, cc = c - r * CPR;
                    const E8 v = *(const E8*)(ct + r * cs + cc * 16);
                    for (int t = 0; t < 8; ++t) o[k][t] = (E)((float)v[t] * g.out_scale);
                }
            }
            touch_next_weights();
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int c = tid + k * NT, r = c / CPR, cc = c - r * CPR;
                if (TOTAL % NT != 0 && c >= TOTAL) break;
                if constexpr (std::is_same<E, f16>::value) {
                    if (g.c_fp8) { *(long*)((char*)g.C + (long)(m0 + r) * g.ldc + no_tile + cc * 8) = iir_fp8x8(o[k]); continue; }
                }
                iir::store16(g.C, ((long)(m0 + r) * g.ldc + no_tile + cc * 8) * 2, o[k], g.st_wt != 0);
            }
            if constexpr (LN_OUT_FITS) {
                if (g.ln_out) {
                    // the rows just written feed a LayerNorm: leave (mean, M2) of this tile's BN columns of every row, taken
                    // from the ROUNDED values (what the next GEMM reads).  Per 8-column chunk in registers, then one thread per
                    // row merges the row's CPR chunks in a fixed order (the merge weights are compile-time constants).
                    float2* lnp = (float2*)(smem + LNP_OFF);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int c = tid + k * NT;
                        if (TOTAL % NT != 0 && c >= TOTAL) break;
                        float f[8], sum = 0.f, q = 0.f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) { f[t] = (float)o[k][t]; sum += f[t]; }
                        const float mu = sum * 0.125f;
#pragma unroll
                        for (int t = 0; t < 8; ++t) { const float d = f[t] - mu; q = fmaf(d, d, q); }
                        lnp[c] = make_float2(mu, q);                                 // c = r * CPR + cc
                    }
                    __syncthreads();
                    if (tid < BM) {
                        float2 acc2 = lnp[tid * CPR];
                        float mean = acc2.x, m2 = acc2.y;
#pragma unroll
                        for (int j = 1; j < CPR; ++j) {
                            const float2 pj = lnp[tid * CPR + j];
                            const float d = pj.x - mean;
                            mean += d * (1.0f / (float)(j + 1));
                            m2 += pj.y + d * d * (8.0f * (float)j / (float)(j + 1));
                        }
                        ((float2*)g.ln_out)[(long)tn * g.M + m0 + tid] = make_float2(mean, m2);
                    }
                }
            }
        } else {
            // ragged edge / unaligned rows / per-image row remap: element-wise, speed irrelevant
            touch_next_weights();
#pragma unroll 1
            for (int c = tid; c < TOTAL; c += NT) {
                const int r = c / CPR, cc = c - r * CPR;
                const int m = m0 + r, n = no_tile + cc * 8;
                if (m >= g.M || n >= No) continue;
                long mc = m, mr = m;
                if (remap) {
                    const int hw = g.Ho * g.Wo, img = m / hw, rem = m - img * hw;
                    if (g.y_img_rows) mc = (long)img * g.y_img_rows + rem;
                    if (g.res_img_rows) mr = (long)img * g.res_img_rows + rem;
                }
                const E* tp = (const E*)(ct + r * cs + cc * 16);
#pragma unroll 1
                for (int t = 0; t < 8 && n + t < No; ++t) {
                    float v = (float)tp[t];
                    if (g.Ct && n + t >= g.tr_from) { ((E*)g.Ct)[(long)(n + t - g.tr_from) * g.ldct + mc] = (E)(v * g.out_scale); continue; }
                    if (use_res) v += (float)((const E*)g.res)[mr * g.ldr + n + t];
                    ((E*)g.C)[mc * g.ldc + n + t] = (E)(v * g.out_scale);
                }
            }
        }
    };
    if (paired) write_out(std::integral_constant<int, BN / 16>{});
    else write_out(std::integral_constant<int, BN / 8>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA touches must land before the LDS is released
}

template <typename E, int BM, int BN, int ST, int WAVES_M = 2, bool W8 = false, bool LW = false, bool F8 = false>
int launch_t(const Geo& g0, bool conv, hipStream_t stream) {
    Geo g = g0;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    static const size_t dbg_pad = getenv("IIR_DBG_LDS_PAD") ? (size_t)atoi(getenv("IIR_DBG_LDS_PAD")) : 0;     // (5.8 bisection: force one workgroup per CU)
    const size_t lds = ST * (BM * 128 + (W8 ? BN * 64 : BN * 128)) + 256 * WAVES_M * 2 * (LW ? 2 : 1) + BM * 8 + dbg_pad;   // ring (reused as the output tile) + prefetch scratch (256 B per wave) + LayerNorm row statistics
    if (g.c_fp8) {          // fp8 output: stored by the straight-line write-out only (whole tiles, 8-byte addressable rows)
        if (conv || g.M % BM || g.N % BN || g.ldc % 8 || (uintptr_t)g.C % 8 || g.Ct || g.c_f32 || g.ln_out || g.gn_out || g.res_img_rows || g.y_img_rows ||
            (g.res && !g.r_vec) || g.dtype != IIR_DT_F16) return IIR_EINVAL;
        g.c_vec = 1; g.st_wt = 0;
    }
    if (g.gn_out) {         // producer of GroupNorm partials: whole tiles, 16-byte rows, plain epilogue, rows of an image tile-aligned (caller)
        constexpr int RINGB = ST * (BM * 128 + (W8 ? BN * 64 : BN * 128)), NTH = 128 * WAVES_M * (LW ? 2 : 1);
        constexpr bool fits = BM % 64 == 0 && BN <= NTH && (BM * (2 * BN + 32) + 15) / 16 * 16 + (NTH / (BN / 8)) * BN * 8 <= RINGB;
        if (!fits || g.epi != IIR_EPI_PLAIN || g.c_f32 || g.Ct || g.splitk == 2 || g.ln_out || g.M % BM || g.N % BN || !g.c_vec || (g.res && !g.r_vec) ||
            (conv && (g.y_img_rows | g.res_img_rows)))
            return IIR_EINVAL;
    }
    if (g.ln_out) {         // producer of LayerNorm partials: whole tiles, 16-byte rows, plain epilogue (see the kernel's fast write-out path)
        constexpr bool fits = (BM * (2 * BN + 32) + 15) / 16 * 16 + BM * (BN / 8) * 8 <= ST * (BM * 128 + (W8 ? BN * 64 : BN * 128));
        if (!fits || conv || g.epi != IIR_EPI_PLAIN || g.c_f32 || g.Ct || g.splitk == 2 || g.M % BM || g.N % BN || !g.c_vec || (g.res && !g.r_vec))
            return IIR_EINVAL;
    }
    // pick the XCD partition (xm x 8/xm rectangles of the tile grid) with the least bytes each 4 MiB L2 pulls over the
    // fabric.  Inside a rectangle tiles walk M fastest, ~64 workgroups are resident per XCD, so its rm x BM rows of A are
    // re-used by successive groups of N-tile columns: if they fit the L2 they are read once, otherwise once per group.
    // (PMC: the 128x160 GEMM class read 137 MB per launch against 52 MB of operands; kbench 16384x5120x640: 615 -> 679 TFLOP/s)
    double best = -1.;
    static const int force_xm = getenv("IIR_XM") ? atoi(getenv("IIR_XM")) : 0;   // tuning knob: 1,2,4,8 forces the split
    const double row_bytes = (double)g.K * 2.;
    for (int xm = 1; xm <= 8; xm *= 2) {
        if (force_xm && xm != force_xm) continue;
        const int xn = 8 / xm;
        const int rm = (g.tiles_m + xm - 1) / xm, rn = (g.tiles_n + xn - 1) / xn;
        const double a_bytes = (double)rm * BM * row_bytes, w_bytes = (double)rn * BN * row_bytes;
        const int cols_per_group = rm >= 64 ? 1 : 64 / rm;
        const double groups = (double)((rn + cols_per_group - 1) / cols_per_group);
        double cost = (a_bytes <= 3.0 * 1048576. ? a_bytes : a_bytes * groups) + w_bytes;
        cost += ((double)rm * rn * 8 - (double)g.tiles_m * g.tiles_n) * 8. * BK * (BM + BN);     // padding workgroups of ragged rectangles
        if (best < 0. || cost < best) { best = cost; g.xm = xm; g.rm = rm; g.rn = rn; }
    }
    const dim3 grid(8 * g.rm * g.rn * (g.splitk == 2 ? 2 : 1)), block(128 * WAVES_M * (LW ? 2 : 1));
    // dynamic LDS above 64 KB needs the function attribute ON THE DEVICE THE LAUNCH GOES TO: tracked per device and kernel
    // instantiation (a function-local static used to run it once, for whichever device was current at the first launch)
    auto ensure_lds = [&](const void* fn) -> bool {
        static unsigned long long done[64] = {};           // one bit per device, per instantiation of this launcher x {gemm, conv}
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
        const int slot = conv ? 1 : 0;
        if (done[dev] & (1ull << slot)) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        done[dev] |= 1ull << slot;
        return true;
    };
    if constexpr (F8) {
        if (conv || g.splitk == 2) return IIR_EINVAL;
        if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, false, WAVES_M, false, LW, true>)) return IIR_ELAUNCH;
        iir_launch(gemm_kernel<E, BM, BN, ST, false, WAVES_M, false, LW, true>, grid, block, lds, stream, g);
        return iir_launch_status();
    } else if constexpr (W8) {
        if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, false, WAVES_M, true>)) return IIR_ELAUNCH;
        iir_launch(gemm_kernel<E, BM, BN, ST, false, WAVES_M, true>, grid, block, lds, stream, g);
        return iir_launch_status();
    } else if constexpr (LW) {
        if (g.splitk == 2) return IIR_EINVAL;
        if (conv) {
            if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, true, WAVES_M, false, true>)) return IIR_ELAUNCH;
            iir_launch(gemm_kernel<E, BM, BN, ST, true, WAVES_M, false, true>, grid, block, lds, stream, g);
        } else {
            if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, false, WAVES_M, false, true>)) return IIR_ELAUNCH;
            iir_launch(gemm_kernel<E, BM, BN, ST, false, WAVES_M, false, true>, grid, block, lds, stream, g);
        }
        return iir_launch_status();
    } else
    if (conv) {
        if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, true, WAVES_M>)) return IIR_ELAUNCH;
        iir_launch(gemm_kernel<E, BM, BN, ST, true, WAVES_M>, grid, block, lds, stream, g);
    } else {
        if (!ensure_lds((const void*)gemm_kernel<E, BM, BN, ST, false, WAVES_M>)) return IIR_ELAUNCH;
        iir_launch(gemm_kernel<E, BM, BN, ST, false, WAVES_M>, grid, block, lds, stream, g);
    }
    return iir_launch_status();
}

// fp16 build of every tile / ring depth; bf16 build (the VAE) of the two-stage tiles the chooser picks for it
template <int BM, int BN, int ST, int WAVES_M = 2>
int launch(const Geo& g, bool conv, hipStream_t stream) {
    if (g.f8) {               // all-fp8 build: the 4-wave tiles the chooser picks for the transformer linears
        if constexpr (WAVES_M == 2 && (ST == 2 || (ST == 3 && BM == 64 && BN == 160))) {
            if (conv || g.dtype != IIR_DT_F16) return IIR_EINVAL;
            return launch_t<f16, BM, BN, ST, WAVES_M, false, false, true>(g, false, stream);
        } else return IIR_EINVAL;
    }
    if (g.wscale) {           // fp8-weight build: fp16 activations, linear layers, the 4-wave tiles
        if constexpr (WAVES_M == 2 && (ST == 2 || (ST == 3 && BM == 64 && BN == 160))) {
            if (conv || g.dtype != IIR_DT_F16 || g.splitk == 2) return IIR_EINVAL;
            return launch_t<f16, BM, BN, ST, WAVES_M, true>(g, false, stream);
        } else return IIR_EINVAL;
    }
    if (g.dtype == IIR_DT_BF16) {
        if constexpr (ST == 2 && WAVES_M == 2) return launch_t<bf16, BM, BN, ST, WAVES_M>(g, conv, stream);
        else return IIR_EINVAL;
    }
    return launch_t<f16, BM, BN, ST, WAVES_M>(g, conv, stream);
}

template <int BM, int BN, int ST>
int lw_launch(const Geo& g, bool conv, hipStream_t stream) {
    if (g.f8) {
        if constexpr (ST == 3 && BM == 64 && BN == 160) { if (conv) return IIR_EINVAL; return launch_t<f16, BM, BN, ST, 2, false, true, true>(g, false, stream); }
        else return IIR_EINVAL;
    }
    if (g.wscale || g.dtype != IIR_DT_F16 || g.splitk == 2) return IIR_EINVAL;
    if constexpr (ST == 3) return launch_t<f16, BM, BN, ST, 2, false, true>(g, conv, stream);
    else { if (conv) return IIR_EINVAL; return launch_t<f16, BM, BN, ST, 2, false, true>(g, false, stream); }
}

struct TileShape { int bm, bn; };
constexpr TileShape kTiles[] = {{0, 0}, {128, 128}, {128, 64}, {64, 64}, {128, 160}, {64, 160}, {256, 128}, {128, 320}, {256, 256}, {32, 160}};

// Tile choice: the per-CU operand fill rate (L2 -> LDS, ~50-70 GB/s) bounds these launches, so pick the
// shape that minimises the bytes the busiest CU has to pull.  Two workgroups per CU overlap each other's
// load latency, so capacity is counted in 512 slots: cost = ceil(blocks / 512) * 2 * (BM + BN) [* K * 2 B].
// One workgroup per CU leaves nothing else to hide the L2 -> LDS latency: launches that fit one 64x160 workgroup per CU
// (M x N = 2048 x 1280: the level-2 projections) take that tile with a 3-deep ring instead of 320 128x64 workgroups.
// Cold-weight sweep (tools/tilebench.py, us): 2048x1280x1280  128x64/2 21.9 | 64x160/3 16.7 | /4 16.3;  2048x1280x5120
// 128x64/2 67.0 | 64x160/3 47.3 | /4 46.1.  In the loop: K >= 2560 only 72.8 ms/step, K >= 1280 71.8 (3 stages), 72.7
// (4 stages) -- same call.  (With the first-generation epilogue the K = 1280 extension had measured as a loss.)
// IIR_T5_MINK / IIR_T5_STAGES override for experiments.
int one_per_cu_min_k() { static const int v = [] { const char* e = getenv("IIR_T5_MINK"); return e ? atoi(e) : 1280; }(); return v; }
int one_per_cu_stages() { static const int v = [] { const char* e = getenv("IIR_T5_STAGES"); return e ? atoi(e) : 3; }(); return v; }

int pick_tile(int M, int N, bool paired, int K = 0) {
    // problems that fit one 64x160 workgroup per CU: that tile with a deep ring (see dispatch())
    if (K >= one_per_cu_min_k() && (long)((M + 63) / 64) * ((N + 159) / 160) <= 256) return 5;
    long best = -1;
    int pick = 1;
    for (int t = 1; t <= 5; ++t) {
        const long blocks = (long)((M + kTiles[t].bm - 1) / kTiles[t].bm) * ((N + kTiles[t].bn - 1) / kTiles[t].bn);
        const long cost = ((blocks + 511) / 512) * 2 * (kTiles[t].bm + kTiles[t].bn);
        if (best < 0 || cost < best || (cost == best && kTiles[t].bm * kTiles[t].bn > kTiles[pick].bm * kTiles[pick].bn)) {
            best = cost;
            pick = t;
        }
    }
    return pick;
}

constexpr long SK_CNT_BYTES = 4096;     // 1024 per-tile arrival counters ahead of the slabs
long splitk_ws_bytes(int M, int N) {
    const long t = (long)((M + 127) / 128) * ((N + 159) / 160);
    return t * 2 <= 256 ? SK_CNT_BYTES + t * 2 * 128 * 160 * (long)sizeof(float) : 0;
}

bool uses_splitk(int M, int N, int K, long ws_bytes) {
    const long need = splitk_ws_bytes(M, N);
    return K >= one_per_cu_min_k() && (K / BK) % 2 == 0 && need > 0 && ws_bytes >= need;
}

bool gemm8_auto(const Geo& g, bool conv) {
    static const bool g8_on = !(getenv("IIR_G8") && atoi(getenv("IIR_G8")) == 0);
    return g8_on && !conv && (g.f8 ? 2 * g.K : g.K) >= 640 && (long)(g.M / 256) * (g.N / 320) >= 256 && iir::gemm8_covers(g, 320);
}

int dispatch(const Geo& g, bool conv, int tile, hipStream_t stream) {
    if (g.xa_on) return conv ? IIR_EINVAL : launch<64, 128, 3>(g, false, stream);      // to_q + cross-attention: the head-aligned 64 x 128 tile, two workgroups per CU
    // Two-slice split-K, taken only when the caller hands over a workspace: two 128x160 workgroups per tile each take half of
    // K and the last one to finish reduces (see the kernel).  Meant for long-K problems too small to fill the chip with
    // 128x160 tiles (M x N = 2048 x 1280 at K = 5120 / 11520); MEASURED SLOWER than one 64x160 workgroup per CU over all
    // of K on exactly those (51.8 vs 44.8 us warm at K = 5120, equal at K = 11520, 73.9 vs 73.0 ms per step): the agent-scope
    // release per workgroup and the fp32 slab round trip cost more than the smaller operand fill saves.  The engine does
    // not pass a workspace unless IIR_SPLITK=1.
    if (tile == 0 && g.sk_slabs && !g.ln_out && !g.ln_in && !g.gn_out && g.dtype == IIR_DT_F16 && uses_splitk(g.M, g.N, g.K, g.sk_bytes)) {
        Geo g2 = g;
        g2.splitk = 2;
        return launch<128, 160, 3>(g2, conv, stream);
    }
    // tile: 0 = auto; t in {1: 128x128, 2: 128x64, 3: 64x64, 4: 128x160, 5: 64x160}; t + 10*stages selects the ring depth.
    // large-N linears whose 256 x 320 tiles fill the chip (the GEGLU projections): the 8-wave two-tile-deep kernel of
    // gemm8.hip (142 FLOP per staged byte against 71 for two 128x160 workgroups per CU).  IIR_G8=0 switches it off (A/B).
    if (tile == 0 && gemm8_auto(g, conv)) return iir::gemm8_launch(g, 320, stream);
    if (tile == 91 || tile == 92) return conv ? IIR_EINVAL : iir::gemm8_launch(g, tile == 91 ? 320 : 256, stream);
    if (tile == 0) tile = pick_tile(g.M, g.N, g.epi != IIR_EPI_PLAIN, g.f8 ? 2 * g.K : g.K);
    if (tile < 10) {
        // ring depth: with at most one workgroup per CU nothing else hides the tile latency, and a long K loop
        // amortises the deeper prologue -> 3 stages for the 64x160 tile (measured +17..40 % on K >= 2560, M*N = 2048x1280)
        const long blocks = (long)((g.M + kTiles[tile].bm - 1) / kTiles[tile].bm) * ((g.N + kTiles[tile].bn - 1) / kTiles[tile].bn);
        const bool one_per_cu = tile == 5 && blocks <= 256 && (g.f8 ? 2 * g.K : g.K) >= one_per_cu_min_k() && g.dtype == IIR_DT_F16;     // (all-fp8: g.K counts 2-byte units)
        const int stages = one_per_cu ? one_per_cu_stages() : IIR_DEFAULT_STAGES;
        // one workgroup per CU and a plain GEMM: the loader-wave build (kbench, warm: 2048x1280x1280 456 -> 535 TFLOP/s,
        // K = 5120 645 -> 707; the per-tile slope stays at the ~70 GB/s per-CU L2 -> LDS fill rate, the fixed part drops)
        static const bool lw_on = !(getenv("IIR_T5_LW") && atoi(getenv("IIR_T5_LW")) == 0);
        static const bool lw_conv = !(getenv("IIR_T5_LWCONV") && atoi(getenv("IIR_T5_LWCONV")) == 0);
        static const bool lw2 = getenv("IIR_T5_LW2") && atoi(getenv("IIR_T5_LW2")) == 1;     // experiment: 2-stage loader-wave build (58 KB of LDS: two kernels can share a CU)
        if (one_per_cu && lw2 && !conv && !g.wscale && g.splitk != 2) tile = 75;
        else
        if (one_per_cu && lw_on && (!conv || (lw_conv && stages == 3)) && (!g.wscale || (g.f8 && stages == 3)) && g.splitk != 2 && stages >= 3 && stages <= 5) tile = stages == 3 ? 55 : stages == 4 ? 65 : 85;
        else
        tile += 10 * stages;
    }
    switch (tile) {
        case 21: return launch<128, 128, 2>(g, conv, stream);
        case 31: return launch<128, 128, 3>(g, conv, stream);
        case 22: return launch<128, 64, 2>(g, conv, stream);
        case 32: return launch<128, 64, 3>(g, conv, stream);
        case 23: return launch<64, 64, 2>(g, conv, stream);
        case 33: return launch<64, 64, 3>(g, conv, stream);
        case 24: return launch<128, 160, 2>(g, conv, stream);
        case 25: return launch<64, 160, 2>(g, conv, stream);
        case 35: return launch<64, 160, 3>(g, conv, stream);
        case 45: return launch<64, 160, 4>(g, conv, stream);
        case 55: return lw_launch<64, 160, 3>(g, conv, stream);     // 4 compute + 4 loader waves (gemm only, fp16)
        case 65: return lw_launch<64, 160, 4>(g, conv, stream);
        case 75: return lw_launch<64, 160, 2>(g, conv, stream);
        case 85: return lw_launch<64, 160, 5>(g, conv, stream);
        case 34: return launch<128, 160, 3>(g, conv, stream);
        case 56: return launch<64, 160, 3, 4>(g, conv, stream);    // 8 waves (4 x 2): two waves per SIMD on the one-per-CU tile
        case 66: return launch<64, 160, 4, 4>(g, conv, stream);
        case 57: return launch<128, 160, 3, 4>(g, conv, stream);
        case 47: return launch<128, 160, 2, 4>(g, conv, stream);
        case 42: return launch<128, 64, 4>(g, conv, stream);
        case 26: return launch<256, 128, 2, 4>(g, conv, stream);   // 8 waves, 1 workgroup per CU
        case 36: return launch<256, 128, 3, 4>(g, conv, stream);
        case 29: return launch<32, 160, 2>(g, conv, stream);       // 512 workgroups on a 2048 x 1280 problem: two per CU, 24 KB per K tile each
        case 27: return launch<128, 320, 2, 4>(g, conv, stream);   // 8 waves (4 x 2), 56 KB of operands per K tile: 91 FLOP per staged byte
        case 90:                                                   // 256x320, 8 waves, 72 KB per K tile: 142 FLOP per staged byte; paired
            if (g.epi == IIR_EPI_PLAIN) return IIR_EINVAL;         //   epilogues only (the half-width output tile is what fits the ring)
            return launch<256, 320, 2, 4>(g, conv, stream);
        case 28:                                                   // 8 waves, 64 KB per K tile: 128 FLOP per staged byte; the output tile
            if (g.epi == IIR_EPI_PLAIN && !g.c_f32) return IIR_EINVAL;   //   only fits the ring in its paired (half-width) form
            return launch<256, 256, 2, 4>(g, conv, stream);
        default: return IIR_EINVAL;
    }
}

// no prefetch requested: the (unconditional) touches re-read the first line of this launch's own weights
void finish_geo(Geo& g) {
    static const int wt = getenv("IIR_ST_WT") ? atoi(getenv("IIR_ST_WT")) : 1;      // A/B switch; default on (step 60.39 -> 60.22 ms, same box)
    g.st_wt = wt && ((long)g.M * g.ldc * 2 < (1L << 31));      // 32-bit buffer offsets
    if (g.pf_lines <= 0) { g.pf = (const char*)g.W; g.pf_lines = 1; }
    g.c_vec = (g.ldc % 8 == 0) && ((uintptr_t)g.C % 16 == 0);
    g.r_vec = g.res && (g.ldr % 8 == 0) && ((uintptr_t)g.res % 16 == 0);
}

}  // namespace

extern "C" int iir_gemm_tile_bn(int32_t tile) { tile %= 10; return (tile >= 1 && tile <= 9) ? kTiles[tile].bn : -1; }

extern "C" int64_t iir_gemm_splitk_workspace_bytes(int32_t M, int32_t N) { return splitk_ws_bytes(M, N); }
extern "C" int iir_gemm_uses_splitk(int32_t M, int32_t N, int32_t K, int64_t ws_bytes) { return uses_splitk(M, N, K, ws_bytes) ? 1 : 0; }

extern "C" int iir_gemm_pick_tile(int32_t M, int32_t N, int32_t K, int32_t paired) { return pick_tile(M, N, paired != 0, K); }

// Partials per row a tile = 0, plain-epilogue launch of (M, N, K) leaves in `ln_stats_out` (= N / BN of the tile it resolves to), or 0
// when that launch cannot produce them (ragged tiles).  The buffer is [parts][M] float2.
extern "C" int iir_gemm_ln_parts(int32_t M, int32_t N, int32_t K) {
    const int t = pick_tile(M, N, false, K);
    const int bm = kTiles[t].bm, bn = kTiles[t].bn;
    if (t < 1 || t > 5 || M % bm || N % bn || N / bn > 8) return 0;      // (the consumer holds at most 8 partials per row in registers)
    return N / bn;
}

static int fill_gemm_geo(const iir_gemm_desc* d, Geo& g);

// 1 when the tile = 0 launch of an (M, N, K) GEMM / implicit-GEMM conv (plain epilogue, fp16) can leave GroupNorm partials
// (`gn_stats_out`): whole tiles and room for the column partials behind the staged output tile.
extern "C" int iir_gemm_gn_supported(int32_t M, int32_t N, int32_t K, int32_t is_conv) {
    if (M <= 0 || N <= 0 || K <= 0 || M % 64) return 0;
    const int t = pick_tile(M, N, false, K);
    if (t < 1 || t > 5) return 0;
    const int bm = kTiles[t].bm, bn = kTiles[t].bn;
    if (M % bm || N % bn) return 0;
    const long blocks = (long)(M / bm) * (N / bn);
    const bool lw = t == 5 && blocks <= 256 && K >= one_per_cu_min_k() && one_per_cu_stages() == 3;
    const int st = lw ? 3 : IIR_DEFAULT_STAGES, nt = lw ? 512 : 256;
    const long ring = (long)st * (bm + bn) * 128, need = ((long)bm * (2 * bn + 32) + 15) / 16 * 16 + (long)(nt / (bn / 8)) * bn * 8;
    (void)is_conv;
    return bn <= nt && need <= ring ? 1 : 0;
}

// 1 when the tile = 0 all-fp8 launch of (M, N, K) can store its result as fp8 bytes (`c_fp8`): whole tiles of the tile it resolves to
extern "C" int iir_gemm_fp8_out_supported(int32_t M, int32_t N, int32_t K, int32_t paired) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 128) return 0;
    const int t = pick_tile(M, N, paired != 0, K);
    if (t < 1 || t > 5) return 0;
    return (M % kTiles[t].bm == 0 && N % kTiles[t].bn == 0) ? 1 : 0;
}

extern "C" int iir_gemm_f16(const iir_gemm_desc* d, void* stream) {
    (void)hipGetLastError();
    Geo g{};
    const int rc = fill_gemm_geo(d, g);
    if (rc != IIR_OK) return rc;
    return dispatch(g, false, d->tile, (hipStream_t)stream);
}

// Which kernel / tile `iir_gemm_f16(d)` resolves to, without launching: 91 = the 8-wave 256x320 kernel (gemm8.hip), otherwise the
// 4-wave tile id of `iir_gemm_pick_tile` (or d->tile when the caller forces one).  Used to NAME launches (bench.py roofline classes).
extern "C" int iir_gemm_resolve_tile(const iir_gemm_desc* d) {
    Geo g{};
    if (fill_gemm_geo(d, g) != IIR_OK) return -1;
    if (g.xa_on) return 93;
    if (d->tile != 0) return d->tile;
    if (gemm8_auto(g, false)) return 91;
    return pick_tile(g.M, g.N, g.epi != IIR_EPI_PLAIN, g.f8 ? 2 * g.K : g.K);
}

static int fill_gemm_geo(const iir_gemm_desc* d, Geo& g) {
    if (!d || !d->A || !d->W || !d->C) return IIR_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->K % BK) return IIR_EINVAL;
    if (d->N % 4 || d->lda % 8 || d->ldc % 4) return IIR_EINVAL;
    if (d->epi != IIR_EPI_PLAIN && (d->N % 16)) return IIR_EINVAL;
    if (d->epi == IIR_EPI_SFT && !d->res) return IIR_EINVAL;
    if (d->rowbias && d->rows_per_rb <= 0) return IIR_EINVAL;
    if (d->epi < IIR_EPI_PLAIN || d->epi > IIR_EPI_XATTN) return IIR_EINVAL;
    if (d->epi == IIR_EPI_XATTN) {
        const iir_attn_kv* kv = d->xattn_kv;
        if (!kv || d->dtype != IIR_DT_F16 || d->N % 128 || d->M % 64 || d->xattn_tq <= 0 || d->xattn_tq % 64 || d->M % d->xattn_tq) return IIR_EINVAL;
        if (d->res || d->rowbias || d->act || d->Ct || d->c_f32 || d->wscale || d->ln_stats_out || d->gn_stats_out || d->splitk_ws) return IIR_EINVAL;
        if (d->ldc % 8 || (uintptr_t)d->C % 16 || (d->tile != 0 && d->tile != 93)) return IIR_EINVAL;
        if (kv[0].Tkv < 1 || kv[0].Tkv > 80 || kv[1].Tkv < 1 || kv[1].Tkv > 64) return IIR_EINVAL;
        for (int sgm = 0; sgm < 2; ++sgm) {
            if (!kv[sgm].K || !kv[sgm].Vt || (uintptr_t)kv[sgm].K % 16 || (uintptr_t)kv[sgm].Vt % 16) return IIR_EINVAL;
            if (kv[sgm].ldk % 8 || kv[sgm].k_batch_stride % 8 || kv[sgm].ldvt % 8 || kv[sgm].vt_batch_stride % 8) return IIR_EINVAL;
            g.xa_k[sgm] = (const f16*)kv[sgm].K; g.xa_ldk[sgm] = kv[sgm].ldk; g.xa_kb[sgm] = kv[sgm].k_batch_stride;
            g.xa_vt[sgm] = (const f16*)kv[sgm].Vt; g.xa_ldvt[sgm] = kv[sgm].ldvt; g.xa_vb[sgm] = kv[sgm].vt_batch_stride;
            g.xa_tk[sgm] = kv[sgm].Tkv;
        }
        g.xa_on = 1; g.xa_tq = d->xattn_tq;
    }
    g.A = (const f16*)d->A; g.lda = d->lda; g.W = (const f16*)d->W; g.C = (f16*)d->C; g.ldc = d->ldc;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.bias = (const f16*)d->bias; g.rowbias = (const f16*)d->rowbias; g.ldrb = d->ldrb; g.rows_per_rb = d->rows_per_rb;
    g.res = (const f16*)d->res; g.ldr = d->ldr; g.epi = d->epi == IIR_EPI_XATTN ? IIR_EPI_PLAIN : d->epi; g.act = d->act;      // (the tile leaves the XATTN epilogue as a plain one)
    g.out_scale = d->out_scale == 0.f ? 1.f : d->out_scale;
    g.pf = (const char*)d->prefetch; g.pf_lines = d->prefetch ? (int)(d->prefetch_bytes / 128) : 0;
    finish_geo(g);
    if (d->dtype != IIR_DT_F16 && d->dtype != IIR_DT_BF16) return IIR_EINVAL;
    g.dtype = d->dtype;
    if (d->c_f32) {
        if (d->epi != IIR_EPI_PLAIN || d->bias || d->rowbias || d->res || d->Ct || d->act || d->ldc % 4 || (uintptr_t)d->C % 16 || d->splitk_ws) return IIR_EINVAL;
        g.c_f32 = 1;
    }
    if (d->wscale) {          // W is fp8-E4M3 [N][K] bytes with one fp32 scale per row of W
        if (d->c_f32 || d->K % 64 || (uintptr_t)d->W % 16 || (uintptr_t)d->wscale % 16 || d->dtype != IIR_DT_F16) return IIR_EINVAL;
        g.wscale = (const float*)d->wscale;
        g.sk_slabs = nullptr;
    }
    if (d->c_fp8) {
        if (d->epi == IIR_EPI_XATTN || d->epi == IIR_EPI_SFT || d->c_f32 || d->Ct || d->ln_stats_out || d->gn_stats_out || d->dtype != IIR_DT_F16) return IIR_EINVAL;
        g.c_fp8 = 1;
    }
    g.a_scale = 1.f;
    if (d->a_fp8) {           // A is fp8-E4M3 [M][K] bytes as well (lda in bytes): both operands by 128-byte rows of 128 K values
        if (!d->wscale || d->K % 128 || d->lda % 16 || (uintptr_t)d->A % 16 || d->ln_stats_in || d->epi == IIR_EPI_XATTN || d->splitk_ws) return IIR_EINVAL;
        g.f8 = 1;
        g.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale;
        g.K = d->K / 2; g.lda = d->lda / 2;          // counted in 2-byte units from here on: the fp16 staging path unchanged
    }
    if (d->Ct) {
        if (d->epi != IIR_EPI_PLAIN || d->tr_from < 0 || d->tr_from >= d->N || d->tr_from % 8 || d->ldct < d->M) return IIR_EINVAL;
        g.Ct = (f16*)d->Ct; g.ldct = d->ldct; g.tr_from = d->tr_from;
        g.ct_vec = (d->ldct % 8 == 0) && ((uintptr_t)d->Ct % 16 == 0);
    }
    if (d->splitk_ws && d->splitk_ws_bytes > SK_CNT_BYTES) {
        g.sk_cnt = (int*)d->splitk_ws; g.sk_slabs = (float*)((char*)d->splitk_ws + SK_CNT_BYTES); g.sk_bytes = d->splitk_ws_bytes;
    }
    if (d->ln_stats_out) {
        if ((uintptr_t)d->ln_stats_out % 8 || d->c_f32 || d->Ct || d->epi != IIR_EPI_PLAIN) return IIR_EINVAL;
        g.ln_out = (float*)d->ln_stats_out;
    }
    if (d->gn_stats_out) {
        if ((uintptr_t)d->gn_stats_out % 8 || d->M % 64 || d->epi != IIR_EPI_PLAIN || d->c_f32 || d->Ct || d->ln_stats_out || d->wscale) return IIR_EINVAL;
        g.gn_out = (float*)d->gn_stats_out;
    }
    if (d->ln_stats_in) {
        if (!d->ln_colsum || d->ln_parts <= 0 || d->ln_parts > 8 || d->ln_part_cols <= 0 || (long)d->ln_parts * d->ln_part_cols != d->K || d->c_f32 ||
            (uintptr_t)d->ln_stats_in % 8 || (uintptr_t)d->ln_colsum % 16 || !(d->ln_eps > 0.f)) return IIR_EINVAL;
        g.ln_in = (const float*)d->ln_stats_in; g.ln_colsum = (const float*)d->ln_colsum;
        g.ln_parts = d->ln_parts; g.ln_part_cols = d->ln_part_cols; g.ln_eps = d->ln_eps;
    }
    return IIR_OK;
}

extern "C" int iir_conv2d_nhwc_f16(const iir_conv_desc* c, void* stream) {
    (void)hipGetLastError();
    if (!c || !c->X || !c->Wt || !c->Y || !c->zero_page) return IIR_EINVAL;
    if (c->ksize != 3 && c->ksize != 1) return IIR_EINVAL;
    if (c->Cin % BK || c->ldx % 8 || c->Cout % 4 || c->ldy % 4) return IIR_EINVAL;
    if (c->stride != 1 && c->stride != 2) return IIR_EINVAL;
    if (c->upsample && c->stride != 1) return IIR_EINVAL;
    if (c->epi != IIR_EPI_PLAIN && (c->Cout % 16)) return IIR_EINVAL;
    if (c->epi == IIR_EPI_SFT && !c->res) return IIR_EINVAL;
    if (c->rowbias && c->rows_per_rb <= 0) return IIR_EINVAL;
    const int pad = c->pad_mode == 1 ? 0 : c->ksize / 2;   // mode 1: taps reach 1 pixel past the bottom/right edge only
    const int pad_hi = c->pad_mode == 1 ? 1 : pad;
    const int Hin = c->upsample ? 2 * c->H : c->H, Win = c->upsample ? 2 * c->Wd : c->Wd;
    Geo g{};
    g.Ho = (Hin + pad + pad_hi - c->ksize) / c->stride + 1;
    g.Wo = (Win + pad + pad_hi - c->ksize) / c->stride + 1;
    g.A = (const f16*)c->X; g.lda = c->ldx; g.W = (const f16*)c->Wt; g.C = (f16*)c->Y; g.ldc = c->ldy;
    g.M = c->R * g.Ho * g.Wo; g.N = c->Cout; g.K = c->ksize * c->ksize * c->Cin;
    g.bias = (const f16*)c->bias; g.rowbias = (const f16*)c->rowbias; g.ldrb = c->ldrb; g.rows_per_rb = c->rows_per_rb;
    g.res = (const f16*)c->res; g.ldr = c->ldr; g.epi = c->epi; g.act = c->act;
    g.out_scale = c->out_scale == 0.f ? 1.f : c->out_scale;
    g.H = c->H; g.Wd = c->Wd; g.Cin = c->Cin; g.ks = c->ksize; g.stride = c->stride; g.pad = pad; g.ups = c->upsample;
    g.zero = (const f16*)c->zero_page;
    g.pf = (const char*)c->prefetch; g.pf_lines = c->prefetch ? (int)(c->prefetch_bytes / 128) : 0;
    finish_geo(g);
    if (c->dtype != IIR_DT_F16 && c->dtype != IIR_DT_BF16) return IIR_EINVAL;
    g.dtype = c->dtype;
    g.x_img_stride = c->x_img_stride ? c->x_img_stride : (int64_t)c->H * c->Wd * c->ldx;
    g.y_img_rows = c->y_img_rows; g.res_img_rows = c->res_img_rows;
    if (c->gn_stats_out) {
        if ((uintptr_t)c->gn_stats_out % 8 || (g.Ho * g.Wo) % 64 || c->epi != IIR_EPI_PLAIN || c->y_img_rows || c->res_img_rows) return IIR_EINVAL;
        g.gn_out = (float*)c->gn_stats_out;
    }
    if (c->splitk_ws && c->splitk_ws_bytes > SK_CNT_BYTES) {
        g.sk_cnt = (int*)c->splitk_ws; g.sk_slabs = (float*)((char*)c->splitk_ws + SK_CNT_BYTES); g.sk_bytes = c->splitk_ws_bytes;
    }
    return dispatch(g, true, c->tile, (hipStream_t)stream);
}
