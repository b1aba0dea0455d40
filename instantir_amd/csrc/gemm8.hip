// K1b: the large-N linear layers (GEGLU feed-forward projection, module/min_sdxl.py:502-528; q|k|v at level 1) on a
// 256 x BN output tile per workgroup, 8 waves, one workgroup per CU -- round 3.
//
// Why a second GEMM kernel: L2 -> LDS fill tops out at ~60 B/clk per CU (tools/probes/fill_probe.hip: 125 GB/s per CU when the
// XCD's working set is L2 resident), so what a launch can reach is set by FLOPs per staged byte = 2*BM*BN / (2*(BM+BN)):
// 71 for the 128x160 tile (two workgroups per CU), 142 for 256x320.  The 4-wave kernel's 256-row instantiations were
// fill-LATENCY bound instead (two 72 KB stages, the whole next tile requested at one barrier and waited for at the next).
// Here the two operands of a K tile are requested at different times, two tiles deep:
//   * the wave grid is 2 (M) x 4 (N): a wave owns 128 x BN/4 outputs.  Its weight fragments of a K tile (NI x 2 reads) are
//     taken into registers at the start of the tile, so the WEIGHT half of the stage is free again after the first of the
//     tile's eight sub-phases: the weights of tile t+2 are requested into it during tile t (one 1-KiB LDS-DMA piece per
//     sub-phase and wave), ~2 tiles ahead of their use;
//   * the ACTIVATION half is read one 16-row fragment pair per sub-phase (double buffered in 16 registers, fetched a
//     sub-phase ahead of its 2*NI MFMAs) and is free at the end of the tile: the activations of tile t+1 are requested
//     right after tile t begins, one tile ahead.
//   Two barriers per K tile (X: tile landed / previous tile's buffer free; Y: weight half free), one counted vmcnt.
// Epilogue: same two-phase form as gemm_conv.hip (registers -> LDS as finished fp16 rows -> 16-byte row-contiguous
// stores), bias / LayerNorm-fold (ln_stats_in) / activation / GEGLU / residual / out_scale; the plain epilogue stages the
// tile in two 128-row halves (256 x 640 B does not fit the ring).
#include "common.h"
#include "gemm_geo.h"
#include "../../include/instantir_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

using iir::Geo;
constexpr int BK = 64;
constexpr int PF_TOUCHES = 2;

// V (schedule variant, A/B switch IIR_G8V): 0 = activation pieces requested in one block at the start of the tile; 1 = one
// piece per sub-phase (activations in sub-phases 0..3, weights behind them); 2 = as 1, and waves 4-7 (the second wave of every
// SIMD) issue a sub-phase's piece AFTER its MFMAs instead of before them, so the two waves of a SIMD do not sit in the
// (~70-125 cycle) LDS-DMA issue at the same moment with the matrix pipe idle behind them
// F8: both operands fp8-E4M3 (gemm_conv.hip, same scheme: K counted in 2-byte units by the host, two fp8 MFMAs per 16-byte fragment pair,
// wscale[n] * a_scale in the epilogue)
template <int BN, int V, bool F8 = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const Geo g) {
    using E = f16;
    using E4 = f16x4;
    using E8 = f16x8;
    constexpr int BM = 256, NT = 512;
    constexpr int WM = 128, WN = BN / 4, MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, RING_BYTES = 2 * STAGE;
    constexpr int A_PIECES = BM / 64, B_PIECES = BN / 64;            // 1-KiB LDS-DMA pieces per wave and K tile
    static_assert(BN % 64 == 0 && NI * 16 == WN && B_PIECES <= MI - 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* rowstat = (float2*)(smem + RING_BYTES + 2048);             // [BM] (rstd, -rstd * mean) (ln_in)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    int tm, tn;
    {   // XCD-aware tile order (see gemm_conv.hip): workgroups b, b+8, ... share an L2
        const int bid = (int)blockIdx.x, xcd = bid & 7, local = bid >> 3;
        const int rx = xcd % g.xm, ry = xcd / g.xm;
        tm = rx * g.rm + local % g.rm;
        tn = ry * g.rn + local / g.rm;
        if (tm >= g.tiles_m || tn >= g.tiles_n) return;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: piece p = q * 8 + wave covers rows 8p .. 8p+7; lane = (row in piece, swizzled 16-byte chunk)
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const unsigned a_voff = (unsigned)(srow * (int)g.lda * 2 + schunk * 16);
    const unsigned b_voff = (unsigned)(srow * g.K * 2 + schunk * 16);
    const char* a_base = (const char*)g.A + (long)(m0 + wave * 8) * g.lda * 2;
    const char* b_base = (const char*)g.W + (long)(n0 + wave * 8) * g.K * 2;
    const long a_pstride = 64L * g.lda * 2, b_pstride = 64L * g.K * 2;
    auto stage_a = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < A_PIECES; ++q)
            glds16(a_base + q * a_pstride + t * 128 + a_voff, smem + buf * STAGE + (q * 8 + wave) * 1024);
    };
    auto stage_b1 = [&](int t, int buf, int q) {
        glds16(b_base + q * b_pstride + t * 128 + b_voff, smem + buf * STAGE + A_BYTES + (q * 8 + wave) * 1024);
    };

    // ---- fragment read offsets
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int phys = (s * 4 + fq) ^ (frow & 7);
        a_off[s] = (wm * WM + frow) * 128 + phys * 16;
        b_off[s] = A_BYTES + (wn * WN + frow) * 128 + phys * 16;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage_a(0, 0);
#pragma unroll
    for (int q = 0; q < B_PIECES; ++q) stage_b1(0, 0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < B_PIECES; ++q) stage_b1(1, 1, q);
    }

    // LayerNorm statistics of this tile's rows from the producer's partials (as gemm_conv.hip): one batch of clamped loads
    if (g.ln_in && tid < BM) {
        constexpr int MAXP = 8;
        float2 lnp_in[MAXP];
        const int m = min(m0 + tid, g.M - 1);
#pragma unroll
        for (int j = 0; j < MAXP; ++j) lnp_in[j] = ((const float2*)g.ln_in)[(long)min(j, g.ln_parts - 1) * g.M + m];
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
        rowstat[tid] = make_float2(rstd, -rstd * mean);
    }

    E8 bfr[2][NI], a0[2], a1[2];
    auto read_b = [&](const char* st) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NI; ++j) bfr[s][j] = *(const E8*)(st + b_off[s] + j * 2048);
    };
    auto read_a = [&](const char* st, int i, E8 (&a)[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) a[s] = *(const E8*)(st + a_off[s] + i * 2048);
    };

    // X(0): tile 0 landed for every wave (the weights of tile 1 may still be in flight)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(B_PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    auto stage_a1 = [&](int t, int buf, int q) {
        glds16(a_base + q * a_pstride + t * 128 + a_voff, smem + buf * STAGE + (q * 8 + wave) * 1024);
    };
    const bool late = V == 2 && wave >= 4;
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        const char* st = smem + cur * STAGE;
        read_b(st);
        read_a(st, 0, a0);
        const bool more_a = t + 1 < nk, more_b = t + 2 < nk;
        if (V == 0 && more_a) stage_a(t + 1, cur ^ 1);      // the other stage's activation half: last read in tile t-1
        // piece of sub-phase i (V >= 1): activation pieces first (all of them are issued before the first weight piece, which
        // is what the counted vmcnt at X relies on), the weight pieces only after Y (i >= 1)
        auto piece = [&](int i) {
            if (V == 0) { if (i >= 1 && i <= B_PIECES && more_b) stage_b1(t + 2, cur, i - 1); return; }
            if (i < A_PIECES) { if (more_a) stage_a1(t + 1, cur ^ 1, i); }
            if (i >= MI - B_PIECES && more_b) stage_b1(t + 2, cur, i - (MI - B_PIECES));
        };
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            E8(&ac)[2] = (i & 1) ? a1 : a0;
            E8(&an)[2] = (i & 1) ? a0 : a1;
            if (i + 1 < MI) read_a(st, i + 1, an);
            if (!late) piece(i);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if constexpr (F8) {
                        typedef long l2 __attribute__((ext_vector_type(2)));
                        const l2 b2 = __builtin_bit_cast(l2, bfr[s][j]), a2 = __builtin_bit_cast(l2, ac[s]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[s][j], ac[s], acc[i][j], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_s_setprio(0);
            if (i == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Y(t): every wave holds its weight fragments
            if (late) piece(i);
        }
        // X(t+1): tile t+1 landed (activations requested a tile ago, weights two tiles ago; the weights of tile t+2 may still
        // be in flight), and every wave is done with tile t's stage
        if (more_b) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(B_PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        cur ^= 1;
    }

    // ---- epilogue: four chunks of 64 tile rows (the accumulator rows i = 2c, 2c+1 of both wave rows), double buffered in
    // the idle ring: chunk c's finished fp16 rows go registers -> LDS (bias / LayerNorm fold / activation / GEGLU), one
    // barrier, then whole-row 16-byte stores -- which are asynchronous, so they drain under the vector work of chunk c+1
    // (with all 256 workgroups of a one-round launch reaching their write-out together, a monolithic epilogue left both the
    // erf arithmetic and the HBM write exposed).
    const bool paired = g.epi != IIR_EPI_PLAIN;
    const int cs = (paired ? BN : 2 * BN) + 32;          // staged row stride in bytes (odd multiple of 32 mod 256)
    constexpr int CHUNK_ROWS = 64, STAGE_STRIDE = CHUNK_ROWS * (2 * BN + 32);
    auto touch_next_weights = [&]() {
        const int per = (g.pf_lines + (int)gridDim.x - 1) / (int)gridDim.x;
        const long l0 = (long)blockIdx.x * per, last = g.pf_lines - 1;
        char* scratch = smem + RING_BYTES + wave * 256;
#pragma unroll
        for (int i = 0; i < PF_TOUCHES; ++i) {
            long l = l0 + min(tid + i * NT, per - 1);
            if (l > last) l = last;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(g.pf + l * 128), (LDS_AS void*)scratch, 4, 0, 0);
        }
    };
    // column constants of this lane's NI column quads
    f32x4 c1v[NI], scv[F8 ? NI : 1];
    E4 c0v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nc = n0 + wn * WN + j * 16 + fq * 4;
        c1v[j] = g.ln_in ? *(const f32x4*)(g.ln_colsum + nc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (F8) scv[j] = *(const f32x4*)(g.wscale + nc) * g.a_scale;
        if (g.bias) c0v[j] = *(const E4*)(g.bias + nc);
        else for (int t = 0; t < 4; ++t) c0v[j][t] = (E)0.f;
    }
    const int cpr = paired ? BN / 16 : BN / 8;                 // 16-byte chunks per staged row
    const int no_tile = paired ? n0 / 2 : n0;
    const bool use_res = g.res && !paired;
    const int total = CHUNK_ROWS * cpr;                        // 16-byte pieces of one chunk (1280 or 2560)
    touch_next_weights();
#pragma unroll
    for (int c = 0; c < MI / 2; ++c) {
        char* ct = smem + (c & 1) * STAGE_STRIDE;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * c + ii;
            const int lr = wm * WM + i * 16 + frow;                       // row inside the tile
            const float2 rs = g.ln_in ? rowstat[lr] : make_float2(1.f, 0.f);
            char* rowp = ct + (wm * 32 + ii * 16 + frow) * cs;            // row inside the chunk
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                float a[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a[t] = fmaf(F8 ? acc[i][j][t] * scv[F8 ? j : 0][t] : acc[i][j][t], rs.x, fmaf(rs.y, c1v[j][t], (float)c0v[j][t]));
                if (!paired) {
                    if (g.act == IIR_ACT_SILU) for (int t = 0; t < 4; ++t) a[t] = silu_f(a[t]);
                    else if (g.act == IIR_ACT_GELU) for (int t = 0; t < 4; ++t) a[t] = gelu_erf_f(a[t]);
                    else if (g.act == IIR_ACT_QUICKGELU) for (int t = 0; t < 4; ++t) a[t] = a[t] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * a[t]));
                    E4 o;
                    for (int t = 0; t < 4; ++t) o[t] = (E)a[t];
                    *(E4*)(rowp + (wn * WN + j * 16 + fq * 4) * 2) = o;
                } else {
                    // GEGLU: value lanes (fq = 0,1) and their gate lanes (fq + 2) sit 32 lanes apart; the value lane finishes
                    // columns 0,1 of the quad, its gate lane columns 2,3 (gemm_conv.hip)
                    float b[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[t] = __shfl_xor(a[t], 32, 64);
                    const bool gate = fq >= 2;
                    const int lco = (wn * WN + j * 16) / 2 + (fq & 1) * 4 + (gate ? 2 : 0);
                    const float v0 = gate ? b[2] : a[0], v1 = gate ? b[3] : a[1], g0 = gate ? a[2] : b[0], g1 = gate ? a[3] : b[1];
                    f16x2 o2 = {(E)(v0 * gelu_erf_f(g0)), (E)(v1 * gelu_erf_f(g1))};
                    *(f16x2*)(rowp + lco * 2) = o2;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // chunk c staged (and chunk c-1's LDS reads long done)
        if (g.Ct && n0 >= g.tr_from) {
            // transposed write-out (the V third of a fused q|k|v projection -> the V^T image the attention kernel reads): a lane
            // gathers 8 consecutive tile rows of one column from the staged chunk and stores them as 16 contiguous bytes of Ct
            // (8 staged rows 8g .. 8g+7 are consecutive tile rows: they lie inside one 32-row run of a wave row)
#pragma unroll 1
            for (int p = tid; p < BN * (CHUNK_ROWS / 8); p += NT) {
                const int col = p >> 3, gq = p & 7;
                const long m = m0 + (gq >> 2) * WM + c * 32 + (gq & 3) * 8;
                E8 o;
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] = (E)((float)*(const E*)(ct + (gq * 8 + t) * cs + col * 2) * g.out_scale);
                *(E8*)(g.Ct + (long)(n0 + col - g.tr_from) * g.ldct + m) = o;
            }
            continue;
        }
        // whole rows, 16 bytes per lane; staged row s = (wave row, row in its 32) -> tile row wmS * 128 + 32 c + (s & 31)
#pragma unroll 1
        for (int p = tid; p < total; p += NT) {
            const int s = p / cpr, cc = p - s * cpr;
            const long m = m0 + (s >> 5) * WM + c * 32 + (s & 31);
            const E8 v = *(const E8*)(ct + s * cs + cc * 16);
            E8 o;
            if (use_res) {
                const E8 rr = *(const E8*)(g.res + m * g.ldr + no_tile + cc * 8);
                for (int t = 0; t < 8; ++t) o[t] = (E)(((float)v[t] + (float)rr[t]) * g.out_scale);
            } else {
                for (int t = 0; t < 8; ++t) o[t] = (E)((float)v[t] * g.out_scale);
            }
            if (g.c_fp8) *(long*)((char*)g.C + m * g.ldc + no_tile + cc * 8) = iir_fp8x8(o);      // the next all-fp8 GEMM's A operand
            else iir::store16(g.C, (m * g.ldc + no_tile + cc * 8) * 2, o, g.st_wt != 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA touches must land before the LDS is released
}

template <int BN, int V, bool F8 = false>
int launch8v(const Geo& g0, hipStream_t stream) {
    constexpr int BM = 256;
    Geo g = g0;
    g.tiles_m = g.M / BM;
    g.tiles_n = g.N / BN;
    // XCD partition: the split whose per-XCD operand panels are smallest (same rule as gemm_conv.hip)
    double best = -1.;
    const double row_bytes = (double)g.K * 2.;
    for (int xm = 1; xm <= 8; xm *= 2) {
        const int xn = 8 / xm;
        const int rm = (g.tiles_m + xm - 1) / xm, rn = (g.tiles_n + xn - 1) / xn;
        double cost = (double)rm * BM * row_bytes + (double)rn * BN * row_bytes;
        cost += ((double)rm * rn * 8 - (double)g.tiles_m * g.tiles_n) * 8. * BK * (BM + BN);
        if (best < 0. || cost < best) { best = cost; g.xm = xm; g.rm = rm; g.rn = rn; }
    }
    const size_t lds = 2 * (BM * 128 + BN * 128) + 2048 + BM * 8;
    static int attr_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_dev != dev) {
        if (hipFuncSetAttribute((const void*)gemm8_kernel<BN, V, F8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return IIR_ELAUNCH;
        attr_dev = dev;
    }
    iir_launch(gemm8_kernel<BN, V, F8>, dim3(8 * g.rm * g.rn), dim3(512), lds, stream, g);
    return iir_launch_status();
}

template <int BN>
int launch8(const Geo& g, hipStream_t stream) {
    if (g.f8) return launch8v<BN, 0, true>(g, stream);
    static const int v = getenv("IIR_G8V") ? atoi(getenv("IIR_G8V")) : 0;
    return v == 2 ? launch8v<BN, 2>(g, stream) : v == 1 ? launch8v<BN, 1>(g, stream) : launch8v<BN, 0>(g, stream);
}

}  // namespace

namespace iir {

bool gemm8_covers(const Geo& g, int bn) {
    if (bn != 320 && bn != 256) return false;
    if (g.dtype != IIR_DT_F16 || g.c_f32 || g.ln_out || g.gn_out || g.splitk == 2 || (g.wscale && !g.f8) || g.rowbias || g.xa_on) return false;
    if (g.f8 && (g.ln_in || g.Ct || bn != 320)) return false;            // all-fp8 form: the GEGLU / plain projections of the fp8 build
    if (g.c_fp8 && (g.ldc % 8 || (uintptr_t)g.C % 8)) return false;
    if (g.Ct && (g.epi != IIR_EPI_PLAIN || g.res || g.tr_from % bn || !g.ct_vec)) return false;      // transposed column range: whole tiles only
    if (g.epi != IIR_EPI_PLAIN && g.epi != IIR_EPI_GEGLU) return false;
    if (g.M % 256 || g.N % bn || g.K % 64 || g.K < 128) return false;
    if ((!g.c_vec && !g.c_fp8) || (g.res && g.epi == IIR_EPI_PLAIN && !g.r_vec)) return false;
    if (g.lda % 8 || ((uintptr_t)g.A % 16) || ((uintptr_t)g.W % 16)) return false;
    if (g.bias && ((uintptr_t)g.bias % 8)) return false;
    if (g.ln_in && g.ln_parts > 8) return false;
    if ((long)g.lda * 2 * 7 + 128 >= (1L << 31) || (long)g.K * 2 * 7 + 128 >= (1L << 31)) return false;
    return true;
}

int gemm8_launch(const Geo& g, int bn, hipStream_t stream) {
    if (!gemm8_covers(g, bn)) return IIR_EINVAL;
    return bn == 320 ? launch8<320>(g, stream) : launch8<256>(g, stream);
}

}  // namespace iir
