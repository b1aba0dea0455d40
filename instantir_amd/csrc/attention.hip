// K3/K4: flash-style attention for head_dim 64, fp16 in / fp32 softmax + accumulate, gfx950.
//
// Replaces F.scaled_dot_product_attention at module/ip_adapter/attention_processor.py:394
// (AttnProcessor2_0, self-attention) and the PAIR of SDPA calls sharing one query at :1165 and :1185
// (TA_IPAttnProcessor2_0: text KV + IP-adapter KV, outputs summed with scale 1.0 at :1192).  Up to
// two KV segments are processed in one pass over Q: out = softmax(Q K1^T) V1 + softmax(Q K2^T) V2.
//
// Layout: Q, K are token-major with heads side by side in a row (row strides given, so they can be
// column slices of a fused QKV GEMM output).  V is consumed TRANSPOSED: Vt[(h*64+d)][token]; the V
// projection is issued as the GEMM  Vt = Wv . X^T  (operands swapped), which costs nothing extra.
// Contract: each Vt row is readable and finite on columns [0, roundup8(Tkv)) (pad with zeros).
//
// Per block: 4 waves x 32 query rows.  Per 64-key tile: S^T = K . Q^T with 32x32x16 MFMA (keys on
// the accumulator rows, the query on the lane), so row max / row sum are in-lane reductions plus one
// cross-half exchange, and the exponentiated accumulator registers are, after fp16 packing, directly
// the B operand of O^T += V^T . P^T (no LDS round trip for P).  K and V^T tiles are staged by
// global_load_lds_dwordx4 into double-buffered XOR-swizzled LDS images.
#include "common.h"
#include "../../include/instantir_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct Seg { const f16* K; long ldk, kbs; const f16* Vt; long ldvt, vbs; int Tkv; };
struct AGeo {
    const f16* Q; long ldq, qbs;
    f16* O; long ldo, obs;
    int Tq, nseg;
    int qtiles, npairs;   // query tiles per (batch, head); number of (batch, head) pairs; heads below
    int heads;
    int causal;            // mask keys with index > query index (CLIP text encoders)
    int qpre;              // Q already multiplied by c
    float c;   // softmax scale * log2(e)
    Seg seg[2];
};

constexpr int KT = 64;   // keys per tile

__global__ __launch_bounds__(256) void attn_kernel(const AGeo g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KT * 128];
    char* Ks = smem;                    // [2][64 keys][128 B]
    char* Vs = smem + 2 * KT * 128;     // [2][64 d][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hh = lane >> 5;
    // XCD-aware placement: workgroups b, b+8, ... share an XCD (one L2).  Give each XCD a contiguous run of
    // the (batch, head)-major tile order so a pair's K / V^T is pulled over the fabric by 1-2 L2s, not all 8.
    // (bijective chunked remap: XCD x gets the contiguous run [start_x, start_x + len_x) of the pair-major order)
    int lin;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
    }
    const int pair = lin / g.qtiles;
    const int h = pair % g.heads, b = pair / g.heads;
    const int q0 = (lin % g.qtiles) * 128 + wave * 32;

    // Q fragments: B operand of S^T = K.Q^T -- lane (q, hh) holds d = 16*ks + 8*hh + [0,8)
    f16x8 qf[4];
    {
        int q = q0 + qi;
        if (q >= g.Tq) q = g.Tq - 1;
        const f16* qp = g.Q + (long)b * g.qbs + (long)q * g.ldq + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
    }

    f32x16 oout[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oout[i][r] = 0.f;

    const int srow = lane >> 3, spos = lane & 7;

    for (int sg = 0; sg < g.nseg; ++sg) {
        const Seg s = g.seg[sg];
        const f16* kbase = s.K + (long)b * s.kbs + h * 64;
        const f16* vbase = s.Vt + (long)(h * 64) * s.ldvt + (long)b * s.vbs;
        const int ntiles = (s.Tkv + KT - 1) / KT;
        const int tpad = (s.Tkv + 7) & ~7;   // contract: Vt rows readable and finite on [0, tpad)

        auto stage = [&](int t, int buf) {
            // K tile: 8 glds instructions (8 key rows each); V^T tile: 8 (8 d rows each); 2+2 per wave
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = (i * 4 + wave) * 8;
                int key = t * KT + r0 + srow;
                if (key >= s.Tkv) key = s.Tkv - 1;
                const int kc = spos ^ srow;                        // swz(key) = key & 7
                glds16(kbase + (long)key * s.ldk + kc * 8, Ks + buf * KT * 128 + r0 * 128);
                const int d = r0 + srow;
                const int vc = spos ^ ((d >> 1) & 7);              // swz(d) = (d >> 1) & 7
                int kcol = t * KT + vc * 8;                        // 8 keys per 16-byte chunk
                if (kcol >= tpad) kcol = 0;                        // chunk fully past the end: all its keys are masked
                glds16(vbase + (long)d * s.ldvt + kcol, Vs + buf * KT * 128 + r0 * 128);
            }
        };

        float m = -INFINITY, l = 0.f;
        f32x16 o[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

        stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();

        for (int t = 0; t < ntiles; ++t) {
            const int buf = t & 1;
            if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
            const char* kt = Ks + buf * KT * 128;
            const char* vt = Vs + buf * KT * 128;

            // ---- S^T = K . Q^T  (2 blocks of 32 keys); first k-step accumulates onto the inline constant 0
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const char* krow = kt + (kb * 32 + qi) * 128;
                const int sw = qi & 7;
                f16x8 kf = *(const f16x8*)(krow + ((hh ^ sw) * 16));
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0], (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) {
                    kf = *(const f16x8*)(krow + (((2 * ks + hh) ^ sw) * 16));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[kb], 0, 0, 0);
                }
            }
            // ---- causal mask (CLIP text): key index > query index
            if (g.causal && (t + 1) * KT > q0) {
                const int qq = q0 + qi;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key > qq) sacc[kb][r] = -INFINITY;
                    }
            }
            // ---- tail mask (last tile only; wave-uniform branch)
            if ((t + 1) * KT > s.Tkv) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key >= s.Tkv) sacc[kb][r] = -INFINITY;
                    }
            }
            // ---- online softmax (base 2).  The O / l rescale runs only when some row's running max grew
            //      (alpha == 1 exactly otherwise), which after the first tiles is rare.
            float mx = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx * g.c);
            if (__any(m_new > m)) {
                const float alpha = __builtin_amdgcn_exp2f(m - m_new);
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m = m_new;
            }
            float lsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], g.c, -m));
                    sacc[kb][r] = p;
                    lsum += p;
                }
            l += lsum;

            // ---- O^T += V^T . P^T : k-step (kb, sp) covers keys 32kb+16sp+[0,16) in the permuted order
            //      element j of lane half hh <-> key 16*sp' + 8*(j>>2) + 4*hh + (j&3)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    f16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (f16)sacc[kb][8 * sp + j];
                    const int key_lo = kb * 32 + sp * 16 + 4 * hh;           // 4 keys, then 4 more at +8
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const int d = db * 32 + qi;
                        const int sw = (d >> 1) & 7;
                        const char* row = vt + d * 128;
                        const f16x4 lo = *(const f16x4*)(row + (((key_lo >> 3) ^ sw) * 16) + (key_lo & 7) * 2);
                        const f16x4 hi = *(const f16x4*)(row + ((((key_lo + 8) >> 3) ^ sw) * 16) + (key_lo & 7) * 2);
                        const f16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oout[i][r] += o[i][r] * inv;
    }

    // ---- store: lane (q, hh) holds d = 32*db + 8*gq + 4*hh + [0,4) in regs 4*gq..4*gq+3
    const int q = q0 + qi;
    if (q < g.Tq) {
        f16* op = g.O + (long)b * g.obs + (long)q * g.ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (f16)oout[db][4 * gq + j];
                *(f16x4*)(op + db * 32 + gq * 8 + hh * 4) = v;
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Second-generation kernel (round 2).  Same tiling and operand layout as above; what changed is the per-tile VECTOR work,
// which (not the MFMAs) set the pace at head_dim 64 -- 32 exponentials per 16 MFMAs and lane:
//   * the softmax scale (x log2 e) is folded into the Q fragments once per workgroup, and the running maximum enters the
//     score MFMA as its C operand (a 16-register vector holding -m of the lane's query), so the accumulator leaves the
//     MFMA chain as  s*c - m  and the exponential is applied to it directly: no multiply-add per score;
//   * the running maximum is only raised when some row's tile maximum exceeds it by more than THR (log2 units): P then
//     stays <= 2^THR (fp16 holds that with the same relative precision), and the O / l / -m-vector rescale -- exact, every
//     quantity at the old maximum is scaled by the same 2^-delta once -- runs on a handful of tiles per row instead of
//     being tested and applied per tile;
//   * the first tile of a KV segment (accumulators empty, no maximum yet) and the masked tiles (ragged tail, causal) are
//     separate instantiations, so the steady-state tile body is one branch-free basic block the scheduler can interleave
//     (the first-generation loop carried 66 s_nop hazard pads per tile between its cvt_pk and MFMA instructions).
// Cross-half row maximum by v_permlane32_swap (VALU) instead of ds_bpermute.
constexpr float THR = 5.0f;

template <bool FIRST>
__device__ __forceinline__ void attn_tile(const char* kt, const char* vt, const f16x8 (&qf)[4], const f32x16& c0, const f32x16& c1,
                                          f32x16& negm, float& m, float& l, f32x16 (&o)[2], int qi, int hh) {
    // ---- S'^T = K . (c Q)^T + C   (2 blocks of 32 keys; keys on accumulator rows, query on the lane).  C is -m of the
    //      lane's query in every register (0 on a segment's first tile), or -inf in the registers of masked keys: the mask
    //      costs the tile body nothing.
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const char* krow = kt + (kb * 32 + qi) * 128;
        const int sw = qi & 7;
        f16x8 kf = *(const f16x8*)(krow + ((hh ^ sw) * 16));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0], kb ? c1 : c0, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
            kf = *(const f16x8*)(krow + (((2 * ks + hh) ^ sw) * 16));
            sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], sacc[kb], 0, 0, 0);
        }
    }
    // ---- tile maximum of the row (relative to the running maximum), both lane halves
    float mx, mxb;                                  // two independent v_max3 chains (latency, not count, is what a lone wave pays)
    mx = fmaxf(sacc[0][0], sacc[0][1]);
    mxb = fmaxf(sacc[1][0], sacc[1][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) {
        mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[0][r + 1]);
        mxb = fmaxf(fmaxf(mxb, sacc[1][r]), sacc[1][r + 1]);
    }
    mx = fmaxf(mx, mxb);
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (FIRST) {
        m = mx;                                     // (a row with no visible key cannot occur: key 0 is visible to every query)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] -= mx;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mx;
    } else if (__any(mx > THR)) {
        const float delta = fmaxf(mx, 0.f);         // rows whose maximum did not grow keep delta = 0, alpha = 1 exactly
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] -= delta;
        m += delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m;
    }
    // ---- P = 2^S', row sums (this lane's half of the keys)
    float ls[4] = {0.f, 0.f, 0.f, 0.f};             // four independent partial sums
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sacc[kb][r]);
            sacc[kb][r] = p;
            ls[r & 3] += p;
        }
    l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
    // ---- O^T += V^T . P^T : k-step (kb, sp).  The K rows of the tile were staged in the order that makes accumulator
    //      register 8*sp + j of lane half hh the score of key 32*kb + 16*sp + 8*hh + j (see `stage`), so the packed P
    //      registers are the B operand as they stand and the matching V^T fragment is ONE 16-byte read of 8 consecutive keys.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)sacc[kb][8 * sp + j];
            const int chunk = kb * 4 + sp * 2 + hh;                  // 16-byte chunk of the V^T row: keys 8*chunk + [0,8)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int d = db * 32 + qi;
                const f16x8 vf = *(const f16x8*)(vt + d * 128 + ((chunk ^ ((d >> 1) & 7)) * 16));
                if (FIRST && kb == 0 && sp == 0) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void attn_kernel2(const AGeo g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * KT * 128];
    char* Ks = smem;                    // [2][64 keys][128 B]
    char* Vs = smem + 2 * KT * 128;     // [2][64 d][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hh = lane >> 5;
    int lin;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = blockIdx.x & 7;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
    }
    const int pair = lin / g.qtiles;
    const int h = pair % g.heads, b = pair / g.heads;
    const int q0 = (lin % g.qtiles) * 128 + wave * 32;
    const int qq = q0 + qi;

    // Q fragments, pre-multiplied by scale * log2(e): B operand of S^T = K.Q^T -- lane (q, hh) holds d = 16*ks + 8*hh + [0,8)
    f16x8 qf[4];
    {
        int q = qq;
        if (q >= g.Tq) q = g.Tq - 1;
        const f16* qp = g.Q + (long)b * g.qbs + (long)q * g.ldq + h * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
        if (!g.qpre) {             // callers that own the projection fold c into its weights instead (no second fp16 rounding of q)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[ks][j] = (f16)((float)qf[ks][j] * g.c);
        }
    }

    f16x2 ohold[16];
    const int srow = lane >> 3, spos = lane & 7;

    for (int sg = 0; sg < g.nseg; ++sg) {
        const Seg s = g.seg[sg];
        const f16* kbase = s.K + (long)b * s.kbs + h * 64;
        const f16* vbase = s.Vt + (long)(h * 64) * s.ldvt + (long)b * s.vbs;
        const int ntiles = (s.Tkv + KT - 1) / KT;
        const int tpad = (s.Tkv + 7) & ~7;

        auto stage = [&](int t, int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r0 = (i * 4 + wave) * 8;
                // LDS row rho holds key pi(rho) = rho with bits 2 and 3 swapped: accumulator row (r&3) + 8*(r>>2) + 4*hh of the
                // score MFMA then is key 16*(r>>3) + 8*hh + (r&7) -- a lane's 8 scores of a k-step are 8 CONSECUTIVE keys
                const int rho = r0 + srow;
                int key = t * KT + ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1));
                if (key >= s.Tkv) key = s.Tkv - 1;
                const int kc = spos ^ srow;
                glds16(kbase + (long)key * s.ldk + kc * 8, Ks + buf * KT * 128 + r0 * 128);
                const int d = r0 + srow;
                const int vc = spos ^ ((d >> 1) & 7);
                int kcol = t * KT + vc * 8;
                if (kcol >= tpad) kcol = 0;
                glds16(vbase + (long)d * s.ldvt + kcol, Vs + buf * KT * 128 + r0 * 128);
            }
        };

        float m = 0.f, l = 0.f;
        f32x16 negm, o[2];
        const bool ragged = (s.Tkv % KT) != 0;

        stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        // C operand of a masked tile: `base` (0 on the first tile, -m afterwards) for visible keys, -inf for keys past the
        // end of the segment or (causal) after the query.  Built outside the steady-state loop: ragged tail / CLIP text only.
        auto build_mask = [&](int t, float base, f32x16& c0, f32x16& c1) {
            int key0 = t * KT + 8 * hh;
            asm volatile("" : "+v"(key0));          // opaque here: keeps the 32 per-register key indices from being hoisted to
                                                    // kernel entry as loop invariants (they lived in VGPRs across the hot loop)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + 16 * (r >> 3) + (r & 7);
                    const float v = (key >= s.Tkv || (g.causal && key > qq)) ? -INFINITY : base;
                    if (kb) c1[r] = v; else c0[r] = v;
                }
        };
        auto begin_tile = [&](int t) { if (t + 1 < ntiles) stage(t + 1, (t & 1) ^ 1); };
        auto end_tile = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); };
        const bool mask_all = g.causal != 0;
        {   // tile 0: accumulators empty, no maximum yet
            f32x16 c0, c1;
            begin_tile(0);
            if (mask_all || (ragged && ntiles == 1)) build_mask(0, 0.f, c0, c1);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
            }
            attn_tile<true>(Ks, Vs, qf, c0, c1, negm, m, l, o, qi, hh);
            end_tile();
        }
        if (!mask_all) {
            const int nfull = ragged ? ntiles - 1 : ntiles;
            for (int t = 1; t < nfull; ++t) {                      // steady state: one branch-free body
                begin_tile(t);
                attn_tile<false>(Ks + (t & 1) * KT * 128, Vs + (t & 1) * KT * 128, qf, negm, negm, negm, m, l, o, qi, hh);
                end_tile();
            }
        }
        for (int t = mask_all ? 1 : (ragged && ntiles > 1 ? ntiles - 1 : ntiles); t < ntiles; ++t) {      // masked tiles
            f32x16 c0, c1;
            begin_tile(t);
            build_mask(t, -m, c0, c1);
            attn_tile<false>(Ks + (t & 1) * KT * 128, Vs + (t & 1) * KT * 128, qf, c0, c1, negm, m, l, o, qi, hh);
            end_tile();
        }
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(ltot);
        // normalised output of this segment.  With two segments the first one's result waits as packed fp16 (16 registers
        // instead of 32) -- the reference itself forms the two SDPA outputs as fp16 tensors before adding them
        // (attention_processor.py:1165-1192).
        if (sg + 1 < g.nseg) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) ohold[i * 8 + r] = (f16x2){(f16)(o[i][2 * r] * inv), (f16)(o[i][2 * r + 1] * inv)};
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= inv;
            if (g.nseg == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 8; ++r) { o[i][2 * r] += (float)ohold[i * 8 + r][0]; o[i][2 * r + 1] += (float)ohold[i * 8 + r][1]; }
            }
            // ---- store: lane (q, hh) holds d = 32*db + 8*gq + 4*hh + [0,4) in regs 4*gq..4*gq+3
            if (qq < g.Tq) {
                f16* op = g.O + (long)b * g.obs + (long)qq * g.ldo + h * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f16x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (f16)o[db][4 * gq + j];
                        *(f16x4*)(op + db * 32 + gq * 8 + hh * 4) = v;
                    }
            }
        }
    }
}

}  // namespace

extern "C" int iir_attention_d64_f16(const iir_attn_desc* a, void* stream) {
    (void)hipGetLastError();
    if (!a || !a->Q || !a->O || a->nseg < 1 || a->nseg > 2) return IIR_EINVAL;
    if (a->Tq <= 0 || a->heads <= 0 || a->batch <= 0) return IIR_EINVAL;
    if (a->ldq % 8 || a->ldo % 4) return IIR_EINVAL;
    AGeo g{};
    g.Q = (const f16*)a->Q; g.ldq = a->ldq; g.qbs = a->q_batch_stride;
    g.O = (f16*)a->O; g.ldo = a->ldo; g.obs = a->o_batch_stride;
    g.Tq = a->Tq; g.nseg = a->nseg;
    g.c = a->scale * 1.4426950408889634f;
    g.qpre = a->q_prescaled;
    for (int i = 0; i < a->nseg; ++i) {
        const iir_attn_kv* s = &a->kv[i];
        if (!s->K || !s->Vt || s->Tkv <= 0 || s->ldk % 8 || s->ldvt % 8 || s->vt_batch_stride % 8) return IIR_EINVAL;
        g.seg[i] = Seg{(const f16*)s->K, s->ldk, s->k_batch_stride, (const f16*)s->Vt, s->ldvt, s->vt_batch_stride, s->Tkv};
    }
    g.qtiles = (a->Tq + 127) / 128;
    g.npairs = a->heads * a->batch;
    g.heads = a->heads;
    g.causal = a->causal;
    const dim3 grid(g.npairs * g.qtiles);
    // IIR_ATTN_V (A/B switch): 1 = first-generation kernel, 2 = second generation built for 3 waves per SIMD, 3 = built for 2
    // waves per SIMD (no spills); default 0 = second generation, the 2-wave build when the grid cannot put more than two
    // workgroups on a CU anyway (measured: T = 1024 x 40 pairs 24.8 vs 27.2 us; T = 8192 462 vs 453 us).
    static const int version = getenv("IIR_ATTN_V") ? atoi(getenv("IIR_ATTN_V")) : 0;
    if (version == 1) {
        if (g.qpre) g.c = 1.0f;
        iir_launch(attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    } else if (version == 3 || (version == 0 && grid.x <= 512)) iir_launch(attn_kernel2<2>, grid, dim3(256), 0, (hipStream_t)stream, g);
    else iir_launch(attn_kernel2<3>, grid, dim3(256), 0, (hipStream_t)stream, g);
    return iir_launch_status();
}
