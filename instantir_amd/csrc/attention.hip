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
    f16* O; long ldo, obs; int o_fp8;      // o_fp8: O is a byte matrix of fp8-E4M3 (ldo / obs in bytes)
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

// swizzle of the K tile image (16-byte chunk c of LDS row rho is stored at chunk c ^ KSWZ(rho)).  Rows are 128 bytes, so two rows
// share a 256-byte bank row and a ds_read_b128 lane group (16 lanes = 16 different rows, same logical chunk) is conflict-free
// only if (rho & 1, chunk) differs for all 16: (rho >> 1) & 7 does that for the 32-row fragment reads of this kernel (lane
// groups {0-3, 12-15, 20-27}, ...); the first form, rho & 7, left every K fragment read two-way conflicted.
#ifndef IIR_ATTN_KSWZ_OLD
#define KSWZ(rho) (((rho) >> 1) & 7)
#else
#define KSWZ(rho) ((rho) & 7)
#endif

// One 64-key tile as TWO online-softmax steps of 32 keys (third form, round 2 late).  The 16 MFMAs of a tile used to sit in two
// groups either side of the whole tile's vector work (row maximum over all 64 keys before the first exponential): a serial
// chain  8 MFMA -> softmax -> 8 MFMA  in which neither pipe ever had the other's work to overlap with (5.3: a lone workgroup
// needs ~2000 cycles per tile for 512 cycles of MFMA).  Per half, the dependences are: scores(b) need only the maximum left by
// half a; P.V of half a needs only P(a).  So the score MFMAs of half b are issued next to the exponentials of half a, and the
// P.V MFMAs of half a next to the maximum / exponentials of half b -- same registers (the two 16-register score blocks), same
// LDS traffic, two threshold tests per tile instead of one.
//   FIRST: the first tile of a KV segment (no maximum yet, accumulators empty).  MASKED: c0 / c1 are per-half C operands built by
//   the caller from -m (and -inf for hidden keys) instead of the live -m vector, so a rescale in half a must correct c1 too.
template <bool FIRST, bool MASKED>
__device__ __forceinline__ void attn_tile(const char* kt, const char* vt, const f16x8 (&qf)[4], f32x16& c0, f32x16& c1,
                                          f32x16& negm, float& m, float& l, f32x16 (&o)[2], int qi, int hh) {
    auto scores = [&](int kb, const f32x16& c) {
        const char* krow = kt + (kb * 32 + qi) * 128;
        const int sw = KSWZ(qi);
        f16x8 kf = *(const f16x8*)(krow + ((hh ^ sw) * 16));
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0], c, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
            kf = *(const f16x8*)(krow + (((2 * ks + hh) ^ sw) * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], acc, 0, 0, 0);
        }
        return acc;
    };
    auto row_max = [&](const f32x16& a) {
        float mx = fmaxf(a[0], a[1]), mxb = fmaxf(a[2], a[3]);      // two independent chains
#pragma unroll
        for (int r = 4; r < 16; r += 4) {
            mx = fmaxf(fmaxf(mx, a[r]), a[r + 1]);
            mxb = fmaxf(fmaxf(mxb, a[r + 2]), a[r + 3]);
        }
        mx = fmaxf(mx, mxb);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    // P = 2^S' of one half, its row sum, and O^T += V^T . P^T for the half's two k-steps.  The K rows of the tile were staged in
    // the order that makes accumulator register 8*sp + j of lane half hh the score of key 32*kb + 16*sp + 8*hh + j (see `stage`),
    // so the packed P registers are the B operand as they stand and the matching V^T fragment is ONE 16-byte read.
    auto exp_pv = [&](int kb, f32x16& a, bool zero_o) {
        float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(a[r]);
            a[r] = p;
            ls[r & 3] += p;
        }
        l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)a[8 * sp + j];
            const int chunk = kb * 4 + sp * 2 + hh;                  // 16-byte chunk of the V^T row: keys 8*chunk + [0,8)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int d = db * 32 + qi;
                const f16x8 vf = *(const f16x8*)(vt + d * 128 + ((chunk ^ ((d >> 1) & 7)) * 16));
                if (zero_o && sp == 0) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
                else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[db], 0, 0, 0);
            }
        }
    };
    // raise the running maximum by delta (rows whose maximum did not grow keep delta = 0, alpha = 1 exactly): every quantity at
    // the old maximum is scaled by the same 2^-delta once
    auto rescale = [&](float mx, f32x16& a, bool fix_c1) {
        const float delta = fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] -= delta;
        m += delta;
        {   // in place (tied operand): left to the allocator, the updated vector got new registers and the COMMON path paid
            // 8 v_mov_b64 per half tile plus 16 v_mov at the loop's back edge to carry it
            const float nm = -m;
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_mov_b32 %0, %1" : "+v"(negm[r]) : "v"(nm));
        }
        if (fix_c1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[r] -= delta;             // (-inf stays -inf)
        }
    };

    // ---- half a: scores, maximum
    f32x16 sa = scores(0, c0);
    {
        const float mx = row_max(sa);
        if (FIRST) {
            m = mx;                                 // (a row with no visible key cannot occur: key 0 is visible to every query)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[r] -= mx;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -mx;
            if (MASKED) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c1[r] -= mx;            // the caller built c1 against m = 0
            }
        } else if (__any(mx > THR)) rescale(mx, sa, MASKED);
    }
    // ---- region 2: half b's four score MFMAs, each followed by a quarter of half a's exponentials / sums / packing
    //      (explicit order: the scheduler clustered the MFMAs when left alone)
    f16x8 pa[2], pb[2];
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 sb;
    {
        const char* krow = kt + (32 + qi) * 128;
        const int sw = KSWZ(qi);
        f16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const f16x8*)(krow + (((2 * ks + hh) ^ sw) * 16));
        // (no sched_barrier: pure exp / MFMA intrinsics are not ordered by it)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sb = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], ks ? sb : (MASKED ? c1 : negm), 0, 0, 0);
#pragma unroll
            for (int r = 4 * ks; r < 4 * ks + 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(sa[r]);
                sa[r] = p;
                ls[r & 3] += p;
            }
            if (ks & 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[ks >> 1][j] = (f16)sa[8 * (ks >> 1) + j];
            }

        }
    }
    {
        const float mx = row_max(sb);
        if (__any(mx > THR)) {
            // half a's P is not in O yet: it is scaled with everything else at the old maximum
            const float delta = fmaxf(mx, 0.f);
            const f16 ah = (f16)__builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[sp][j] = pa[sp][j] * ah;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int t = 0; t < 4; ++t) ls[t] *= alpha;
            rescale(mx, sb, false);
        }
    }
    // ---- region 3: half a's four P.V MFMAs, each followed by a quarter of half b's exponentials; then half b's P.V
    {
        f16x8 vf[4];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int d = db * 32 + qi, chunk = sp * 2 + hh;
                vf[sp * 2 + db] = *(const f16x8*)(vt + d * 128 + ((chunk ^ ((d >> 1) & 7)) * 16));
            }
        // (no sched_barrier: pure exp / MFMA intrinsics are not ordered by it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sp = q >> 1, db = q & 1;
            if (FIRST && sp == 0) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[q], pa[sp], (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
            else o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[q], pa[sp], o[db], 0, 0, 0);
#pragma unroll
            for (int r = 4 * q; r < 4 * q + 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(sb[r]);
                sb[r] = p;
                ls[r & 3] += p;
            }
            if (q & 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[q >> 1][j] = (f16)sb[8 * (q >> 1) + j];
            }

        }
        l += (ls[0] + ls[1]) + (ls[2] + ls[3]);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int d = db * 32 + qi, chunk = 4 + sp * 2 + hh;
                const f16x8 v2 = *(const f16x8*)(vt + d * 128 + ((chunk ^ ((d >> 1) & 7)) * 16));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v2, pb[sp], o[db], 0, 0, 0);
            }
    }
}

// NB: depth of the K / V^T tile ring.  With two buffers the loads of tile t+1 have one tile's compute (~0.5 us for a workgroup
// alone on its CU) to land; the 320-workgroup level-2 launches and the cross-attention launches waited ~1 us per tile on them.
// NB - 1 tiles are in flight; the wait at the end of tile t is counted (`vmcnt(4 * tiles still allowed in flight)`), 4 LDS-DMA
// instructions per wave and tile.
// PRE (round 3): the whole KV of every segment fits the ring (<= NB tiles in all: the text + IP cross-attention, 77 + 64 keys
// = 3 tiles).  Every tile of every segment is requested at kernel entry, there is ONE wait and one barrier, and the tile
// bodies then run back to back -- the ring form paid a full load latency per segment plus a wait and a barrier per tile
// for launches whose arithmetic is ~1 us (140 such launches per step, ~18 us each).
template <int WPS, int NB, bool PRE = false>
__global__ __launch_bounds__(256, WPS) void attn_kernel2(const AGeo g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                    // [NB][64 keys][128 B]
    char* Vs = smem + NB * KT * 128;    // [NB][64 d][128 B]

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

    auto stage_seg = [&](const Seg& s, int t, int buf) {
        const f16* kbase = s.K + (long)b * s.kbs + h * 64;
        const f16* vbase = s.Vt + (long)(h * 64) * s.ldvt + (long)b * s.vbs;
        const int tpad = (s.Tkv + 7) & ~7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r0 = (i * 4 + wave) * 8;
            // LDS row rho holds key pi(rho) = rho with bits 2 and 3 swapped: accumulator row (r&3) + 8*(r>>2) + 4*hh of the
            // score MFMA then is key 16*(r>>3) + 8*hh + (r&7) -- a lane's 8 scores of a k-step are 8 CONSECUTIVE keys
            const int rho = r0 + srow;
            int key = t * KT + ((rho & ~12) | ((rho & 4) << 1) | ((rho & 8) >> 1));
            if (key >= s.Tkv) key = s.Tkv - 1;
            const int kc = spos ^ KSWZ(rho);
            glds16(kbase + (long)key * s.ldk + kc * 8, Ks + buf * KT * 128 + r0 * 128);
            const int d = r0 + srow;
            const int vc = spos ^ ((d >> 1) & 7);
            int kcol = t * KT + vc * 8;
            if (kcol >= tpad) kcol = 0;
            glds16(vbase + (long)d * s.ldvt + kcol, Vs + buf * KT * 128 + r0 * 128);
        }
    };
    int pre_slot = 0;                     // PRE: ring slot of the current segment's first tile
    if (PRE) {
        int slot = 0;
        for (int sg = 0; sg < g.nseg; ++sg) {
            const int nt = (g.seg[sg].Tkv + KT - 1) / KT;
            for (int t = 0; t < nt; ++t) stage_seg(g.seg[sg], t, slot++);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }

    for (int sg = 0; sg < g.nseg; ++sg) {
        const Seg s = g.seg[sg];
        const int ntiles = (s.Tkv + KT - 1) / KT;

        auto stage = [&](int t, int buf) { stage_seg(s, t, buf); };

        float m = 0.f, l = 0.f;
        f32x16 negm, o[2];
        const bool ragged = (s.Tkv % KT) != 0;

        if (!PRE) {
#pragma unroll
            for (int i = 0; i < NB - 1; ++i)
                if (i < ntiles) stage(i, i);
        }
        // wait until at most `allowed` later tiles are still in flight (4 LDS-DMA instructions each), then barrier
        auto wait_tiles = [&](int allowed) {
            if (NB >= 4 && allowed >= 2) __builtin_amdgcn_s_waitcnt(0x0F78);
            else if (NB >= 3 && allowed >= 1) __builtin_amdgcn_s_waitcnt(0x0F74);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        };
        if (!PRE) wait_tiles(min(NB - 2, ntiles - 1));
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
        int buf = PRE ? pre_slot : 0;                           // ring slot of the current tile
        auto begin_tile = [&](int t) { if (!PRE && t + NB - 1 < ntiles) stage(t + NB - 1, buf == 0 ? NB - 1 : buf - 1); };   // (t + NB - 1) % NB
        auto end_tile = [&](int t) {                            // tile t + 1 landed for every wave, everyone done with tile t
            if (PRE) { ++buf; return; }                         // everything landed before the first tile; no slot is reused
            wait_tiles(max(0, min(NB - 2, ntiles - 2 - t)));
            buf = buf + 1 == NB ? 0 : buf + 1;
        };
        const bool mask_all = g.causal != 0;
        {   // tile 0: accumulators empty, no maximum yet
            f32x16 c0, c1;
            begin_tile(0);
            if (mask_all || (ragged && ntiles == 1)) {
                build_mask(0, 0.f, c0, c1);
                attn_tile<true, true>(Ks + buf * KT * 128, Vs + buf * KT * 128, qf, c0, c1, negm, m, l, o, qi, hh);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = 0.f;
                attn_tile<true, false>(Ks + buf * KT * 128, Vs + buf * KT * 128, qf, c0, c0, negm, m, l, o, qi, hh);
            }
            end_tile(0);
        }
        if (!mask_all) {
            const int nfull = ragged ? ntiles - 1 : ntiles;
            for (int t = 1; t < nfull; ++t) {                      // steady state: one branch-free body
                begin_tile(t);
                attn_tile<false, false>(Ks + buf * KT * 128, Vs + buf * KT * 128, qf, negm, negm, negm, m, l, o, qi, hh);
                end_tile(t);
            }
        }
        for (int t = mask_all ? 1 : (ragged && ntiles > 1 ? ntiles - 1 : ntiles); t < ntiles; ++t) {      // masked tiles
            f32x16 c0, c1;
            begin_tile(t);
            build_mask(t, -m, c0, c1);
            attn_tile<false, true>(Ks + buf * KT * 128, Vs + buf * KT * 128, qf, c0, c1, negm, m, l, o, qi, hh);
            end_tile(t);
        }
        pre_slot += ntiles;
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
                char* op8 = (char*)g.O + (long)b * g.obs + (long)qq * g.ldo + h * 64;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f16x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (f16)o[db][4 * gq + j];
                        if (g.o_fp8) *(int*)(op8 + db * 32 + gq * 8 + hh * 4) = iir_fp8x4(v);      // rounded to fp16 first, as the fp16 output would be
                        else *(f16x4*)(op + db * 32 + gq * 8 + hh * 4) = v;
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
    g.O = (f16*)a->O; g.ldo = a->ldo; g.obs = a->o_batch_stride; g.o_fp8 = a->o_fp8 != 0;
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
    if (g.o_fp8 && version == 1) return IIR_EINVAL;          // the first-generation kernel has no fp8 store
    // (ring depths 3 and 4 were built and measured: no change on any of the step's shapes, `profiles/r02_attn_ring_depth.log`;
    //  only the two-buffer instantiations are compiled)
    auto launch2 = [&](auto kern, int nb) {
        const size_t lds = (size_t)nb * 2 * KT * 128;
        // (32 KB of dynamic LDS: below the 64 KB that needs hipFuncAttributeMaxDynamicSharedMemorySize)
        iir_launch(kern, grid, dim3(256), lds, (hipStream_t)stream, g);
    };
    int total_tiles = 0;
    for (int i = 0; i < a->nseg; ++i) total_tiles += (a->kv[i].Tkv + KT - 1) / KT;
    static const bool pre_on = !(getenv("IIR_ATTN_PRE") && atoi(getenv("IIR_ATTN_PRE")) == 0);
    if (version != 1 && pre_on && !g.causal && total_tiles <= 4) {
        // short KV (text + IP cross-attention): every tile requested at entry, one wait, no ring (see attn_kernel2<.., PRE>)
        static int attr_dev = -1;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (attr_dev != dev) {
            if (hipFuncSetAttribute((const void*)attn_kernel2<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * KT * 128) != hipSuccess) return IIR_ELAUNCH;
            attr_dev = dev;
        }
        launch2(attn_kernel2<2, 4, true>, 4);
        return iir_launch_status();
    }
    if (version == 1) {
        if (g.qpre) g.c = 1.0f;
        iir_launch(attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    } else if (version == 3 || (version == 0 && grid.x <= 512)) launch2(attn_kernel2<2, 2>, 2);
    else launch2(attn_kernel2<3, 2>, 2);
    return iir_launch_status();
}
