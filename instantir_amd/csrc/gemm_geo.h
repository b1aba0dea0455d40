// Per-launch constants shared by the GEMM / implicit-GEMM convolution kernels (gemm_conv.hip, gemm8.hip).
#pragma once
#include "common.h"

namespace iir {

typedef unsigned iir_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte store of an output row chunk.  `wt`: write-through (`buffer_store_dwordx4 ... sc1`).  A GEMM leaves MBs of dirty lines
// in the eight L2s; the kernel's end-of-dispatch release has to write them back before the next (dependent) launch can start --
// time after the last workgroup has finished (MI355X_MICROARCH.md, row publish-large: 8.2 vs 3.0 us for 2 MB per XCD).  The
// consumer is another launch on other XCDs anyway (it reads through the Infinity Cache), so nothing is lost by not keeping the
// lines.
template <typename V>
__device__ __forceinline__ void store16(void* base, long byte_off, const V& v, bool wt) {
    static_assert(sizeof(V) == 16, "16-byte chunk");
    if (wt) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(iir_u32x4, v), r, (int)byte_off, 0, 16);
    } else {
        *(V*)((char*)base + byte_off) = v;
    }
}

struct Geo {   // per-launch constants shared by GEMM and CONV paths
    const f16* A; long lda;
    const f16* W;
    f16* C; long ldc;
    int M, N, K;
    const f16* bias;
    const f16* rowbias; long ldrb; int rows_per_rb;
    const f16* res; long ldr;
    int epi, act;
    float out_scale;
    // conv
    int H, Wd, Cin, Ho, Wo, ks, stride, pad, ups;
    const f16* zero;
    long x_img_stride;        // elements between input images
    int y_img_rows, res_img_rows;   // rows between images in C / res (conv mode); 0 = dense
    int tiles_m, tiles_n;
    int xm, rm, rn;           // XCD partition: xm x (8/xm) rectangles of rm x rn tiles
    const char* pf; int pf_lines;   // weight prefetch: 128-byte lines to pull towards the Infinity Cache
    int splitk;                     // 1 or 2 K slices per output tile (workgroups z = 0 / 1 of a tile share an XCD)
    long sk_bytes;                  // host only: bytes of the caller's split-K workspace
    float* sk_slabs; int* sk_cnt;   // split-K workspace: fp32 partial tiles [tile][z][BM*BN] and per-tile arrival counters
    f16* Ct; long ldct; int tr_from, ct_vec;   // columns n >= tr_from are stored transposed: Ct[(n - tr_from) * ldct + m]
    int c_vec, r_vec;               // C / res rows allow 16-byte accesses (ld % 8 == 0, base 16-byte aligned)
    int dtype;                      // IIR_DT_F16 / IIR_DT_BF16: element type of A, W, C, bias, rowbias, res
    int c_f32;                      // C is float (plain epilogue, out_scale only): the VAE's attention scores
    const float* wscale;            // W8 build: W holds fp8-E4M3 bytes [N][K], wscale[n] its per-output-channel scale (fp32)
    // LayerNorm folded into the GEMMs either side of it ("LayerNorm without a LayerNorm launch"):
    float* ln_out;                  // producer: per (column tile, row) partial (mean, M2) of the rows it writes, [N/BN][M] float2
    const float* ln_in;             // consumer: those partials; A holds the RAW rows, W has gamma folded in
    int ln_parts, ln_part_cols;     //   partial count per row and the columns each one covers
    float ln_eps;
    float* gn_out;                  // GroupNorm statistics of the rows this launch writes: [M / 64][N] (mean, M2) per 64-row slab and channel
    int st_wt;                      // write the output through to memory (sc1 stores): nothing is left dirty in the L2s for the end-of-kernel write-back
    const float* ln_colsum;         //   s[n] = sum_k W[n][k] (fp32): y = rstd * (x . w_n) - rstd * mean * s[n] (+ bias, which carries W . beta)
    // IIR_EPI_XATTN (64 x 128 tile): the finished tile is a q projection (two heads x 64 rows, scale x log2 e folded in by the caller);
    // the workgroup runs the text + IP cross-attention of its rows and heads on it and stores the attention output instead
    int xa_on, xa_tq;               // xa_tq: query rows per image (tile rows never straddle an image: xa_tq % 64 == 0)
    const f16* xa_k[2]; long xa_ldk[2], xa_kb[2];       // K[img][t][head * 64 + d]   (segment 0: text, 1: IP tokens)
    const f16* xa_vt[2]; long xa_ldvt[2], xa_vb[2];     // Vt[head * 64 + d][img * vb + t], readable on [0, roundup8(Tkv))
    int xa_tk[2];                   // 1 <= Tkv[0] <= 80, 1 <= Tkv[1] <= 64
    int c_fp8;                      // C is a byte matrix of fp8-E4M3 (ldc in bytes): stored by the fast write-out path only
    int f8; float a_scale;          // all-fp8 build: A and W are E4M3 bytes, K / lda counted in 2-byte units; C = (A8 . W8^T) * wscale[n] * a_scale
};

// gemm8.hip: 256 x BN tile, 8 waves, two-tile-deep LDS-DMA pipeline (BN = 320 or 256).  Returns IIR_EINVAL when the launch is
// outside what that kernel covers (the caller then takes the 4-wave kernel).
int gemm8_launch(const Geo& g, int bn, hipStream_t stream);
bool gemm8_covers(const Geo& g, int bn);

}  // namespace iir
