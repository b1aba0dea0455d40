// K5/K6: GroupNorm(+SiLU) over NHWC images and LayerNorm(+adaLN modulation) over token rows.
// HBM-bound kernels: 16-byte vector accesses, fp32 statistics, wavefront / LDS reductions.
//
// Replaces: nn.GroupNorm call sites of ResnetBlock2D (module/min_sdxl.py:245,252; eps 1e-5, SiLU
// fused: :257,:269-271), Transformer2DModel.norm (:568, eps 1e-6), conv_norm_out (:841); nn.LayerNorm
// call sites of BasicTransformerBlock (:534-538), Resampler (module/ip_adapter/resampler.py:15,43-44,98)
// and AdaLayerNorm (module/ip_adapter/attention_processor.py:18-25: LN without affine, eps 1e-6,
// x * (1 + scale) + shift).
#include "common.h"
#include <stdlib.h>
#include "../../include/instantir_hip.h"

namespace {

constexpr int GN_MAXC = 2560;

// Statistics are carried as (count, mean, M2 = sum of squared deviations) and merged with Chan's pairwise update in a FIXED
// order: no E[x^2] - mean^2 cancellation (real SDXL activations have group means far above their spread), and bitwise
// reproducible run to run.  Inside a thread the sums are taken about the thread's first sample, which is of the size of the
// data, so the one-pass form is as well conditioned as a two-pass one.
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb <= 0.f) return;
    const float tot = n + nb, d = mb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}

// ---- GroupNorm pass 1: per (image, slab, group) partial (mean, M2) ------------------------------------
// grid (nslab, R); each thread owns fixed 8-channel chunks so channel sums live in registers.
template <typename E>
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* Xp, long ldx, int HW, int C, int G, int pix_per_slab,
                                                       float* part /*[R][nslab][G][2]*/) {
    using E8 = typename ET<E>::x8;
    const E* X = (const E*)Xp;
    __shared__ float cmean[2048], cm2[2048];                // per (pixel-row lane, channel) partials: prows * C <= 2048
    __shared__ float chmean[GN_MAXC], chm2[GN_MAXC];
    const int r = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int nchunk = C >> 3;
    const int p0 = slab * pix_per_slab;
    const int p1 = min(HW, p0 + pix_per_slab);
    const E* base = X + (long)r * HW * ldx;
    const int lanes_c = nchunk < 256 ? nchunk : 256;
    const int prows = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, tp = threadIdx.x / lanes_c;
    for (int c0 = 0; c0 < nchunk; c0 += lanes_c) {            // one pass unless C > 2048
        const int ch = c0 + tc;
        if (tp < prows && ch < nchunk) {
            float s[8], q[8], k0[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; k0[j] = 0.f; }
            int p = p0 + tp, cnt = 0;
            if (p < p1) {
                const E8 v = *(const E8*)(base + (long)p * ldx + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) k0[j] = (float)v[j];
            }
            for (; p + 3 * prows < p1; p += 4 * prows) {            // 4 loads in flight per thread
                E8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *(const E8*)(base + (long)(p + u * prows) * ldx + ch * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = (float)v[u][j] - k0[j]; s[j] += f; q[j] += f * f; }
                cnt += 4;
            }
            for (; p < p1; p += prows) {
                const E8 v = *(const E8*)(base + (long)p * ldx + ch * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j] - k0[j]; s[j] += f; q[j] += f * f; }
                cnt += 1;
            }
            const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cmean[(tp * lanes_c + tc) * 8 + j] = k0[j] + s[j] * inv;
                cm2[(tp * lanes_c + tc) * 8 + j] = fmaxf(q[j] - s[j] * s[j] * inv, 0.f);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < lanes_c * 8 && c0 * 8 + c < C; c += 256) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
            for (int t = 0; t < prows; ++t) {               // pixel-row lane t saw pixels p0 + t, p0 + t + prows, ...
                const int first = p0 + t;
                const float nt = first < p1 ? (float)((p1 - first + prows - 1) / prows) : 0.f;
                chan_merge(n, mean, m2, nt, cmean[t * lanes_c * 8 + c], cm2[t * lanes_c * 8 + c]);
            }
            chmean[c0 * 8 + c] = mean; chm2[c0 * 8 + c] = m2;
        }
        __syncthreads();
    }
    const int cpg = C / G;
    const float npix = (float)(p1 - p0);
    for (int gi = threadIdx.x; gi < G; gi += 256) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int c = gi * cpg; c < (gi + 1) * cpg; ++c) chan_merge(n, mean, m2, npix, chmean[c], chm2[c]);
        float* o = part + (((long)r * nslab + slab) * G + gi) * 2;
        o[0] = mean; o[1] = m2;
    }
}

// ---- GroupNorm pass 2: merge the slab partials to mean / rstd per (image, group) ------------------
// One workgroup per (image, 8 groups): thread = (group g8 of 8, slab lane sl of 32); every lane issues ALL its loads (slabs
// sl, sl + 32, ...: at most 8) before the first merge, merges them in order, and the 32 lane results of a group go through a
// fixed two-level tree in LDS -- reproducible, no serial chain of dependent global loads (the first form, one 256-thread
// workgroup per image walking 32 slabs per thread with a load -> merge -> load chain, took 11 us per launch, 113 per step).
// (An intermediate form, one wave per (image, group) with a `__shfl_xor` butterfly, produced run-to-run different statistics
//  whenever a conv kernel of the other stream shared the chip -- `tools/racecheck_concurrent.py`; not root-caused, replaced.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* part, int nslab, int G, int HW, int pix_per_slab, int cpg,
                                                          float eps, float* stat /*[R][G][2]*/, int gblocks) {
    __shared__ float sn[256], smean[256], sm2[256];
    const int r = blockIdx.x / gblocks, gb = blockIdx.x - r * gblocks;
    const int g8 = threadIdx.x & 7, sl = threadIdx.x >> 3;       // 8 groups x 32 slab lanes
    const int gi = gb * 8 + g8;
    constexpr int MAXS = 8;                                      // nslab <= 256
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (gi < G) {
        const float* base = part + ((long)r * nslab * G + gi) * 2;
        float2 a[MAXS];
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int s_ = min(sl + k * 32, nslab - 1);          // clamped: unconditional loads, one batch
            a[k] = *(const float2*)(base + (long)s_ * G * 2);
        }
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int s_ = sl + k * 32;
            if (s_ < nslab) {
                const int p0 = s_ * pix_per_slab, p1 = min(HW, p0 + pix_per_slab);
                chan_merge(n, mean, m2, (float)(p1 - p0) * (float)cpg, a[k].x, a[k].y);
            }
        }
    }
    sn[threadIdx.x] = n; smean[threadIdx.x] = mean; sm2[threadIdx.x] = m2;
    __syncthreads();
    if (threadIdx.x < 64) {                                      // level 1: (group g8, segment seg of 8) merges 4 slab lanes
        const int seg = threadIdx.x >> 3;
        float tn = 0.f, tm = 0.f, tq = 0.f;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int src = (seg * 4 + l) * 8 + g8;
            chan_merge(tn, tm, tq, sn[src], smean[src], sm2[src]);
        }
        n = tn; mean = tm; m2 = tq;
    }
    __syncthreads();
    if (threadIdx.x < 64) { sn[threadIdx.x] = n; smean[threadIdx.x] = mean; sm2[threadIdx.x] = m2; }
    __syncthreads();
    if (threadIdx.x < 8 && gi < G) {                             // level 2: the 8 segments of a group
        float tn = 0.f, tm = 0.f, tq = 0.f;
#pragma unroll
        for (int sgm = 0; sgm < 8; ++sgm) chan_merge(tn, tm, tq, sn[sgm * 8 + g8], smean[sgm * 8 + g8], sm2[sgm * 8 + g8]);
        stat[((long)r * G + gi) * 2] = tm;
        stat[((long)r * G + gi) * 2 + 1] = rsqrtf(tq / tn + eps);
    }
}

// ---- GroupNorm pass 2': (mean, rstd) per (image, group) from the PRODUCER's column partials (round 3): the launch that wrote
// the tensor left (mean, M2) of every channel per 64-row slab (`gn_stats_out`); one workgroup per (image, group) merges its
// slabs x channels -- equal counts, so mean = average of the means and M2 = sum M2 + 64 * sum (mean_i - mean)^2, two block
// reductions in a fixed tree.  All loads unconditional and clamped (section 5.8 rule), masked at the merge.
__global__ __launch_bounds__(256) void gn_finalize_cols_kernel(const float2* part, long ldp, int slabs_per_img, int G, int cpg, float eps,
                                                               float* stat /*[R][G][2]*/) {
    __shared__ float red[256];
    const int r = blockIdx.x / G, gi = blockIdx.x - r * G;
    const int n = slabs_per_img * cpg;                                 // partials of this (image, group), 64 samples each
    constexpr int MAXP = 20;                                           // n <= 5120 (512 slabs x 10 channels: the Aggregator's 256 x 128 level-0 maps)
    float2 a[MAXP];
    const float2* base = part + (long)r * slabs_per_img * ldp + gi * cpg;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int i = min((int)threadIdx.x + k * 256, n - 1);
        const int sl = i / cpg, c = i - sl * cpg;
        a[k] = base[(long)sl * ldp + c];
    }
    auto block_sum = [&](float v) {
        red[threadIdx.x] = v;
        __syncthreads();
#pragma unroll
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        const float t = red[0];
        __syncthreads();
        return t;
    };
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) sm += ((int)threadIdx.x + k * 256 < n) ? a[k].x : 0.f;
    const float mean = block_sum(sm) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const float d = a[k].x - mean;
        q += ((int)threadIdx.x + k * 256 < n) ? fmaf(64.0f * d, d, a[k].y) : 0.f;
    }
    const float m2 = block_sum(q);
    if (threadIdx.x == 0) {
        stat[((long)r * G + gi) * 2] = mean;
        stat[((long)r * G + gi) * 2 + 1] = rsqrtf(m2 / ((float)n * 64.0f) + eps);
    }
}

// ---- GroupNorm pass 3: normalize + affine (+SiLU).  A thread owns fixed 8-channel chunks, so its 16
// scale/shift coefficients live in registers and the pixel loop is load - 8 fma - store.
template <typename E>
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* Xp, long ldx, f16* Yp, long ldy, int HW, int C, int G,
                                                       int pix_per_blk, const float* stat, const f16* gammap,
                                                       const f16* betap, int silu) {
    using E8 = typename ET<E>::x8;
    const E* X = (const E*)Xp; E* Y = (E*)Yp; const E* gamma = (const E*)gammap; const E* beta = (const E*)betap;
    const int r = blockIdx.y;
    const int cpg = C / G;
    const int nchunk = C >> 3;
    const int lanes_c = nchunk < 256 ? nchunk : 256;
    const int prows = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, tp = threadIdx.x / lanes_c;
    if (tp >= prows) return;
    const int p0 = blockIdx.x * pix_per_blk;
    const int p1 = min(HW, p0 + pix_per_blk);
    const E* xb = X + (long)r * HW * ldx;
    E* yb = Y + (long)r * HW * ldy;
    const float* st = stat + (long)r * G * 2;
    for (int ch = tc; ch < nchunk; ch += lanes_c) {
        float a[8], b[8];
        const E8 gm = *(const E8*)(gamma + ch * 8), bt = *(const E8*)(beta + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gi = (ch * 8 + j) / cpg;
            const float mean = st[gi * 2], rstd = st[gi * 2 + 1];
            a[j] = rstd * (float)gm[j];
            b[j] = (float)bt[j] - mean * a[j];
        }
        for (int p = p0 + tp; p < p1; p += prows) {
            const E8 v = *(const E8*)(xb + (long)p * ldx + ch * 8);
            E8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = fmaf((float)v[j], a[j], b[j]);
                if (silu) f = silu_f(f);
                o[j] = (E)f;
            }
            *(E8*)(yb + (long)p * ldy + ch * 8) = o;
        }
    }
}

// ---- LayerNorm: one wave per row, row kept in registers (C <= 2560) -------------------------------
// y = LN(x) * gamma + beta                      (gamma/beta optional)
// y = y * (1 + scale[row / rows_per_mod]) + shift[row / rows_per_mod]   (optional adaLN modulation)
// transposed != 0 writes y^T: Y[c][col_off(row)] with col = (row / tr_rows) * tr_bstride + row % tr_rows.
__device__ __forceinline__ void ln_row(const f16* X, long ldx, f16* Y, long ldy, int row, int C, const f16* gamma,
                                       const f16* beta, float eps, const f16* shift, const f16* scale, long ldmod,
                                       int rows_per_mod, int transposed, int tr_rows, long tr_bstride) {
    const int lane = threadIdx.x & 63;
    const int nchunk = C >> 3;
    constexpr int MAXK = GN_MAXC / 8 / 64;   // 5 chunks per lane
    f16x8 v[MAXK];
    float s = 0.f;
    const f16* x = X + (long)row * ldx;
    // (unconditional, clamped loads: the form `if (ch < nchunk) v[k] = load` -- one predicated load per basic block feeding a
    //  `__shfl_xor` reduction -- is the one that produced run-to-run different results beside a concurrent 64-row-tile GEMM in
    //  the GroupNorm finalize kernel; this kernel never showed it, the form is avoided all the same)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) v[k] = *(const f16x8*)(x + min(lane + k * 64, nchunk - 1) * 8);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = lane + k * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)v[k][j];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = lane + k * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)v[k][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const f16* sh = shift ? shift + (long)(row / rows_per_mod) * ldmod : nullptr;
    const f16* sc = scale ? scale + (long)(row / rows_per_mod) * ldmod : nullptr;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = lane + k * 64;
        if (ch < nchunk) {
            f16x8 o;
            f16x8 gm, bt, s8, c8;
            if (gamma) gm = *(const f16x8*)(gamma + ch * 8);
            if (beta) bt = *(const f16x8*)(beta + ch * 8);
            if (sh) { s8 = *(const f16x8*)(sh + ch * 8); c8 = *(const f16x8*)(sc + ch * 8); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = ((float)v[k][j] - mean) * rstd;
                if (gamma) f *= (float)gm[j];
                if (beta) f += (float)bt[j];
                if (sh) f = f * (1.0f + (float)c8[j]) + (float)s8[j];
                o[j] = (f16)f;
            }
            if (transposed == 2) {          // fp8-E4M3 bytes (the A operand of an all-fp8 GEMM): Y is a byte matrix, ldy in bytes
                *(long*)((char*)Y + (long)row * ldy + ch * 8) = iir_fp8x8(o);
            } else if (!transposed) {
                *(f16x8*)(Y + (long)row * ldy + ch * 8) = o;
            } else {
                const long col = (long)(row / tr_rows) * tr_bstride + (row % tr_rows);
#pragma unroll
                for (int j = 0; j < 8; ++j) Y[(long)(ch * 8 + j) * ldy + col] = o[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void ln_kernel(const f16* X, long ldx, f16* Y, long ldy, int rows, int C, const f16* gamma,
                                                 const f16* beta, float eps, const f16* shift, const f16* scale, long ldmod,
                                                 int rows_per_mod, int transposed, int tr_rows, long tr_bstride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    ln_row(X, ldx, Y, ldy, row, C, gamma, beta, eps, shift, scale, ldmod, rows_per_mod, transposed, tr_rows, tr_bstride);
}

// Every adaLN of one UNet forward in one launch: blockIdx.y walks a device table of iir_adaln_job records.
__global__ __launch_bounds__(256) void adaln_batch_kernel(const iir_adaln_job* jobs, int rows, float eps, long ldmod,
                                                          int rows_per_mod, int tr_rows, long tr_bstride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const iir_adaln_job j = jobs[blockIdx.y];
    ln_row((const f16*)j.X, j.ldx, (f16*)j.Y, j.ldy, row, j.C, nullptr, nullptr, eps, (const f16*)j.shift, (const f16*)j.scale,
           ldmod, rows_per_mod, j.transposed, tr_rows, tr_bstride);
}

// ---- row softmax in place (VAE mid-block attention scores: one head of dim C, T up to 16384) --------
// one workgroup per row; rows up to 16384 columns stay in registers between the three phases.
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* X, long ld, int cols) {
    __shared__ float red[4];
    f16* x = X + (long)blockIdx.x * ld;
    const int nchunk = cols >> 3;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int MAXK = 8;
    f16x8 v[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) v[k] = *(const f16x8*)(x + min((int)threadIdx.x + k * 256, nchunk - 1) * 8);      // unconditional, clamped (see ln_row)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[k][j]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float e[MAXK][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { e[k][j] = __expf((float)v[k][j] - mx); sum += e[k][j]; }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (f16)(e[k][j] * inv);
            *(f16x8*)(x + ch * 8) = o;
        }
    }
}

// ---- row softmax of fp32 scores, written as fp16 / bf16 probabilities (VAE mid-block attention: the scores never leave
// fp32 before the softmax, as in the reference's fp32 VAE).  One workgroup per row, the row stays in registers.
template <typename E>
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* S, long lds, f16* Pp, long ldp, int cols) {
    using E4 = typename ET<E>::x4;
    __shared__ float red[4];
    const float* x = S + (long)blockIdx.x * lds;
    E* y = (E*)Pp + (long)blockIdx.x * ldp;
    const int nchunk = cols >> 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int MAXK = 16;                        // 256 threads x 4 floats x 16 = 16384 columns
    f32x4 v[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) v[k] = *(const f32x4*)(x + min((int)threadIdx.x + k * 256, nchunk - 1) * 4);     // unconditional, clamped (see ln_row)
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) mx = fmaxf(fmaxf(mx, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[k][j] = __expf(v[k][j] - mx); sum += v[k][j]; }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int ch = threadIdx.x + k * 256;
        if (ch < nchunk) {
            E4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (E)(v[k][j] * inv);
            *(E4*)(y + ch * 4) = o;
        }
    }
}

}  // namespace

extern "C" int iir_softmax_rows_f32(const float* S, int64_t lds, void* P, int64_t ldp, int32_t rows, int32_t cols, int32_t dtype,
                                    void* stream) {
    (void)hipGetLastError();
    if (!S || !P || rows <= 0 || cols <= 0 || cols % 4 || cols > 16384 || lds % 4 || ldp % 4) return IIR_EINVAL;
    if (dtype == IIR_DT_BF16) hipLaunchKernelGGL(softmax_rows_f32_kernel<bf16>, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, (long)lds, (f16*)P, (long)ldp, cols);
    else if (dtype == IIR_DT_F16) hipLaunchKernelGGL(softmax_rows_f32_kernel<f16>, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, (long)lds, (f16*)P, (long)ldp, cols);
    else return IIR_EINVAL;
    return iir_launch_status();
}

extern "C" int iir_softmax_rows_f16(void* X, int64_t ld, int32_t rows, int32_t cols, void* stream) {
    (void)hipGetLastError();
    if (!X || rows <= 0 || cols <= 0 || cols % 8 || cols > 16384 || ld % 8) return IIR_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (f16*)X, (long)ld, cols);
    return iir_launch_status();
}

extern "C" int iir_groupnorm_nhwc(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R, int32_t HW, int32_t C,
                                  int32_t groups, const void* gamma, const void* beta, float eps, int32_t silu,
                                  void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !Y || !gamma || !beta || !workspace) return IIR_EINVAL;
    if (C % 8 || C > GN_MAXC || groups <= 0 || groups > 64 || C % groups || ldx % 8 || ldy % 8) return IIR_EINVAL;
    if (R <= 0 || HW <= 0 || (dtype != IIR_DT_F16 && dtype != IIR_DT_BF16)) return IIR_EINVAL;
    // slabs: enough blocks to fill the chip, at least 8 pixels each (a 32x32 map of 1280-2560 channels took 14.6 us with
    // 64 workgroups walking 32 pixels each, 8 us with 256 walking 8)
    static const int min_pix = getenv("IIR_GN_MINPIX") ? atoi(getenv("IIR_GN_MINPIX")) : 8;
    int nslab = (HW + min_pix - 1) / min_pix;
    const int want = (1024 + R - 1) / R;
    if (nslab > want) nslab = want;
    if (nslab > 256) nslab = 256;
    if (nslab < 1) nslab = 1;
    const int pps = (HW + nslab - 1) / nslab;
    nslab = (HW + pps - 1) / pps;
    const int64_t part_floats = (int64_t)R * nslab * groups * 2;
    if ((part_floats + (int64_t)R * groups * 2) * 4 > workspace_bytes) return IIR_EINVAL;
    float* part = (float*)workspace;
    float* stat = part + part_floats;
    const hipStream_t st = (hipStream_t)stream;
    // apply: ~1024 blocks over the batch
    int nblk = (1024 + R - 1) / R;
    static const int min_ppb = getenv("IIR_GN_APPLY_MINPIX") ? atoi(getenv("IIR_GN_APPLY_MINPIX")) : 8;
    int ppb = (HW + nblk - 1) / nblk; if (ppb < min_ppb) ppb = min_ppb;
    nblk = (HW + ppb - 1) / ppb;
    if (dtype == IIR_DT_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16>, dim3(nslab, R), dim3(256), 0, st, (const f16*)X, (long)ldx, HW, C, groups, pps, part);
    else hipLaunchKernelGGL(gn_stats_kernel<f16>, dim3(nslab, R), dim3(256), 0, st, (const f16*)X, (long)ldx, HW, C, groups, pps, part);
    const int gblocks = (groups + 7) / 8;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(R * gblocks), dim3(256), 0, st, (const float*)part, nslab, groups, HW, pps, C / groups, eps, stat, gblocks);
    if (dtype == IIR_DT_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16>, dim3(nblk, R), dim3(256), 0, st, (const f16*)X, (long)ldx, (f16*)Y, (long)ldy, HW, C, groups, ppb, (const float*)stat, (const f16*)gamma, (const f16*)beta, silu);
    else hipLaunchKernelGGL(gn_apply_kernel<f16>, dim3(nblk, R), dim3(256), 0, st, (const f16*)X, (long)ldx, (f16*)Y, (long)ldy, HW, C, groups, ppb, (const float*)stat, (const f16*)gamma, (const f16*)beta, silu);
    return iir_launch_status();
}

extern "C" int iir_groupnorm_from_partials(const void* partials, int64_t ldp, const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R,
                                           int32_t HW, int32_t C, int32_t groups, const void* gamma, const void* beta, float eps,
                                           int32_t silu, void* workspace, int64_t workspace_bytes, int32_t dtype, void* stream) {
    (void)hipGetLastError();
    if (!partials || !X || !Y || !gamma || !beta || !workspace || (uintptr_t)partials % 8) return IIR_EINVAL;
    if (C % 8 || C > GN_MAXC || groups <= 0 || groups > 64 || C % groups || ldx % 8 || ldy % 8 || ldp < C) return IIR_EINVAL;
    if (R <= 0 || HW <= 0 || HW % 64 || (dtype != IIR_DT_F16 && dtype != IIR_DT_BF16)) return IIR_EINVAL;
    const int slabs = HW / 64, cpg = C / groups;
    if ((long)slabs * cpg > 20 * 256) return IIR_EINVAL;
    if ((int64_t)R * groups * 2 * 4 > workspace_bytes) return IIR_EINVAL;
    float* stat = (float*)workspace;
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(R * groups), dim3(256), 0, st, (const float2*)partials, (long)ldp, slabs, groups, cpg, eps, stat);
    int nblk = (1024 + R - 1) / R;
    static const int min_ppb = getenv("IIR_GN_APPLY_MINPIX") ? atoi(getenv("IIR_GN_APPLY_MINPIX")) : 8;
    int ppb = (HW + nblk - 1) / nblk; if (ppb < min_ppb) ppb = min_ppb;
    nblk = (HW + ppb - 1) / ppb;
    if (dtype == IIR_DT_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16>, dim3(nblk, R), dim3(256), 0, st, (const f16*)X, (long)ldx, (f16*)Y, (long)ldy, HW, C, groups, ppb, (const float*)stat, (const f16*)gamma, (const f16*)beta, silu);
    else hipLaunchKernelGGL(gn_apply_kernel<f16>, dim3(nblk, R), dim3(256), 0, st, (const f16*)X, (long)ldx, (f16*)Y, (long)ldy, HW, C, groups, ppb, (const float*)stat, (const f16*)gamma, (const f16*)beta, silu);
    return iir_launch_status();
}

extern "C" int iir_groupnorm_nhwc_f16(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t R, int32_t HW, int32_t C,
                                      int32_t groups, const void* gamma, const void* beta, float eps, int32_t silu,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    return iir_groupnorm_nhwc(X, ldx, Y, ldy, R, HW, C, groups, gamma, beta, eps, silu, workspace, workspace_bytes, IIR_DT_F16, stream);
}

extern "C" int64_t iir_groupnorm_workspace_bytes(int32_t R, int32_t groups) { return (int64_t)R * 257 * groups * 2 * 4; }

extern "C" int iir_layernorm_f16(const void* X, int64_t ldx, void* Y, int64_t ldy, int32_t rows, int32_t C, const void* gamma,
                                 const void* beta, float eps, const void* shift, const void* scale, int64_t ldmod,
                                 int32_t rows_per_mod, int32_t transposed, int32_t tr_rows, int64_t tr_bstride, void* stream) {
    (void)hipGetLastError();
    if (!X || !Y || rows <= 0 || C % 8 || C > GN_MAXC || ldx % 8) return IIR_EINVAL;
    if ((shift == nullptr) != (scale == nullptr)) return IIR_EINVAL;
    if (shift && (rows_per_mod <= 0 || ldmod % 8)) return IIR_EINVAL;
    if (transposed < 0 || transposed > 2) return IIR_EINVAL;
    if (transposed != 1 && ldy % 8) return IIR_EINVAL;       // (2: fp8 bytes out, 8 per store)
    if (transposed == 1 && tr_rows <= 0) return IIR_EINVAL;
    hipLaunchKernelGGL(ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const f16*)X, (long)ldx, (f16*)Y,
                       (long)ldy, rows, C, (const f16*)gamma, (const f16*)beta, eps, (const f16*)shift, (const f16*)scale,
                       (long)ldmod, rows_per_mod, transposed, tr_rows, (long)tr_bstride);
    return iir_launch_status();
}

extern "C" int iir_adaln_batch_f16(const iir_adaln_job* jobs_dev, int32_t njobs, int32_t rows, int32_t max_C, float eps,
                                   int64_t ldmod, int32_t rows_per_mod, int32_t tr_rows, int64_t tr_bstride, void* stream) {
    (void)hipGetLastError();
    if (!jobs_dev || njobs <= 0 || njobs > 65535 || rows <= 0 || rows_per_mod <= 0) return IIR_EINVAL;
    if (max_C <= 0 || max_C > GN_MAXC || max_C % 8) return IIR_EINVAL;
    hipLaunchKernelGGL(adaln_batch_kernel, dim3((rows + 3) / 4, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, rows, eps,
                       (long)ldmod, rows_per_mod, tr_rows, (long)tr_bstride);
    return iir_launch_status();
}
