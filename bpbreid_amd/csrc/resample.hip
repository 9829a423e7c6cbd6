// Layout / resampling kernels of the backbone (all HBM-bound, NHWC fp32, 16-byte accesses):
//   * NCHW image batch -> NHWC with C padded 3 -> 4           (boundary: part_based_engine.py:347-351 hands NCHW)
//   * 3x3 stride-2 max-pool fwd/bwd                            (torchreid/models/resnet.py:217, 346)
//   * bilinear (align_corners=True) upsample of the four HRNet head branches written straight into
//     their channel slice of the 1920-channel feature map, fwd/bwd (torchreid/models/hrnet.py:568-573:
//     F.interpolate x3 + torch.cat become ONE write of the concatenated tensor, no intermediate copies)
#include "bpb_common.h"

__global__ __launch_bounds__(256) void bpb_nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                                int C, int HW)
{
    const long total = (long)N * HW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long n = i / HW, p = i - n * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C && c < 4; ++c) v[c] = x[(n * C + c) * HW + p];
        *(f32x4*)(y + i * 4) = v;
    }
}

// Zero insertion: dst[n][2a][2b][:] (+)= src[n][a][b][:], every other pixel of dst 0 (or untouched when accumulating).  The data
// gradient of a 1x1 stride-2 convolution (resnet.py:119-127 downsample paths) is the 1x1 stride-1 data gradient on dy -- the lean
// kernel, into a compact [N][A][B][C] buffer -- spread over the even pixels of dx by this pass.
__global__ __launch_bounds__(256) void bpb_scatter_stride2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int A, int B,
                                                                  int H, int W, int c4, int accumulate)
{
    if (accumulate) {       // only the even pixels change
        const long total = (long)N * A * B * c4;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
            const int cq = (int)(i % c4);
            long q = i / c4;
            const int b = (int)(q % B);
            q /= B;
            const int a = (int)(q % A);
            const long n = q / A;
            float* o = dst + (((n * H + 2 * a) * W + 2 * b) * (long)c4 + cq) * 4;
            const f32x4 v = *(const f32x4*)(src + i * 4), old = *(const f32x4*)o;
            *(f32x4*)o = v + old;
        }
        return;
    }
    const long total = (long)N * H * W * c4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int cq = (int)(i % c4);
        long q = i / c4;
        const int w = (int)(q % W);
        q /= W;
        const int h = (int)(q % H);
        const long n = q / H;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!((h | w) & 1)) v = *(const f32x4*)(src + (((n * A + (h >> 1)) * B + (w >> 1)) * (long)c4 + cq) * 4);
        *(f32x4*)(dst + i * 4) = v;
    }
}

// NHWC -> NCHW (used for returning pixel-classifier logits and for tests)
__global__ __launch_bounds__(256) void bpb_nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                               int C, int HW)
{
    const long total = (long)N * C * HW;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long p = i % HW;
        const long r = i / HW;
        const int c = (int)(r % C);
        const long n = r / C;
        y[i] = x[(n * HW + p) * C + c];
    }
}

// ---- max pool 3x3, stride 2, pad 1 ------------------------------------------------------------
// forward stores the arg-max tap index (0..8, first maximum wins like ATen) as a byte for the backward.
__global__ __launch_bounds__(256) void bpb_maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   unsigned char* __restrict__ idx, int N, int H, int W,
                                                                   int C, int Ho, int Wo)
{
    const int c4 = C >> 2;
    const long total = (long)N * Ho * Wo * c4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int cq = (int)(i % c4);
        long p = i / c4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho);
        const long n = p / Ho;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int h = ho * 2 - 1 + r, w = wo * 2 - 1 + s;
                if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                    const f32x4 v = *(const f32x4*)(x + ((n * H + h) * W + w) * C + cq * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
                }
            }
        *(f32x4*)(y + i * 4) = best;
        *(uchar4*)(idx + i * 4) = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    }
}

// gather form (deterministic): dx[h][w] = sum of dy[ho][wo] whose arg-max tap points at (h, w)
__global__ __launch_bounds__(256) void bpb_maxpool3x3s2_bwd_kernel(const float* __restrict__ dy,
                                                                   const unsigned char* __restrict__ idx,
                                                                   float* __restrict__ dx, int N, int H, int W, int C,
                                                                   int Ho, int Wo, int accumulate)
{
    const int c4 = C >> 2;
    const long total = (long)N * H * W * c4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int cq = (int)(i % c4);
        long p = i / c4;
        const int w = (int)(p % W);
        p /= W;
        const int h = (int)(p % H);
        const long n = p / H;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // output positions whose window covers (h, w): ho*2-1+r == h  -> ho in {(h+1)/2 .. } with r in 0..2
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hh = h + 1 - r;
            if (hh < 0 || (hh & 1)) continue;
            const int ho = hh >> 1;
            if (ho >= Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ww = w + 1 - s;
                if (ww < 0 || (ww & 1)) continue;
                const int wo = ww >> 1;
                if (wo >= Wo) continue;
                const long o = (((n * Ho + ho) * Wo + wo) * c4 + cq) * 4;
                const uchar4 k = *(const uchar4*)(idx + o);
                const f32x4 d = *(const f32x4*)(dy + o);
                const int tap = r * 3 + s;
                if (k.x == tap) g[0] += d[0];
                if (k.y == tap) g[1] += d[1];
                if (k.z == tap) g[2] += d[2];
                if (k.w == tap) g[3] += d[3];
            }
        }
        float* o = dx + i * 4;
        if (accumulate) {
            const f32x4 old = *(const f32x4*)o;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        *(f32x4*)o = g;
    }
}

// ---- bilinear (align_corners=True) upsample into a channel slice ----------------------------------
// ATen semantics (UpSampleBilinear2d, align_corners): scale = (in-1)/(out-1) (0 if out==1), src = scale*dst,
// i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1;
// out = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11).     scale == 1 degenerates to a copy.

__global__ __launch_bounds__(256) void bpb_bilinear_concat_fwd_kernel(BpbBilinearArgs A)
{
    const int c4 = A.Cs >> 2;
    const long total = (long)A.N * A.H * A.W * c4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int cq = (int)(i % c4);
        long p = i / c4;
        const int w = (int)(p % A.W);
        p /= A.W;
        const int h = (int)(p % A.H);
        const long n = p / A.H;
        const float fh = A.sh * h, fw = A.sw * w;
        const int h0 = (int)fh, w0 = (int)fw;
        const int h1 = h0 + (h0 < A.Hs - 1), w1 = w0 + (w0 < A.Ws - 1);
        const float lh1 = fh - h0, lw1 = fw - w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        const float* b = A.src + n * A.Hs * A.Ws * A.Cs + cq * 4;
        const f32x4 v00 = *(const f32x4*)(b + ((long)h0 * A.Ws + w0) * A.Cs);
        const f32x4 v01 = *(const f32x4*)(b + ((long)h0 * A.Ws + w1) * A.Cs);
        const f32x4 v10 = *(const f32x4*)(b + ((long)h1 * A.Ws + w0) * A.Cs);
        const f32x4 v11 = *(const f32x4*)(b + ((long)h1 * A.Ws + w1) * A.Cs);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lh0 * (lw0 * v00[e] + lw1 * v01[e]) + lh1 * (lw0 * v10[e] + lw1 * v11[e]);
        *(f32x4*)(A.dst + ((n * A.H + h) * A.W + w) * A.Ct + A.c0 + cq * 4) = o;
    }
}

// backward, gather form (deterministic, no atomics): for each source pixel, visit the destination pixels
// whose 2x2 footprint contains it and recompute their weights exactly as the forward does.
// here A.dst is the gradient of the concatenated map (read) and A.src is written (dsrc).
__global__ __launch_bounds__(256) void bpb_bilinear_concat_bwd_kernel(BpbBilinearArgs A, float* __restrict__ dsrc)
{
    const int c4 = A.Cs >> 2;
    const long total = (long)A.N * A.Hs * A.Ws * c4;
    const float inv_sh = A.sh > 0.f ? 1.f / A.sh : 0.f, inv_sw = A.sw > 0.f ? 1.f / A.sw : 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int cq = (int)(i % c4);
        long p = i / c4;
        const int ws = (int)(p % A.Ws);
        p /= A.Ws;
        const int hs = (int)(p % A.Hs);
        const long n = p / A.Hs;
        // conservative destination ranges: rows h with (int)(sh*h) in {hs-1, hs}
        int hlo = 0, hhi = A.H - 1, wlo = 0, whi = A.W - 1;
        if (A.sh > 0.f) {
            hlo = max(0, (int)floorf((hs - 1) * inv_sh) - 1);
            hhi = min(A.H - 1, (int)ceilf((hs + 1) * inv_sh) + 1);
        }
        if (A.sw > 0.f) {
            wlo = max(0, (int)floorf((ws - 1) * inv_sw) - 1);
            whi = min(A.W - 1, (int)ceilf((ws + 1) * inv_sw) + 1);
        }
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int h = hlo; h <= hhi; ++h) {
            const float fh = A.sh * h;
            const int h0 = (int)fh, h1 = h0 + (h0 < A.Hs - 1);
            const float lh1 = fh - h0, lh0 = 1.f - lh1;
            float wh = 0.f;
            if (h0 == hs) wh += lh0;
            if (h1 == hs) wh += lh1;     // h0 == h1 at the border: both weights land on the same pixel
            if (h0 != hs && h1 != hs) continue;
            for (int w = wlo; w <= whi; ++w) {
                const float fw = A.sw * w;
                const int w0 = (int)fw, w1 = w0 + (w0 < A.Ws - 1);
                if (w0 != ws && w1 != ws) continue;
                const float lw1 = fw - w0, lw0 = 1.f - lw1;
                float ww = 0.f;
                if (w0 == ws) ww += lw0;
                if (w1 == ws) ww += lw1;
                const f32x4 d = *(const f32x4*)(A.dst + ((n * A.H + h) * A.W + w) * A.Ct + A.c0 + cq * 4);
                const float k = wh * ww;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += k * d[e];
            }
        }
        float* o = dsrc + i * 4;
        if (A.accumulate) {
            const f32x4 old = *(const f32x4*)o;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        *(f32x4*)o = g;
    }
}


// ---- the whole concatenated map in ONE launch ------------------------------------------------------------------------
// All sources of the concatenation in one kernel: a thread owns a channel quad of the CONCATENATED map (its source follows
// from the quad), a block walks the pixel groups blk, blk + grid, ... (group = rows * 4 pixels), so every pixel row of the
// output (Ct * 4 bytes, 1920 B for HRNet-W32) is written as one contiguous run -- four launches that each write a 128..1024 B
// slice of every row ran at 0.7 TB/s.  Optionally (training, partials != nullptr) the block also emits the per-channel
// (sum, sum of squares) partials of what it wrote: the batch statistics of the pixel classifier's BatchNorm2d
// (bpbreid.py:379,384) without re-reading the 1 GB map (partials [gridDim.x][2][Ct], fp64).
__device__ __forceinline__ f32x4 bpb_bilinear_sample(const BpbBilinearArgs& A, const float* __restrict__ b, int h, int w)
{
    if (A.Hs == A.H && A.Ws == A.W) return *(const f32x4*)(b + ((long)h * A.Ws + w) * A.Cs);
    const float fh = A.sh * h, fw = A.sw * w;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < A.Hs - 1), w1 = w0 + (w0 < A.Ws - 1);
    const float lh1 = fh - h0, lw1 = fw - w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const f32x4 v00 = *(const f32x4*)(b + ((long)h0 * A.Ws + w0) * A.Cs);
    const f32x4 v01 = *(const f32x4*)(b + ((long)h0 * A.Ws + w1) * A.Cs);
    const f32x4 v10 = *(const f32x4*)(b + ((long)h1 * A.Ws + w0) * A.Cs);
    const f32x4 v11 = *(const f32x4*)(b + ((long)h1 * A.Ws + w1) * A.Cs);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = lh0 * (lw0 * v00[e] + lw1 * v01[e]) + lh1 * (lw0 * v10[e] + lw1 * v11[e]);
    return o;
}

__global__ __launch_bounds__(256) void bpb_bilinear_concat_multi_fwd_kernel(const BpbBilinearArgs* __restrict__ descs, int nsrc,
                                                                            double* __restrict__ partials, float* __restrict__ dst_override)
{
    __shared__ double red[256 * 8];
    constexpr int U = 4;
    const int Ct = descs[0].Ct, H = descs[0].H, W = descs[0].W;
    const long P = (long)descs[0].N * H * W;
    const int c4 = Ct >> 2;
    const int tx = c4 >= 256 ? 256 : c4, rows = 256 / tx;
    const int tcq = threadIdx.x % tx, trow = threadIdx.x / tx;
    const long gsz = (long)U * rows;
    const long ngroups = (P + gsz - 1) / gsz;
    for (int cq = tcq; cq < c4; cq += tx) {
        int si = 0;
        for (int i = 1; i < nsrc; ++i)
            if (cq * 4 >= descs[i].c0) si = i;
        const BpbBilinearArgs A = descs[si];
        float* const dst = dst_override ? dst_override : A.dst;
        const int lc = cq * 4 - A.c0;
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        if (trow < rows) {
            for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
                long p = g * gsz + trow;
                int w = (int)(p % W);
                long t = p / W;
                int h = (int)(t % H);
                long n = t / H;
                float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};   // fp32 over the group's U pixels, then fp64
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (p < P) {
                        const f32x4 o = bpb_bilinear_sample(A, A.src + n * A.Hs * A.Ws * A.Cs + lc, h, w);
                        *(f32x4*)(dst + p * Ct + cq * 4) = o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            fs[e] += o[e];
                            fq[e] += o[e] * o[e];
                        }
                    }
                    p += rows;
                    w += rows;
                    while (w >= W) {
                        w -= W;
                        if (++h == H) {
                            h = 0;
                            ++n;
                        }
                    }
                }
                if (partials) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] += (double)fs[e];
                        q[e] += (double)fq[e];
                    }
                }
            }
        }
        if (partials) {            // (uniform) combine the pixel rows in a fixed order, as bpb_channel_stats does
            if (rows > 1) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[threadIdx.x * 8 + e] = s[e];
                    red[threadIdx.x * 8 + 4 + e] = q[e];
                }
                __syncthreads();
                if (trow == 0)
                    for (int r = 1; r < rows; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s[e] += red[(r * tx + tcq) * 8 + e];
                            q[e] += red[(r * tx + tcq) * 8 + 4 + e];
                        }
            }
            if (trow == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    partials[((size_t)blockIdx.x * 2 + 0) * Ct + cq * 4 + e] = s[e];
                    partials[((size_t)blockIdx.x * 2 + 1) * Ct + cq * 4 + e] = q[e];
                }
            }
        }
    }
}

// Backward, separable and in gather form (deterministic, no atomics).  The 2-D tent weights factor: wh(h -> hs) * ww(w -> ws), so
//   pass W:  tmp[n][h][ws][c]   = sum_w ww(w, ws) * dcat[n][h][w][c0 + c]      (every element of dcat read once; a source at
//                                  the output resolution is copied straight to dsrc here)
//   pass H:  dsrc[n][hs][ws][c] = sum_h wh(h, hs) * tmp[n][h][ws][c]
// The one-pass gather visited (2f - 1)^2 destination pixels per source pixel (f = 2, 4, 8): 3.6 GB of fetches for a 252 MB map.
__device__ __forceinline__ void bpb_tent_range(int is, int nin, int nout, float scale, int& lo, int& hi)
{
    lo = 0;
    hi = nout - 1;
    if (scale > 0.f) {     // conservative: outputs o with (int)(scale * o) in {is - 1, is}
        const float inv = 1.f / scale;
        lo = max(0, (int)floorf((is - 1) * inv) - 1);
        hi = min(nout - 1, (int)ceilf((is + 1) * inv) + 1);
    }
}

__device__ __forceinline__ float bpb_tent_weight(int o, int is, int nin, float scale)
{
    const float f = scale * o;
    const int i0 = (int)f, i1 = i0 + (i0 < nin - 1);
    const float l1 = f - i0, l0 = 1.f - l1;
    float wgt = 0.f;
    if (i0 == is) wgt += l0;
    if (i1 == is) wgt += l1;      // i0 == i1 at the border: both weights land on the same pixel
    return wgt;
}

__global__ __launch_bounds__(256) void bpb_bilinear_concat_multi_bwd_w_kernel(const BpbBilinearBwdDesc* __restrict__ descs, int nsrc)
{
    int di = 0;
    for (int i = 1; i < nsrc; ++i)
        if ((int)blockIdx.x >= descs[i].blk_begin_w) di = i;
    const BpbBilinearBwdDesc A = descs[di];
    const int c4 = A.Cs >> 2;
    const bool same = A.Hs == A.H && A.Ws == A.W;
    const long total = (long)A.N * A.H * A.Ws * c4;
    const long i = (long)((int)blockIdx.x - A.blk_begin_w) * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % c4);
    long t = i / c4;
    const int ws = (int)(t % A.Ws);
    const long nh = t / A.Ws;                      // n * H + h
    const float* row = A.dcat + nh * A.W * A.Ct + A.c0 + cq * 4;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (A.Ws == A.W) {
        g = *(const f32x4*)(row + (long)ws * A.Ct);
    } else {
        int wlo, whi;
        bpb_tent_range(ws, A.Ws, A.W, A.sw, wlo, whi);
        for (int w = wlo; w <= whi; ++w) {
            const float ww = bpb_tent_weight(w, ws, A.Ws, A.sw);
            if (ww == 0.f) continue;
            const f32x4 d = *(const f32x4*)(row + (long)w * A.Ct);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += ww * d[e];
        }
    }
    float* o = (same ? A.dsrc : A.tmp) + i * 4;    // same resolution: tmp layout == dsrc layout, no pass H
    if (same && A.accumulate) {
        const f32x4 old = *(const f32x4*)o;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += old[e];
    }
    *(f32x4*)o = g;
}

__global__ __launch_bounds__(256) void bpb_bilinear_concat_multi_bwd_h_kernel(const BpbBilinearBwdDesc* __restrict__ descs, int nsrc)
{
    int di = 0;
    for (int i = 1; i < nsrc; ++i)
        if ((int)blockIdx.x >= descs[i].blk_begin_h) di = i;
    const BpbBilinearBwdDesc A = descs[di];
    const int c4 = A.Cs >> 2;
    const long total = (long)A.N * A.Hs * A.Ws * c4;
    const long i = (long)((int)blockIdx.x - A.blk_begin_h) * 256 + threadIdx.x;
    if (i >= total || (A.Hs == A.H && A.Ws == A.W)) return;
    const long rowq = (long)A.Ws * c4;             // float4s per row of tmp / dsrc
    const long col = i % rowq;
    const long t = i / rowq;
    const int hs = (int)(t % A.Hs);
    const long n = t / A.Hs;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (A.Hs == A.H) {
        g = *(const f32x4*)(A.tmp + ((n * A.H + hs) * rowq + col) * 4);
    } else {
        int hlo, hhi;
        bpb_tent_range(hs, A.Hs, A.H, A.sh, hlo, hhi);
        for (int h = hlo; h <= hhi; ++h) {
            const float wh = bpb_tent_weight(h, hs, A.Hs, A.sh);
            if (wh == 0.f) continue;
            const f32x4 d = *(const f32x4*)(A.tmp + ((n * A.H + h) * rowq + col) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += wh * d[e];
        }
    }
    float* o = A.dsrc + i * 4;
    if (A.accumulate) {
        const f32x4 old = *(const f32x4*)o;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += old[e];
    }
    *(f32x4*)o = g;
}

static int ew_grid(long total_vec)
{
    long g = (total_vec + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int bpb_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, hipStream_t stream)
{
    BPB_REQUIRE(C >= 1 && C <= 4, "bpb_nchw_to_nhwc4: C=%d", C);
    hipLaunchKernelGGL(bpb_nchw_to_nhwc4_kernel, dim3(ew_grid((long)N * H * W)), dim3(256), 0, stream, x, y, N, C, H * W);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_scatter_stride2(const float* src, float* dst, int N, int A, int B, int H, int W, int C, int accumulate, hipStream_t stream)
{
    BPB_REQUIRE(C % 4 == 0 && N >= 1 && A == (H + 1) / 2 && B == (W + 1) / 2, "bpb_scatter_stride2: C=%d, %dx%d -> %dx%d", C, A, B, H, W);
    const long total = accumulate ? (long)N * A * B * (C / 4) : (long)N * H * W * (C / 4);
    hipLaunchKernelGGL(bpb_scatter_stride2_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, src, dst, N, A, B, H, W, C / 4, accumulate);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, hipStream_t stream)
{
    hipLaunchKernelGGL(bpb_nhwc_to_nchw_kernel, dim3(ew_grid((long)N * C * H * W)), dim3(256), 0, stream, x, y, N, C, H * W);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, hipStream_t stream)
{
    BPB_REQUIRE(C % 4 == 0, "bpb_maxpool: C must be a multiple of 4");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(bpb_maxpool3x3s2_fwd_kernel, dim3(ew_grid((long)N * Ho * Wo * (C / 4))), dim3(256), 0, stream, x, y,
                       idx, N, H, W, C, Ho, Wo);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C, int accumulate,
                         hipStream_t stream)
{
    BPB_REQUIRE(C % 4 == 0, "bpb_maxpool: C must be a multiple of 4");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(bpb_maxpool3x3s2_bwd_kernel, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(256), 0, stream, dy, idx,
                       dx, N, H, W, C, Ho, Wo, accumulate);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bilinear_concat_fwd(const BpbBilinearArgs* a, hipStream_t stream)
{
    BPB_REQUIRE(a->Cs % 4 == 0 && a->Ct % 4 == 0 && a->c0 % 4 == 0, "bpb_bilinear_concat: channels must be multiples of 4");
    hipLaunchKernelGGL(bpb_bilinear_concat_fwd_kernel, dim3(ew_grid((long)a->N * a->H * a->W * (a->Cs / 4))), dim3(256), 0,
                       stream, *a);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bilinear_concat_bwd(const BpbBilinearArgs* a, float* dsrc, hipStream_t stream)
{
    BPB_REQUIRE(a->Cs % 4 == 0 && a->Ct % 4 == 0 && a->c0 % 4 == 0, "bpb_bilinear_concat: channels must be multiples of 4");
    hipLaunchKernelGGL(bpb_bilinear_concat_bwd_kernel, dim3(ew_grid((long)a->N * a->Hs * a->Ws * (a->Cs / 4))), dim3(256),
                       0, stream, *a, dsrc);
    BPB_LAUNCH_OK();
    return 0;
}

// All sources of one concatenation (<= 8, channel slices in ascending order covering [0, Ct)) in one launch.
// partials (optional): nblocks * 2 * Ct doubles <- per-block (sum, sum of squares) of the written map, per channel.
// dst_override (optional): write the map there instead of to the descriptors' dst (same shape) -- the eval forward hands every
// call a fresh output tensor without re-uploading the descriptors.
int bpb_bilinear_concat_multi_fwd(const BpbBilinearArgs* d_descs, const BpbBilinearArgs* h_descs, int n, double* partials,
                                  int nblocks, float* dst_override, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 8 && nblocks >= 1, "bpb_bilinear_concat_multi_fwd: n=%d nblocks=%d", n, nblocks);
    int c0 = 0;
    for (int i = 0; i < n; ++i) {
        const BpbBilinearArgs* a = &h_descs[i];
        BPB_REQUIRE(a->Cs % 4 == 0 && a->Ct % 4 == 0 && a->c0 == c0, "bpb_bilinear_concat_multi_fwd: source %d: slices must tile the map", i);
        BPB_REQUIRE(a->N == h_descs[0].N && a->H == h_descs[0].H && a->W == h_descs[0].W && a->Ct == h_descs[0].Ct &&
                        a->dst == h_descs[0].dst, "bpb_bilinear_concat_multi_fwd: source %d targets another map", i);
        c0 += a->Cs;
    }
    BPB_REQUIRE(c0 == h_descs[0].Ct, "bpb_bilinear_concat_multi_fwd: the slices cover %d of %d channels", c0, h_descs[0].Ct);
    hipLaunchKernelGGL(bpb_bilinear_concat_multi_fwd_kernel, dim3(nblocks), dim3(256), 0, stream, d_descs, n, partials, dst_override);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bilinear_concat_multi_bwd(const BpbBilinearBwdDesc* d_descs, const BpbBilinearBwdDesc* h_descs, int n, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 8, "bpb_bilinear_concat_multi_bwd: n=%d", n);
    int bw = 0, bh = 0;
    bool any_h = false;
    for (int i = 0; i < n; ++i) {
        const BpbBilinearBwdDesc* a = &h_descs[i];
        const bool same = a->Hs == a->H && a->Ws == a->W;
        BPB_REQUIRE(a->Cs % 4 == 0 && a->Ct % 4 == 0 && a->c0 % 4 == 0 && (same || a->tmp != nullptr),
                    "bpb_bilinear_concat_multi_bwd: source %d", i);
        BPB_REQUIRE(a->blk_begin_w == bw && a->blk_begin_h == bh, "bpb_bilinear_concat_multi_bwd: block prefix of source %d", i);
        bw += bpb_cdiv((long)a->N * a->H * a->Ws * (a->Cs / 4), 256);
        bh += bpb_cdiv((long)a->N * a->Hs * a->Ws * (a->Cs / 4), 256);
        any_h = any_h || !same;
    }
    hipLaunchKernelGGL(bpb_bilinear_concat_multi_bwd_w_kernel, dim3(bw), dim3(256), 0, stream, d_descs, n);
    if (any_h) hipLaunchKernelGGL(bpb_bilinear_concat_multi_bwd_h_kernel, dim3(bh), dim3(256), 0, stream, d_descs, n);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
