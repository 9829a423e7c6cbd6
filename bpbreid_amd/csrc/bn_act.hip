// BatchNorm2d (training + eval) fused with residual/fuse sums, nearest upsampling and ReLU.
//
// Every post-convolution elementwise stage of HRNet / ResNet on the path has the form
//     out = act( sum_t  affine_t( nearest_up_t( src_t ) ) ),   t < 4
// with affine_t = BatchNorm (batch or running statistics) or identity, act = ReLU or identity:
//   BasicBlock / Bottleneck tails      torchreid/models/hrnet.py:80-96, 117-137; resnet.py:133-154
//   transition / stem conv-BN-ReLU     hrnet.py:533-538, 458-481
//   HighResolutionModule fuse sums     hrnet.py:269-277 (+ nn.Upsample nearest, :231)
// One HBM-bound pass reads each operand once and writes `out` once (the reference does one pass
// per BN, per add, per upsample and per ReLU).  BatchNorm batch statistics come from per-tile fp64
// partials emitted by the conv epilogue (conv_igemm.hip) -> bpb_bn_finalize -> (scale, shift).
#include "bpb_common.h"


// ---- (1) batch statistics -> affine, running stats (nn.BatchNorm2d training semantics) ----------
// partials: [nparts][2][C] doubles (sum, sumsq).  count = elements per channel.
// Outputs: scale = gamma*invstd, shift = beta - mean*scale, mean, invstd (saved for backward);
// running_mean = (1-m)*running_mean + m*mean; running_var uses the unbiased variance (torch semantics).
// Column sums of the partial rows for BPB_FIN_CH channels per workgroup: 1024 threads = BPB_FIN_CH channels x 128 row lanes.
// Lane r adds rows r, r+128, ... (4 independent loads in flight: the rows are L2-resident, the loop is latency-bound), then
// the 128 lane sums are combined in a FIXED order through LDS (16 groups of 8, then the 16 group sums) -> deterministic.
// A 32-channel x 32-lane layout took ~10 us per BatchNorm (64 dependent L2 round trips); this one ~3-4 us.
#define BPB_FIN_LANES (1024 / BPB_FIN_CH)
template <int FIN_CH_T = BPB_FIN_CH>
__device__ __forceinline__ bool bpb_fin_column_sums(int blk, double (*red)[BPB_FIN_LANES][FIN_CH_T], const double* __restrict__ partials,
                                                    int nparts, int C, double& s_out, double& q_out)
{
    constexpr int BPB_FIN_CH_L = FIN_CH_T;
    const int cl = threadIdx.x % BPB_FIN_CH_L, rg = threadIdx.x / BPB_FIN_CH_L;
    const int c = blk * BPB_FIN_CH_L + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int p = rg;
        for (; p + 3 * BPB_FIN_LANES < nparts; p += 4 * BPB_FIN_LANES) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = partials[((size_t)(p + u * BPB_FIN_LANES) * 2 + 0) * C + c];
                b[u] = partials[((size_t)(p + u * BPB_FIN_LANES) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s += a[u]; q += b[u]; }
        }
        for (; p < nparts; p += BPB_FIN_LANES) {
            s += partials[((size_t)p * 2 + 0) * C + c];
            q += partials[((size_t)p * 2 + 1) * C + c];
        }
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg < 16) {
        s = 0.0;
        q = 0.0;
#pragma unroll
        for (int i = 0; i < BPB_FIN_LANES / 16; ++i) {
            s += red[0][rg * (BPB_FIN_LANES / 16) + i][cl];
            q += red[1][rg * (BPB_FIN_LANES / 16) + i][cl];
        }
    }
    __syncthreads();
    if (rg < 16) {
        red[0][rg][cl] = s;
        red[1][rg][cl] = q;
    }
    __syncthreads();
    if (rg != 0 || c >= C) return false;
    s = 0.0;
    q = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        s += red[0][i][cl];
        q += red[1][i][cl];
    }
    s_out = s;
    q_out = q;
    return true;
}

template <int FIN_CH_T = BPB_FIN_CH>
__device__ __forceinline__ void bpb_bn_finalize_body(int blk, double (*red)[BPB_FIN_LANES][FIN_CH_T], const double* __restrict__ partials,
                                                     int nparts, int C, double count, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float momentum,
                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                     float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var)
{
    double s, q;
    if (!bpb_fin_column_sums<FIN_CH_T>(blk, red, partials, nparts, C, s, q)) return;
    const int c = blk * FIN_CH_T + threadIdx.x % FIN_CH_T;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ __launch_bounds__(1024) void bpb_bn_finalize_kernel(const double* __restrict__ partials, int nparts, int C,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ scale, float* __restrict__ shift,
                                                              float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                              float* __restrict__ running_mean,
                                                              float* __restrict__ running_var)
{
    __shared__ double red[2][BPB_FIN_LANES][BPB_FIN_CH];
    bpb_bn_finalize_body(blockIdx.x, red, partials, nparts, C, count, gamma, beta, eps, momentum, scale, shift, mean_out, invstd_out,
                         running_mean, running_var);
}

__global__ __launch_bounds__(1024) void bpb_bn_finalize_multi_kernel(const BpbBnFinDesc* __restrict__ descs, BpbBlkBegins bb)
{
    __shared__ double red[2][BPB_FIN_LANES][BPB_FIN_CH];
    const int di = bpb_find_problem(bb, (int)blockIdx.x);
    const BpbBnFinDesc D = descs[di];
    bpb_bn_finalize_body(blockIdx.x - D.blk_begin, red, D.partials, D.nparts, D.C, D.count, D.gamma, D.beta, D.eps, D.momentum, D.scale,
                         D.shift, D.mean, D.invstd, D.running_mean, D.running_var);
}

// eval mode, every BatchNorm of the network in one launch (descriptor table, 256 channels per block)
__global__ __launch_bounds__(256) void bpb_bn_eval_affine_batched_kernel(const BpbBnEvalDesc* __restrict__ descs, int n, float eps)
{
    int bid = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].blk_begin <= bid) lo = mid; else hi = mid - 1;
    }
    const BpbBnEvalDesc D = descs[lo];
    const int c = (bid - D.blk_begin) * 256 + threadIdx.x;
    if (c < D.C) {
        const float invstd = 1.f / sqrtf(D.running_var[c] + eps);
        const float g = D.gamma ? D.gamma[c] : 1.f, b = D.beta ? D.beta[c] : 0.f;
        const float sc = g * invstd;
        D.scale[c] = sc;
        D.shift[c] = b - D.running_mean[c] * sc;
    }
}

// eval mode: affine from running statistics
__global__ void bpb_bn_eval_affine_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                          float eps, float* __restrict__ scale, float* __restrict__ shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float invstd = 1.f / sqrtf(running_var[c] + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * invstd;
        shift[c] = b - running_mean[c] * scale[c];
    }
}

// per-channel (sum, sumsq) partials of an NHWC tensor: used where no conv epilogue produced them
// (pixel-classifier BN over the concatenated feature map, bpbreid.py:379,384).
__global__ __launch_bounds__(256) void bpb_channel_stats_kernel(const float* __restrict__ x, long P, int C,
                                                                double* __restrict__ partials)
{
    // the 256 threads are tx channel quads x rows pixel rows, each with 8 independent 16-byte loads in flight; every element
    // is accumulated in fp64 (4.3 TB/s on the 1 GB HRNet-W32 map).
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int U = 8;
    const int c4 = C >> 2;
    const int tx = c4 >= 256 ? 256 : c4;         // threads along channels
    const int rows = 256 / tx;                   // pixel rows per sweep; threads with trow >= rows idle
    const int tcq = threadIdx.x % tx, trow = threadIdx.x / tx;
    double* red = (double*)smem_f;               // [thread][8]
    // Block b takes the pixel groups b, b + grid, ... (group = U * rows pixels): the chip sweeps one contiguous window.
    const long gsz = (long)U * rows;
    const long nfull = P / gsz;                  // whole groups; the ragged tail goes to block 0 below
    for (int cq = tcq; cq < c4; cq += tx) {
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        if (trow < rows) {
            for (long g = blockIdx.x; g < nfull; g += gridDim.x) {
                const long p = g * gsz + trow;
                f32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *(const f32x4*)(x + (p + (long)u * rows) * C + cq * 4);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] += (double)v[u][e];
                        q[e] += (double)v[u][e] * (double)v[u][e];
                    }
            }
            for (long p = nfull * gsz + trow; blockIdx.x == 0 && p < P; p += rows) {
                const f32x4 v = *(const f32x4*)(x + p * C + cq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += (double)v[e];
                    q[e] += (double)v[e] * (double)v[e];
                }
            }
        }
        if (rows > 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[(threadIdx.x) * 8 + e] = s[e];
                red[(threadIdx.x) * 8 + 4 + e] = q[e];
            }
            __syncthreads();
            if (trow == 0) {
                for (int r = 1; r < rows; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] += red[(r * tx + tcq) * 8 + e];
                        q[e] += red[(r * tx + tcq) * 8 + 4 + e];
                    }
            }
        }
        if (trow == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                partials[((size_t)blockIdx.x * 2 + 0) * C + cq * 4 + e] = s[e];
                partials[((size_t)blockIdx.x * 2 + 1) * C + cq * 4 + e] = q[e];
            }
        }
    }
}

// ---- (2) fused forward: out = act(sum_t affine_t(up_t(src_t))) -----------------------------------

__device__ __forceinline__ unsigned bpb_fdiv2(unsigned x, unsigned d, unsigned magic)
{
    return d == 1 ? x : __umulhi(x, magic);
}

__device__ __forceinline__ void bpb_fuse_fwd_body(const BpbFuseArgs& A, int blk, int nblk)
{
    const int c4 = A.C >> 2;
    const long total = (long)A.N * A.H * A.W * c4;
    bool any_up = false;
#pragma unroll
    for (int t = 0; t < BPB_MAX_TERMS; ++t)
        if (t < A.nterms && A.up[t] > 0) any_up = true;
    for (long i = blk * 256L + threadIdx.x; i < total; i += nblk * 256L) {
        const int cq = (int)(i % c4);
        const long p = i / c4;
        int n = 0, h = 0, w = 0;
        if (any_up) {
            const unsigned pw = bpb_fdiv2((unsigned)p, A.W, A.magic_w);
            w = (int)((unsigned)p - pw * A.W);
            const unsigned ph = bpb_fdiv2(pw, A.H, A.magic_h);
            h = (int)(pw - ph * A.H);
            n = (int)ph;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < BPB_MAX_TERMS; ++t) {
            if (t < A.nterms) {
                long sp = p;
                if (A.up[t] > 0) sp = ((long)n * (A.H >> A.up[t]) + (h >> A.up[t])) * (A.W >> A.up[t]) + (w >> A.up[t]);
                f32x4 v = *(const f32x4*)(A.src[t] + sp * A.C + cq * 4);
                if (A.scale[t]) {
                    const f32x4 sc = *(const f32x4*)(A.scale[t] + cq * 4);
                    const f32x4 sh = *(const f32x4*)(A.shift[t] + cq * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += v[e];
            }
        }
        if (A.relu) {
            if (A.maskbits) {
                // ReLU mask of this wave's 64 float4s as four 64-bit words (one per component); i of lane 0 is a multiple of 64
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned long long b = __ballot(acc[e] > 0.f);
                    if ((threadIdx.x & 63) == 0) A.maskbits[(i >> 6) * 4 + e] = b;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = acc[e] > 0.f ? acc[e] : 0.f;
        }
        *(f32x4*)(A.out + p * A.C + cq * 4) = acc;
    }
}

__global__ __launch_bounds__(256) void bpb_fuse_fwd_kernel(BpbFuseArgs A)
{
    bpb_fuse_fwd_body(A, blockIdx.x, gridDim.x);
}

// grouped launches: block -> record through the blk_begin prefix held in the kernel arguments (bpb_common.h)
__global__ __launch_bounds__(256) void bpb_fuse_fwd_multi_kernel(const BpbFuseArgs* __restrict__ descs, BpbBlkBegins bb)
{
    const BpbFuseArgs A = descs[bpb_find_problem(bb, (int)blockIdx.x)];
    bpb_fuse_fwd_body(A, blockIdx.x - A.blk_begin, A.nblk);
}

// ---- (3) backward of one term ------------------------------------------------------------------
// G[q][c] = sum over the 2^up x 2^up window of dout * (out > 0 if relu).

// ReLU mask of float4 number idx of the fuse output: from the bit array written by the forward pass (32x fewer bytes than the
// output tensor itself, identical mask), else from the output
typedef unsigned long long bpb_u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bpb_apply_relu_mask(const BpbTermBwdArgs& A, long idx, f32x4& g)
{
    if (A.maskbits) {
        const bpb_u64x2* w = (const bpb_u64x2*)(A.maskbits + (idx >> 6) * 4);
        const bpb_u64x2 lo = w[0], hi = w[1];
        const int b = (int)(idx & 63);
        g[0] = ((lo[0] >> b) & 1ull) ? g[0] : 0.f;
        g[1] = ((lo[1] >> b) & 1ull) ? g[1] : 0.f;
        g[2] = ((hi[0] >> b) & 1ull) ? g[2] : 0.f;
        g[3] = ((hi[1] >> b) & 1ull) ? g[3] : 0.f;
    } else {
        const f32x4 o = *(const f32x4*)(A.out + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
}

__device__ __forceinline__ f32x4 bpb_window_grad(const BpbTermBwdArgs& A, long q, int cq)
{
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int c4 = A.C >> 2;
    if (A.up == 0) {
        g = *(const f32x4*)(A.dout + q * A.C + cq * 4);
        if (A.relu) bpb_apply_relu_mask(A, q * c4 + cq, g);
        return g;
    }
    const unsigned pw = bpb_fdiv2((unsigned)q, A.Ws, A.magic_w);
    const int w = (int)((unsigned)q - pw * A.Ws);
    const unsigned ph = bpb_fdiv2(pw, A.Hs, A.magic_h);
    const int h = (int)(pw - ph * A.Hs), n = (int)ph;
    const int f = 1 << A.up, H = A.Hs << A.up, W = A.Ws << A.up;
    for (int dh = 0; dh < f; ++dh)
        for (int dw = 0; dw < f; ++dw) {
            const long p = ((long)n * H + (h * f + dh)) * W + (w * f + dw);
            f32x4 v = *(const f32x4*)(A.dout + p * A.C + cq * 4);
            if (A.relu) bpb_apply_relu_mask(A, p * c4 + cq, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += v[e];
        }
    return g;
}

// identity term: dsrc (+)= G
__device__ __forceinline__ void bpb_term_bwd_identity_body(const BpbTermBwdArgs& A, int blk, int nblk)
{
    const int c4 = A.C >> 2;
    const long total = (long)A.N * A.Hs * A.Ws * c4;
    for (long i = blk * 256L + threadIdx.x; i < total; i += nblk * 256L) {
        const int cq = (int)(i % c4);
        const long q = i / c4;
        f32x4 g = bpb_window_grad(A, q, cq);
        float* d = A.dsrc + q * A.C + cq * 4;
        if (A.accumulate) {
            const f32x4 old = *(const f32x4*)d;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        *(f32x4*)d = g;
    }
}

__global__ __launch_bounds__(256) void bpb_term_bwd_identity_kernel(BpbTermBwdArgs A)
{
    bpb_term_bwd_identity_body(A, blockIdx.x, gridDim.x);
}

// BN term, pass 1: per-block partials of (sum G, sum G*xhat) per channel
__device__ __forceinline__ void bpb_term_bwd_bn_reduce_body(const BpbTermBwdArgs& A, int blk, int nblk, double* red)
{
    const int c4 = A.C >> 2;
    const long P = (long)A.N * A.Hs * A.Ws;
    const long ppb = (P + nblk - 1) / nblk;
    const long p0 = blk * ppb, p1 = min(P, p0 + ppb);
    const int tx = c4 >= 256 ? 256 : c4;
    const int rows = 256 / tx;             // threads with trow >= rows idle (c4 need not be a power of two)
    const int tcq = threadIdx.x % tx, trow = threadIdx.x / tx;
    for (int cq = tcq; cq < c4; cq += tx) {
        const f32x4 mu = *(const f32x4*)(A.mean + cq * 4);
        const f32x4 is = *(const f32x4*)(A.invstd + cq * 4);
        float s[4] = {0, 0, 0, 0}, sx[4] = {0, 0, 0, 0};
        double ds[4] = {0, 0, 0, 0}, dsx[4] = {0, 0, 0, 0};
        int cnt = 0;
        long q = p0 + trow;
        if (trow < rows) {
            // four pixels per round: their 8..12 loads are in flight together (one pixel per round left the pass at 3.9 TB/s
            // against 5.5 for the apply pass); fp32 running sums are flushed into fp64 every 64 pixels
            for (; q + 3L * rows < p1; q += 4L * rows) {
                f32x4 g[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    g[u] = bpb_window_grad(A, q + (long)u * rows, cq);
                    x[u] = *(const f32x4*)(A.src + (q + (long)u * rows) * A.C + cq * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s[e] += g[u][e];
                        sx[e] += g[u][e] * ((x[u][e] - mu[e]) * is[e]);
                    }
                if (++cnt == 16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ds[e] += s[e]; dsx[e] += sx[e]; s[e] = 0.f; sx[e] = 0.f; }
                    cnt = 0;
                }
            }
            for (; q < p1; q += rows) {
                const f32x4 g = bpb_window_grad(A, q, cq);
                const f32x4 x = *(const f32x4*)(A.src + q * A.C + cq * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += g[e];
                    sx[e] += g[e] * ((x[e] - mu[e]) * is[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { ds[e] += s[e]; dsx[e] += sx[e]; }
        if (rows > 1) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[threadIdx.x * 8 + e] = ds[e];
                red[threadIdx.x * 8 + 4 + e] = dsx[e];
            }
            __syncthreads();
            if (trow == 0)
                for (int r = 1; r < rows; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ds[e] += red[(r * tx + tcq) * 8 + e];
                        dsx[e] += red[(r * tx + tcq) * 8 + 4 + e];
                    }
        }
        if (trow == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // write-through (sc1) stores: see the hand-off at the end of the kernel
                __hip_atomic_store(&A.partials[((size_t)blk * 2 + 0) * A.C + cq * 4 + e], ds[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&A.partials[((size_t)blk * 2 + 1) * A.C + cq * 4 + e], dsx[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bpb_term_bwd_bn_reduce_kernel(BpbTermBwdArgs A)
{
    __shared__ double red[256 * 8];
    bpb_term_bwd_bn_reduce_body(A, blockIdx.x, gridDim.x, red);
    // ---- fused finalisation (see bpb_conv_igemm_kernel): the last workgroup to finish turns the partials into
    // dgamma / dbeta and the per-channel constants c1, c2 of the apply pass -- no separate 2-8 workgroup launch in between.
    if (A.counter) {
        // hand-off without fences: sc1 partial stores, every wave drains, one relaxed agent-scope ticket, ONE acquire + plain loads
        // in the last arriver (a __threadfence() here writes back / invalidates the L2 once per workgroup: measured 70 % slower)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) red[0] = (double)__hip_atomic_fetch_add(A.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = (int)red[0];
        if (ticket == (int)gridDim.x - 1) {
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire, then plain pipelined loads
            __syncthreads();
            const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
            const int nparts = (int)gridDim.x;
            for (int cb = 0; cb < A.C; cb += 32) {
                const int c = cb + cl;
                double s0 = 0.0, s1 = 0.0;
                if (c < A.C)
#pragma unroll 8
                    for (int p = rl; p < nparts; p += 8) {
                        s0 += A.partials[((size_t)p * 2 + 0) * A.C + c];
                        s1 += A.partials[((size_t)p * 2 + 1) * A.C + c];
                    }
                __syncthreads();
                red[(0 * 8 + rl) * 32 + cl] = s0;
                red[(1 * 8 + rl) * 32 + cl] = s1;
                __syncthreads();
                if (rl == 0 && c < A.C) {
                    s0 = 0.0;
                    s1 = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        s0 += red[(0 * 8 + i) * 32 + cl];
                        s1 += red[(1 * 8 + i) * 32 + cl];
                    }
                    if (A.dbeta) A.dbeta[c] = A.acc_param ? A.dbeta[c] + (float)s0 : (float)s0;
                    if (A.dgamma) A.dgamma[c] = A.acc_param ? A.dgamma[c] + (float)s1 : (float)s1;
                    ((float*)A.c1)[c] = (float)(s0 / A.count);
                    ((float*)A.c2)[c] = (float)(s1 / A.count);
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(A.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// BN term, between passes: dbeta = sum G, dgamma = sum G*xhat -> parameter grads (+ optional accumulate)
// and the per-channel constants c1 = dbeta / M, c2 = dgamma / M for the apply pass.
__device__ __forceinline__ void bpb_bn_bwd_finalize_body(int blk, double (*red)[BPB_FIN_LANES][BPB_FIN_CH], const double* __restrict__ partials,
                                                         int nparts, int C, double count, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int accumulate, float* __restrict__ c1,
                                                         float* __restrict__ c2)
{
    double s, q;
    if (!bpb_fin_column_sums(blk, red, partials, nparts, C, s, q)) return;
    const int c = blk * BPB_FIN_CH + threadIdx.x % BPB_FIN_CH;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
    c1[c] = (float)(s / count);
    c2[c] = (float)(q / count);
}

__global__ __launch_bounds__(1024) void bpb_bn_bwd_finalize_kernel(const double* __restrict__ partials, int nparts, int C,
                                                                  double count, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int accumulate,
                                                                  float* __restrict__ c1, float* __restrict__ c2)
{
    __shared__ double red[2][BPB_FIN_LANES][BPB_FIN_CH];
    bpb_bn_bwd_finalize_body(blockIdx.x, red, partials, nparts, C, count, dgamma, dbeta, accumulate, c1, c2);
}

__global__ __launch_bounds__(1024) void bpb_bn_bwd_finalize_multi_kernel(const BpbBnBwdFinDesc* __restrict__ descs, BpbBlkBegins bb)
{
    __shared__ double red[2][BPB_FIN_LANES][BPB_FIN_CH];
    const int di = bpb_find_problem(bb, (int)blockIdx.x);
    const BpbBnBwdFinDesc D = descs[di];
    bpb_bn_bwd_finalize_body(blockIdx.x - D.blk_begin, red, D.partials, D.nparts, D.C, D.count, D.dgamma, D.dbeta, D.accumulate, D.c1, D.c2);
}

// BN term, pass 2: dsrc = scale * (G - c1 - xhat * c2)
__device__ __forceinline__ void bpb_term_bwd_bn_apply_body(const BpbTermBwdArgs& A, int blk, int nblk)
{
    const int c4 = A.C >> 2;
    const long total = (long)A.N * A.Hs * A.Ws * c4;
    for (long i = blk * 256L + threadIdx.x; i < total; i += nblk * 256L) {
        const int cq = (int)(i % c4);
        const long q = i / c4;
        const f32x4 g = bpb_window_grad(A, q, cq);
        const f32x4 x = *(const f32x4*)(A.src + q * A.C + cq * 4);
        const f32x4 mu = *(const f32x4*)(A.mean + cq * 4);
        const f32x4 is = *(const f32x4*)(A.invstd + cq * 4);
        const f32x4 sc = *(const f32x4*)(A.scale + cq * 4);
        const f32x4 k1 = *(const f32x4*)(A.c1 + cq * 4);
        const f32x4 k2 = *(const f32x4*)(A.c2 + cq * 4);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = sc[e] * (g[e] - k1[e] - (x[e] - mu[e]) * is[e] * k2[e]);
        float* o = A.dsrc + q * A.C + cq * 4;
        if (A.accumulate) {
            const f32x4 old = *(const f32x4*)o;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] += old[e];
        }
        *(f32x4*)o = d;
        if (A.dsrc2) {   // the identity (skip) term of the same fuse op, same resolution (up == 0): dsrc2 (+)= G
            float* o2 = A.dsrc2 + q * A.C + cq * 4;
            f32x4 g2 = g;
            if (A.accumulate2) {
                const f32x4 old2 = *(const f32x4*)o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) g2[e] += old2[e];
            }
            *(f32x4*)o2 = g2;
        }
    }
}

__global__ __launch_bounds__(256) void bpb_term_bwd_bn_apply_kernel(BpbTermBwdArgs A)
{
    bpb_term_bwd_bn_apply_body(A, blockIdx.x, gridDim.x);
}

template <int MODE>
__global__ __launch_bounds__(256) void bpb_term_bwd_multi_kernel(const BpbTermBwdArgs* __restrict__ descs, BpbBlkBegins bb)
{
    __shared__ double red[MODE == 1 ? 256 * 8 : 1];
    const BpbTermBwdArgs A = descs[bpb_find_problem(bb, (int)blockIdx.x)];
    const int blk = blockIdx.x - A.blk_begin;
    if (MODE == 0) bpb_term_bwd_identity_body(A, blk, A.nblk);
    else if (MODE == 1) bpb_term_bwd_bn_reduce_body(A, blk, A.nblk, red);
    else bpb_term_bwd_bn_apply_body(A, blk, A.nblk);
}

// eval-mode / frozen-stat BN term backward is not on the training path and is not provided.

// ------------------------------------ C ABI ------------------------------------------
static int ew_grid(long total_vec)
{
    long g = (total_vec + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int bpb_bn_finalize(const double* partials, int nparts, int C, double count, const float* gamma, const float* beta,
                    float eps, float momentum, float* scale, float* shift, float* mean, float* invstd,
                    float* running_mean, float* running_var, hipStream_t stream)
{
    BPB_REQUIRE(nparts >= 1 && C >= 1 && count >= 1.0, "bpb_bn_finalize: bad sizes");
    hipLaunchKernelGGL(bpb_bn_finalize_kernel, dim3(bpb_cdiv(C, BPB_FIN_CH)), dim3(1024), 0, stream, partials, nparts, C, count,
                       gamma, beta, eps, momentum, scale, shift, mean, invstd, running_mean, running_var);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn_eval_affine_batched(const BpbBnEvalDesc* d_descs, int ndescs, int total_blocks, float eps, hipStream_t stream)
{
    BPB_REQUIRE(ndescs >= 1 && total_blocks >= 1, "bpb_bn_eval_affine_batched: empty");
    hipLaunchKernelGGL(bpb_bn_eval_affine_batched_kernel, dim3(total_blocks), dim3(256), 0, stream, d_descs, ndescs, eps);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, hipStream_t stream)
{
    BPB_REQUIRE(C >= 1, "bpb_bn_eval_affine: C");
    hipLaunchKernelGGL(bpb_bn_eval_affine_kernel, dim3(bpb_cdiv(C, 256)), dim3(256), 0, stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    BPB_LAUNCH_OK();
    return 0;
}

// partials must hold nblocks*2*C doubles.  C % 4 == 0.
int bpb_channel_stats(const float* x, long P, int C, double* partials, int nblocks, hipStream_t stream)
{
    BPB_REQUIRE(C % 4 == 0 && P >= 1 && nblocks >= 1, "bpb_channel_stats: bad sizes");
    hipLaunchKernelGGL(bpb_channel_stats_kernel, dim3(nblocks), dim3(256), 256 * 8 * sizeof(double), stream, x, P, C,
                       partials);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_fuse_fwd(const BpbFuseArgs* a, hipStream_t stream)
{
    BPB_REQUIRE(a->nterms >= 1 && a->nterms <= BPB_MAX_TERMS, "bpb_fuse_fwd: nterms=%d", a->nterms);
    BPB_REQUIRE(a->C % 4 == 0, "bpb_fuse_fwd: C must be a multiple of 4");
    for (int t = 0; t < a->nterms; ++t)
        BPB_REQUIRE(a->up[t] >= 0 && (a->H % (1 << a->up[t])) == 0 && (a->W % (1 << a->up[t])) == 0,
                    "bpb_fuse_fwd: output dims must be multiples of the upsample factor");
    const long total = (long)a->N * a->H * a->W * (a->C / 4);
    BPB_REQUIRE(total < (1L << 32), "bpb_fuse_fwd: tensor too large for 32-bit pixel index");
    hipLaunchKernelGGL(bpb_fuse_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, *a);
    BPB_LAUNCH_OK();
    return 0;
}

// mode 0: identity term, 1: BN reduce (nblocks partial rows), 2: BN apply
int bpb_term_bwd(const BpbTermBwdArgs* a, int mode, int nblocks, hipStream_t stream)
{
    BPB_REQUIRE(a->C % 4 == 0, "bpb_term_bwd: C must be a multiple of 4");
    const long total = (long)a->N * a->Hs * a->Ws * (a->C / 4);
    if (mode == 0) {
        hipLaunchKernelGGL(bpb_term_bwd_identity_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, *a);
    } else if (mode == 1) {
        BPB_REQUIRE(nblocks >= 1, "bpb_term_bwd: nblocks");
        hipLaunchKernelGGL(bpb_term_bwd_bn_reduce_kernel, dim3(nblocks), dim3(256), 0, stream, *a);
    } else if (mode == 2) {
        hipLaunchKernelGGL(bpb_term_bwd_bn_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, *a);
    } else {
        return bpb_set_error(-1, "bpb_term_bwd: mode %d", mode);
    }
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn_bwd_finalize(const double* partials, int nparts, int C, double count, float* dgamma, float* dbeta,
                        int accumulate, float* c1, float* c2, hipStream_t stream)
{
    BPB_REQUIRE(nparts >= 1 && C >= 1, "bpb_bn_bwd_finalize: bad sizes");
    hipLaunchKernelGGL(bpb_bn_bwd_finalize_kernel, dim3(bpb_cdiv(C, BPB_FIN_CH)), dim3(1024), 0, stream, partials, nparts, C,
                       count, dgamma, dbeta, accumulate, c1, c2);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_fuse_fwd_multi(const BpbFuseArgs* d_descs, const BpbFuseArgs* h_descs, int n, int total_blocks, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 16 && total_blocks >= 1, "bpb_fuse_fwd_multi: n=%d blocks=%d", n, total_blocks);
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const BpbFuseArgs* a = &h_descs[i];
        BPB_REQUIRE(a->nterms >= 1 && a->nterms <= BPB_MAX_TERMS && a->C % 4 == 0, "bpb_fuse_fwd_multi: record %d", i);
        for (int t = 0; t < a->nterms; ++t)
            BPB_REQUIRE(a->up[t] >= 0 && (a->H % (1 << a->up[t])) == 0 && (a->W % (1 << a->up[t])) == 0,
                        "bpb_fuse_fwd_multi: output dims must be multiples of the upsample factor");
        BPB_REQUIRE((long)a->N * a->H * a->W * (a->C / 4) < (1L << 32), "bpb_fuse_fwd_multi: tensor too large for 32-bit pixel index");
        BPB_REQUIRE(a->blk_begin == blk && a->nblk >= 1, "bpb_fuse_fwd_multi: blk_begin mismatch");
        blk += a->nblk;
    }
    BPB_REQUIRE(blk == total_blocks, "bpb_fuse_fwd_multi: block count mismatch");
    hipLaunchKernelGGL(bpb_fuse_fwd_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, d_descs, bpb_blk_begins(h_descs, n));
    BPB_LAUNCH_OK();
    return 0;
}

// mode 0: identity terms, 1: BN reduce (nblk partial rows per record), 2: BN apply
int bpb_term_bwd_multi(const BpbTermBwdArgs* d_descs, const BpbTermBwdArgs* h_descs, int n, int total_blocks, int mode,
                       hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 16 && total_blocks >= 1, "bpb_term_bwd_multi: n=%d blocks=%d", n, total_blocks);
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        BPB_REQUIRE(h_descs[i].C % 4 == 0 && h_descs[i].blk_begin == blk && h_descs[i].nblk >= 1, "bpb_term_bwd_multi: record %d", i);
        BPB_REQUIRE(h_descs[i].counter == nullptr, "bpb_term_bwd_multi: fused finalisation is a single-launch feature");
        blk += h_descs[i].nblk;
    }
    BPB_REQUIRE(blk == total_blocks, "bpb_term_bwd_multi: block count mismatch");
    const BpbBlkBegins bb = bpb_blk_begins(h_descs, n);
    if (mode == 0) hipLaunchKernelGGL(bpb_term_bwd_multi_kernel<0>, dim3(total_blocks), dim3(256), 0, stream, d_descs, bb);
    else if (mode == 1) hipLaunchKernelGGL(bpb_term_bwd_multi_kernel<1>, dim3(total_blocks), dim3(256), 0, stream, d_descs, bb);
    else if (mode == 2) hipLaunchKernelGGL(bpb_term_bwd_multi_kernel<2>, dim3(total_blocks), dim3(256), 0, stream, d_descs, bb);
    else return bpb_set_error(-1, "bpb_term_bwd_multi: mode %d", mode);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn_finalize_multi(const BpbBnFinDesc* d_descs, const BpbBnFinDesc* h_descs, int n, int total_blocks, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 16, "bpb_bn_finalize_multi: n=%d", n);
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        BPB_REQUIRE(h_descs[i].nparts >= 1 && h_descs[i].C >= 1 && h_descs[i].count >= 1.0 && h_descs[i].blk_begin == blk,
                    "bpb_bn_finalize_multi: record %d", i);
        blk += bpb_cdiv(h_descs[i].C, BPB_FIN_CH);
    }
    BPB_REQUIRE(blk == total_blocks, "bpb_bn_finalize_multi: block count mismatch");
    hipLaunchKernelGGL(bpb_bn_finalize_multi_kernel, dim3(total_blocks), dim3(1024), 0, stream, d_descs, bpb_blk_begins(h_descs, n));
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn_bwd_finalize_multi(const BpbBnBwdFinDesc* d_descs, const BpbBnBwdFinDesc* h_descs, int n, int total_blocks,
                              hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 16, "bpb_bn_bwd_finalize_multi: n=%d", n);
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        BPB_REQUIRE(h_descs[i].nparts >= 1 && h_descs[i].C >= 1 && h_descs[i].blk_begin == blk, "bpb_bn_bwd_finalize_multi: record %d", i);
        blk += bpb_cdiv(h_descs[i].C, BPB_FIN_CH);
    }
    BPB_REQUIRE(blk == total_blocks, "bpb_bn_bwd_finalize_multi: block count mismatch");
    hipLaunchKernelGGL(bpb_bn_bwd_finalize_multi_kernel, dim3(total_blocks), dim3(1024), 0, stream, d_descs, bpb_blk_begins(h_descs, n));
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
