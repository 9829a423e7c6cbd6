// Small dense layers after pooling: Linear (fwd, dX, dW) on the fp32 MFMA pipe and BatchNorm1d (+ReLU).
//
// Replaces torchreid/models/bpbreid.py:324-350 (AfterPoolingDimReduceLayer: Linear(C->D,bias)+BN1d+ReLU),
// :398-415 (BNClassifier: BN1d with frozen bias -> Linear(D->classes, no bias)) and :261-279 (per-part
// classifiers).  These GEMMs are skinny (M = N or N*K rows = 64..320, K up to 2560): one generic strided
// kernel C = A(MxK) . B(KxN) with split-K so that more than a handful of CUs participate; the slabs are
// summed in a fixed order (deterministic) by the reduce kernel, which also adds the bias.
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Grouped launch: the <= BPB_GEMM_MAX independent GEMMs of one stage of the head (the 3 + K dimension-reduce layers, the
// 4 + K identity classifiers, or their dX / dW products) run as ONE launch.  Each of these GEMMs alone is a 20-30 us latency
// chain (a dozen k-steps of dependent global loads on a few dozen workgroups); 37 of them back to back were ~1 ms of the
// train step.  The descriptors travel by value in the kernel-argument segment (<= 3 KiB): no device-side table to upload.
//
// C_part[split][M][N] = sum_{k in split} A[m*sam + k*sak] * B[k*sbk + n*sbn]
// block = 256 threads = 2x2 waves, block tile 64x64, k-step 16, LDS tiles stored k-major ([k][64+pad]); the global loads of
// k-step i+1 are in flight (registers) while the MFMAs of k-step i run.
struct BpbGemmGroup {
    BpbGemmProb p[BPB_GEMM_MAX];
};

__global__ __launch_bounds__(256) void bpb_gemm_grouped_kernel(const BpbGemmGroup G, int nprobs, float* __restrict__ ws)
{
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    int bid = blockIdx.x, pi = 0;
    for (int i = 1; i < nprobs; ++i)
        if (bid >= G.p[i].blk_begin) pi = i;
    const BpbGemmProb& P = G.p[pi];
    bid -= P.blk_begin;
    const int M = P.M, N = P.N;
    const int tiles = P.tiles_m * P.tiles_n;
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int k_begin = split * P.kchunk, k_end = min(P.K, k_begin + P.kchunk);
    const float* __restrict__ A = P.A;
    const float* __restrict__ B = P.B;
    const long sam = P.sam, sak = P.sak, sbk = P.sbk, sbn = P.sbn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // element (row, k) of the 64 x 16 tile that this thread stages in round i: k fastest when the operand is k-contiguous
    const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);
    int am[4], ak[4], bn[4], bk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (a_kfast) { ak[i] = threadIdx.x & 15; am[i] = (threadIdx.x >> 4) + 16 * i; }
        else { am[i] = threadIdx.x & 63; ak[i] = (threadIdx.x >> 6) + 4 * i; }
        if (b_kfast) { bk[i] = threadIdx.x & 15; bn[i] = (threadIdx.x >> 4) + 16 * i; }
        else { bn[i] = threadIdx.x & 63; bk[i] = (threadIdx.x >> 6) + 4 * i; }
    }
    float ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = tm * 64 + am[i], gk = k0 + ak[i];
            ra[i] = (gm < M && gk < k_end) ? A[gm * sam + gk * sak] : 0.f;
            const int gn = tn * 64 + bn[i], gkb = k0 + bk[i];
            rb[i] = (gn < N && gkb < k_end) ? B[gkb * sbk + gn * sbn] : 0.f;
        }
    };
    gload(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[ak[i]][am[i]] = ra[i];
            Bs[bk[i]][bn[i]] = rb[i];
        }
        __syncthreads();
        if (k0 + 16 < k_end) gload(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) acc = MFMA32(As[kk + half][wm + l31], Bs[kk + half][wn + l31], acc);
    }
    float* Cp = ws + P.ws_off + (long)split * M * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int col = tn * 64 + wn + l31;
        if (row < M && col < N) Cp[(long)row * N + col] = acc[r];
    }
}

// C[m*ldc + n] (+)= sum_s part[s][m][n] + bias[n], slabs in a fixed order (deterministic).  A problem that was joined to its
// predecessor (`join`: further k-slices of the same output, e.g. the K part products of the shared dimension-reduce weight
// gradient) has no blocks here: its slabs follow the leader's in the workspace and are counted in the leader's red_slabs.
__global__ __launch_bounds__(256) void bpb_gemm_reduce_grouped_kernel(const BpbGemmGroup G, int nprobs, const float* __restrict__ ws)
{
    int bid = blockIdx.x, pi = 0;
    for (int i = 1; i < nprobs; ++i)
        if (G.p[i].red_blocks > 0 && bid >= G.p[i].red_begin) pi = i;
    const BpbGemmProb& P = G.p[pi];
    bid -= P.red_begin;
    const long total = (long)P.M * P.N;
    const float* __restrict__ part = ws + P.ws_off;
    const float* __restrict__ bias = P.bias;
    const int N = P.N, nslab = P.red_slabs, accumulate = P.accumulate;
    for (long i = bid * 256L + threadIdx.x; i < total; i += P.red_blocks * 256L) {
        const int n = (int)(i % N);
        const long m = i / N;
        float s = bias ? bias[n] : 0.f;
        for (int sp = 0; sp < nslab; ++sp) s += part[(long)sp * total + i];
        float* o = P.C + m * P.ldc + n;
        *o = accumulate ? *o + s : s;
    }
}

// column sums: out[n] (+)= sum_m X[m][n]   (bias gradient of Linear)
__global__ __launch_bounds__(256) void bpb_colsum_kernel(const float* __restrict__ X, float* __restrict__ out, int M, int N,
                                                         int accumulate)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += X[(long)m * N + n];
        out[n] = accumulate ? out[n] + s : s;
    }
}

// Tall matrices (the bias gradient of a 1x1 convolution over N*H*W pixels, bpbreid.py:283-293 / hrnet.py:361-371): 64 columns
// x 16 row lanes per block, fixed summation order (row lanes strided, then lanes 0..15) -> deterministic.
__global__ __launch_bounds__(1024) void bpb_colsum_tall_kernel(const float* __restrict__ X, float* __restrict__ out, int M, int N,
                                                               int accumulate)
{
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = rl;
        for (; m + 48 < M; m += 64) {
            s0 += X[(long)m * N + n];
            s1 += X[(long)(m + 16) * N + n];
            s2 += X[(long)(m + 32) * N + n];
            s3 += X[(long)(m + 48) * N + n];
        }
        for (; m < M; m += 16) s0 += X[(long)m * N + n];
    }
    red[rl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i][cl];
        out[n] = accumulate ? out[n] + s : s;
    }
}

// ---- BatchNorm1d over rows, fused optional ReLU -------------------------------------------------
// training: batch statistics (biased var for normalisation, unbiased for running_var), saves mean/invstd.
// x rows have stride ldx (so that a [N][K][D] tensor can be normalised per part column-block).
// Workgroup = 32 features x 8 row lanes: lane r handles rows r, r+8, ... with every load of a sweep in flight at once (a
// one-thread-per-feature loop over 64..320 rows is a chain of dependent L2 round trips: 35-75 us per call); the 8 lane sums
// are combined in a fixed order through LDS (deterministic).
#define BN1D_FEATS 32
#define BN1D_LANES 8
__device__ __forceinline__ void bpb_bn1d_fwd_body(int blk, const float* __restrict__ x, long ldx, float* __restrict__ y,
                                                  long ldy, int R, int F, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ running_mean,
                                                  float* __restrict__ running_var, float* __restrict__ save_mean,
                                                  float* __restrict__ save_invstd, float eps, float momentum,
                                                  int training, int relu)
{
    __shared__ double red[2][BN1D_LANES][BN1D_FEATS];
    __shared__ float stat[2][BN1D_FEATS];
    const int fl = threadIdx.x % BN1D_FEATS, rl = threadIdx.x / BN1D_FEATS;
    const int f = blk * BN1D_FEATS + fl;
    const bool live = f < F;
    if (training) {
        double s = 0.0, q = 0.0;
        if (live) {
            int r = rl;
            for (; r + 7 * BN1D_LANES < R; r += 8 * BN1D_LANES) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = x[(long)(r + u * BN1D_LANES) * ldx + f];
#pragma unroll
                for (int u = 0; u < 8; ++u) { s += (double)v[u]; q += (double)v[u] * (double)v[u]; }
            }
            for (; r < R; r += BN1D_LANES) {
                const double v = (double)x[(long)r * ldx + f];
                s += v;
                q += v * v;
            }
        }
        red[0][rl][fl] = s;
        red[1][rl][fl] = q;
        __syncthreads();
        if (rl == 0 && live) {
            s = 0.0;
            q = 0.0;
#pragma unroll
            for (int i = 0; i < BN1D_LANES; ++i) { s += red[0][i][fl]; q += red[1][i][fl]; }
            const double mu = s / R;
            double var = q / R - mu * mu;
            if (var < 0.0) var = 0.0;
            const float mean = (float)mu, invstd = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unbiased = R > 1 ? var * R / (R - 1.0) : var;
                running_mean[f] = (1.f - momentum) * running_mean[f] + momentum * mean;
                running_var[f] = (1.f - momentum) * running_var[f] + momentum * (float)unbiased;
            }
            if (save_mean) { save_mean[f] = mean; save_invstd[f] = invstd; }
            stat[0][fl] = mean;
            stat[1][fl] = invstd;
        }
        __syncthreads();
    } else if (rl == 0 && live) {
        stat[0][fl] = running_mean[f];
        stat[1][fl] = 1.f / sqrtf(running_var[f] + eps);
    }
    if (!training) __syncthreads();
    if (!live) return;
    const float mean = stat[0][fl], invstd = stat[1][fl];
    const float g = gamma ? gamma[f] : 1.f, b = beta ? beta[f] : 0.f;
#pragma unroll 4
    for (int r = rl; r < R; r += BN1D_LANES) {
        float v = (x[(long)r * ldx + f] - mean) * invstd * g + b;
        if (relu && v < 0.f) v = 0.f;
        y[(long)r * ldy + f] = v;
    }
}

__global__ __launch_bounds__(256) void bpb_bn1d_fwd_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y,
                                                           long ldy, int R, int F, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, float eps, float momentum,
                                                           int training, int relu)
{
    bpb_bn1d_fwd_body((int)blockIdx.x, x, ldx, y, ldy, R, F, gamma, beta, running_mean, running_var, save_mean, save_invstd, eps, momentum,
                      training, relu);
}

// The independent BatchNorm1d layers of one stage of the head (3 + 1 dimension-reduce layers, 4 + K BN-necks) in ONE launch: the
// descriptors travel by value in the kernel arguments, block -> layer through the first-block prefix (13 launches of ~5 us on the
// head's latency chain became 2, the 9 backward ones 2).
struct BpbBn1dPack {
    BpbBn1dDesc d[BPB_BN1D_MAX];
};
__global__ __launch_bounds__(256) void bpb_bn1d_fwd_multi_kernel(BpbBn1dPack pk, BpbBlkBegins bb, float eps, float momentum, int training)
{
    const int i = bpb_find_problem(bb, (int)blockIdx.x);
    const BpbBn1dDesc& D = pk.d[i];
    bpb_bn1d_fwd_body((int)blockIdx.x - D.blk_begin, D.x, D.ldx, D.y, D.ldy, D.R, D.F, D.gamma, D.beta, D.running_mean, D.running_var,
                      D.save_mean, D.save_invstd, eps, momentum, training, D.relu);
}

// backward (training statistics): dx = g*invstd/R * (R*dy - sum dy - xhat * sum(dy*xhat)); ReLU mask from y.
__device__ __forceinline__ void bpb_bn1d_bwd_body(int blk, const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                  long ldx, const float* __restrict__ y, long ldy,
                                                  float* __restrict__ dx, long lddx, int R, int F,
                                                  const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                  const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, int relu, int accumulate_params)
{
    __shared__ double red[2][BN1D_LANES][BN1D_FEATS];
    __shared__ float kk[2][BN1D_FEATS];
    const int fl = threadIdx.x % BN1D_FEATS, rl = threadIdx.x / BN1D_FEATS;
    const int f = blk * BN1D_FEATS + fl;
    const bool live = f < F;
    const float mean = live ? save_mean[f] : 0.f, invstd = live ? save_invstd[f] : 0.f;
    const float g = (gamma && live) ? gamma[f] : 1.f;
    double s = 0.0, q = 0.0;
    if (live) {
        int r = rl;
        for (; r + 3 * BN1D_LANES < R; r += 4 * BN1D_LANES) {
            float d[4], xv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = dy[(long)(r + u * BN1D_LANES) * lddy + f];
                xv[u] = x[(long)(r + u * BN1D_LANES) * ldx + f];
                yv[u] = relu ? y[(long)(r + u * BN1D_LANES) * ldy + f] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dd = yv[u] > 0.f ? d[u] : 0.f;
                s += (double)dd;
                q += (double)dd * (double)((xv[u] - mean) * invstd);
            }
        }
        for (; r < R; r += BN1D_LANES) {
            float d = dy[(long)r * lddy + f];
            if (relu && !(y[(long)r * ldy + f] > 0.f)) d = 0.f;
            s += (double)d;
            q += (double)d * (double)((x[(long)r * ldx + f] - mean) * invstd);
        }
    }
    red[0][rl][fl] = s;
    red[1][rl][fl] = q;
    __syncthreads();
    if (rl == 0 && live) {
        s = 0.0;
        q = 0.0;
#pragma unroll
        for (int i = 0; i < BN1D_LANES; ++i) { s += red[0][i][fl]; q += red[1][i][fl]; }
        if (dgamma) dgamma[f] = accumulate_params ? dgamma[f] + (float)q : (float)q;
        if (dbeta) dbeta[f] = accumulate_params ? dbeta[f] + (float)s : (float)s;
        kk[0][fl] = (float)(s / R);
        kk[1][fl] = (float)(q / R);
    }
    __syncthreads();
    if (!live) return;
    const float k1 = kk[0][fl], k2 = kk[1][fl];
#pragma unroll 4
    for (int r = rl; r < R; r += BN1D_LANES) {
        float d = dy[(long)r * lddy + f];
        if (relu && !(y[(long)r * ldy + f] > 0.f)) d = 0.f;
        const float xh = (x[(long)r * ldx + f] - mean) * invstd;
        dx[(long)r * lddx + f] = g * invstd * (d - k1 - xh * k2);
    }
}

__global__ __launch_bounds__(256) void bpb_bn1d_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                           long ldx, const float* __restrict__ y, long ldy,
                                                           float* __restrict__ dx, long lddx, int R, int F,
                                                           const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int relu, int accumulate_params)
{
    bpb_bn1d_bwd_body((int)blockIdx.x, dy, lddy, x, ldx, y, ldy, dx, lddx, R, F, gamma, save_mean, save_invstd, dgamma, dbeta, relu,
                      accumulate_params);
}

__global__ __launch_bounds__(256) void bpb_bn1d_bwd_multi_kernel(BpbBn1dPack pk, BpbBlkBegins bb)
{
    const int i = bpb_find_problem(bb, (int)blockIdx.x);
    const BpbBn1dDesc& D = pk.d[i];
    bpb_bn1d_bwd_body((int)blockIdx.x - D.blk_begin, D.dy, D.lddy, D.x, D.ldx, D.y, D.ldy, D.dx, D.lddx, D.R, D.F, D.gamma, D.save_mean,
                      D.save_invstd, D.dgamma, D.dbeta, D.relu, D.accumulate_params);
}

extern "C" {

// Grouped launch of up to BPB_GEMM_MAX independent GEMMs C = A . B (+ bias) (strided operands, split-K slabs in `ws`, fixed
// summation order).  probs: HOST array; M, N, K, operands, accumulate and join are inputs, the launch geometry fields are
// filled in here.  ws == nullptr: only compute the workspace need (floats) into *need_out.
// Replaces the F.linear calls of bpbreid.py:324-350, :398-415 and their autograd products.
int bpb_gemm_grouped(BpbGemmProb* probs, int nprobs, float* ws, long ws_floats, long* need_out, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= BPB_GEMM_MAX, "bpb_gemm_grouped: nprobs=%d out of range", nprobs);
    long need = 0;
    int blk = 0, red = 0, leader = -1;
    for (int i = 0; i < nprobs; ++i) {
        BpbGemmProb& p = probs[i];
        BPB_REQUIRE(p.M >= 1 && p.N >= 1 && p.K >= 1 && p.A && p.B && p.C, "bpb_gemm_grouped: problem %d: bad sizes %d %d %d", i, p.M, p.N, p.K);
        p.tiles_m = bpb_cdiv(p.M, 64);
        p.tiles_n = bpb_cdiv(p.N, 64);
        // k-slices of <= 256 products keep a workgroup's chain of dependent k-steps short; never slices below 128
        int nsplit = 1;
        while (nsplit < 16 && p.K / nsplit > 256 && p.K / (nsplit * 2) >= 128) nsplit *= 2;
        p.nsplit = nsplit;
        p.kchunk = (bpb_cdiv(p.K, nsplit) + 15) & ~15;
        p.blk_begin = blk;
        blk += p.tiles_m * p.tiles_n * nsplit;
        p.ws_off = need;
        need += (long)nsplit * p.M * p.N;
        if (p.join) {
            BPB_REQUIRE(leader >= 0 && probs[leader].M == p.M && probs[leader].N == p.N && probs[leader].C == p.C,
                        "bpb_gemm_grouped: problem %d joins a predecessor of another shape / output", i);
            probs[leader].red_slabs += nsplit;
            p.red_blocks = 0;
            p.red_begin = red;
            p.red_slabs = 0;
        } else {
            leader = i;
            long g = ((long)p.M * p.N + 1023) / 1024;     // four elements per thread
            p.red_blocks = (int)(g > 512 ? 512 : g);
            p.red_begin = red;
            p.red_slabs = nsplit;
            red += p.red_blocks;
        }
    }
    if (need_out) *need_out = need;
    if (!ws) return 0;
    BPB_REQUIRE(ws_floats >= need, "bpb_gemm_grouped: workspace of %ld floats, %ld needed", ws_floats, need);
    BpbGemmGroup G;
    for (int i = 0; i < nprobs; ++i) G.p[i] = probs[i];
    hipLaunchKernelGGL(bpb_gemm_grouped_kernel, dim3(blk), dim3(256), 0, stream, G, nprobs, ws);
    hipLaunchKernelGGL(bpb_gemm_reduce_grouped_kernel, dim3(red), dim3(256), 0, stream, G, nprobs, (const float*)ws);
    BPB_LAUNCH_OK();
    return 0;
}

// One GEMM (the grouped launch with a single problem).  workspace: nsplit*M*N floats; ws == nullptr returns the split count it
// wants through *nsplit_out.
int bpb_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, const float* bias,
             int M, int N, int K, int accumulate, float* ws, int* nsplit_out, hipStream_t stream)
{
    BPB_REQUIRE(M >= 1 && N >= 1 && K >= 1, "bpb_gemm: bad sizes %d %d %d", M, N, K);
    BpbGemmProb p = {};
    p.A = A; p.sam = sam; p.sak = sak; p.B = B; p.sbk = sbk; p.sbn = sbn; p.C = C; p.ldc = ldc; p.bias = bias;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate;
    long need = 0;
    int rc = bpb_gemm_grouped(&p, 1, nullptr, 0, &need, stream);
    if (rc) return rc;
    if (nsplit_out) *nsplit_out = p.nsplit;
    if (!ws) return 0;
    return bpb_gemm_grouped(&p, 1, ws, need, nullptr, stream);
}

int bpb_colsum(const float* X, float* out, int M, int N, int accumulate, hipStream_t stream)
{
    // (one thread per column walking M rows is a chain of M dependent loads: 74 us for the 320 x 512 bias gradient of the parts'
    //  dimension-reduce layer on the head's backward chain -- sixteen row lanes from 64 rows on: ~6 us)
    if (M > 64)
        hipLaunchKernelGGL(bpb_colsum_tall_kernel, dim3(bpb_cdiv(N, 64)), dim3(1024), 0, stream, X, out, M, N, accumulate);
    else
        hipLaunchKernelGGL(bpb_colsum_kernel, dim3(bpb_cdiv(N, 256)), dim3(256), 0, stream, X, out, M, N, accumulate);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn1d_fwd(const float* x, long ldx, float* y, long ldy, int R, int F, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float* save_mean, float* save_invstd, float eps, float momentum,
                 int training, int relu, hipStream_t stream)
{
    BPB_REQUIRE(R >= 1 && F >= 1, "bpb_bn1d_fwd: bad sizes");
    BPB_REQUIRE(training || (running_mean && running_var), "bpb_bn1d_fwd: eval mode needs running statistics");
    hipLaunchKernelGGL(bpb_bn1d_fwd_kernel, dim3(bpb_cdiv(F, BN1D_FEATS)), dim3(256), 0, stream, x, ldx, y, ldy, R, F, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, eps, momentum, training, relu);
    BPB_LAUNCH_OK();
    return 0;
}

static int bn1d_pack(const BpbBn1dDesc* h, int n, BpbBn1dPack& pk, BpbBlkBegins& bb, const char* who)
{
    BPB_REQUIRE(h != nullptr && n >= 1 && n <= BPB_BN1D_MAX, "%s: %d layers (1..%d)", who, n, BPB_BN1D_MAX);
    int blk = 0;
    for (int i = 0; i < BPB_BN1D_MAX; ++i) {
        bb.begin[i] = 0x7fffffff;
        if (i >= n) continue;
        BPB_REQUIRE(h[i].R >= 1 && h[i].F >= 1 && h[i].x != nullptr, "%s: layer %d: bad sizes", who, i);
        pk.d[i] = h[i];
        pk.d[i].blk_begin = blk;
        bb.begin[i] = blk;
        blk += bpb_cdiv(h[i].F, BN1D_FEATS);
    }
    return blk;
}

// h_descs: HOST array of n <= BPB_BN1D_MAX layers (copied by value into the launch); bpbreid.py:335, :405 (nn.BatchNorm1d, training
// and eval semantics of bpb_bn1d_fwd) for all of them at once
int bpb_bn1d_fwd_multi(const BpbBn1dDesc* h_descs, int n, float eps, float momentum, int training, hipStream_t stream)
{
    BpbBn1dPack pk;
    BpbBlkBegins bb;
    const int blk = bn1d_pack(h_descs, n, pk, bb, "bpb_bn1d_fwd_multi");
    if (blk < 0) return blk;
    for (int i = 0; i < n; ++i)
        BPB_REQUIRE(h_descs[i].y != nullptr && (training || (h_descs[i].running_mean && h_descs[i].running_var)),
                    "bpb_bn1d_fwd_multi: layer %d: eval mode needs running statistics", i);
    hipLaunchKernelGGL(bpb_bn1d_fwd_multi_kernel, dim3(blk), dim3(256), 0, stream, pk, bb, eps, momentum, training);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn1d_bwd_multi(const BpbBn1dDesc* h_descs, int n, hipStream_t stream)
{
    BpbBn1dPack pk;
    BpbBlkBegins bb;
    const int blk = bn1d_pack(h_descs, n, pk, bb, "bpb_bn1d_bwd_multi");
    if (blk < 0) return blk;
    for (int i = 0; i < n; ++i)
        BPB_REQUIRE(h_descs[i].dy != nullptr && h_descs[i].dx != nullptr && h_descs[i].save_mean != nullptr, "bpb_bn1d_bwd_multi: layer %d: null operand", i);
    hipLaunchKernelGGL(bpb_bn1d_bwd_multi_kernel, dim3(blk), dim3(256), 0, stream, pk, bb);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_bn1d_bwd(const float* dy, long lddy, const float* x, long ldx, const float* y, long ldy, float* dx, long lddx, int R,
                 int F, const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                 int relu, int accumulate_params, hipStream_t stream)
{
    BPB_REQUIRE(R >= 1 && F >= 1, "bpb_bn1d_bwd: bad sizes");
    hipLaunchKernelGGL(bpb_bn1d_bwd_kernel, dim3(bpb_cdiv(F, BN1D_FEATS)), dim3(256), 0, stream, dy, lddy, x, ldx, y, ldy, dx, lddx,
                       R, F, gamma, save_mean, save_invstd, dgamma, dbeta, relu, accumulate_params);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
