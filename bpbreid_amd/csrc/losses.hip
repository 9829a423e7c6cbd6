// GiLt objective kernels: label-smoothed identity cross-entropy, pixel-wise part cross-entropy and the
// part-based batch-hard triplet loss with visibility-masked pairwise distances (forward + gradients).
//
// Replaces torchreid/losses/cross_entropy_loss.py:34-56 (one-hot built on the CPU + H2D copy per call),
// torchreid/losses/part_averaged_triplet_loss.py:35-224 and the part_{max,min,max_min,individual}
// variants, torchreid/utils/tensortools.py:3-21, torchreid/losses/body_part_attention_loss.py:45-52 and
// the target construction of engine/image/part_based_engine.py:114-128.  These are latency-bound
// (a few hundred KB): the goal is few launches and no host synchronisation (the reference's boolean
// indexing of valid triplets, part_averaged_triplet_loss.py:159, forces a device->host sync).
#include "bpb_common.h"

#define BPB_FMAX 3.402823466e+38f

__device__ __forceinline__ float block_sum(float v, float* red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float block_max(float v, float* red)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = blockDim.x >> 1; o >= 1; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// ---- label-smoothed CE over rows ------------------------------------------------------------------
// loss = sum_i w_i * CE_i / max(sum_i |w_i|, 1e-12),  CE_i = -sum_c t_ic log p_ic,  t = (1-eps) onehot + eps/C.
// w == nullptr -> w_i = 1 (plain mean over rows, cross_entropy_loss.py:55).  Boolean visibility filtering
// (GiLt_loss.py:111-113) is w in {0,1}.  row_loss[i] = CE_i, row_ok[i] = (argmax == target); dlogits is
// written UNSCALED by the upstream gradient: dlogits[i][c] = w_i/W * (p_ic - t_ic).
__global__ __launch_bounds__(256) void bpb_ce_rows_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ targets,
                                                          int target_div, const float* __restrict__ w, int R, int C, float eps,
                                                          float* __restrict__ row_loss, float* __restrict__ row_ok,
                                                          float* __restrict__ dlogits, long ldd)
{
    __shared__ float red[256];
    const int i = blockIdx.x;
    const float* l = logits + (long)i * ld;
    const int y = (int)targets[i / target_div];
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, l[c]);
    mx = block_max(mx, red);
    float se = 0.f, sl = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        se += expf(l[c] - mx);
        sl += l[c];
    }
    se = block_sum(se, red);
    sl = block_sum(sl, red);
    const float lse = mx + logf(se);
    // -sum_c t_c (l_c - lse) = lse - (1-eps) l_y - eps/C * sum_c l_c
    if (threadIdx.x == 0) row_loss[i] = lse - (1.f - eps) * l[y] - eps / (float)C * sl;
    // arg-max == target (first maximum wins)
    float best = -INFINITY;
    int bi = C;
    for (int c = threadIdx.x; c < C; c += 256)
        if (l[c] > best) { best = l[c]; bi = c; }
    red[threadIdx.x] = best;
    __shared__ int redi[256];
    redi[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) {
            const float a = red[threadIdx.x], b = red[threadIdx.x + o];
            const int ia = redi[threadIdx.x], ib = redi[threadIdx.x + o];
            if (b > a || (b == a && ib < ia)) { red[threadIdx.x] = b; redi[threadIdx.x] = ib; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) row_ok[i] = (redi[0] == y) ? 1.f : 0.f;
    if (dlogits) {
        float* d = dlogits + (long)i * ldd;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float p = expf(l[c] - lse);
            const float t = (c == y ? 1.f - eps : 0.f) + eps / (float)C;
            d[c] = p - t;   // scaled by w_i / W in the finish kernel
        }
    }
}

// out[0] = loss, out[1] = accuracy over rows with w != 0 (all rows when w == nullptr); scales dlogits rows by w_i/W.
// Every workgroup re-derives the (tiny) row sums in the same fixed order, workgroup 0 writes the scalars, and the R x C scaling
// of dlogits is spread over the grid (one workgroup per 4 rows): a single workgroup walking 320 x 751 elements took 73 us.
#define CE_FIN_ROWS 4
__global__ __launch_bounds__(256) void bpb_ce_finish_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_ok,
                                                            const float* __restrict__ w, int acc_on_selected, int R, int C,
                                                            float* __restrict__ dlogits, long ldd, float* __restrict__ out)
{
    __shared__ float red[256];
    float sw = 0.f, sl = 0.f, sa = 0.f, sn = 0.f;
    for (int i = threadIdx.x; i < R; i += 256) {
        const float wi = w ? w[i] : 1.f;
        sw += fabsf(wi);
        sl += wi * row_loss[i];
        const float sel = (acc_on_selected && w) ? (wi != 0.f ? 1.f : 0.f) : 1.f;
        sa += sel * row_ok[i];
        sn += sel;
    }
    sw = block_sum(sw, red);
    const float W = fmaxf(sw, 1e-12f);
    if (blockIdx.x == 0) {
        sl = block_sum(sl, red);
        sa = block_sum(sa, red);
        sn = block_sum(sn, red);
        if (threadIdx.x == 0) {
            out[0] = sl / W;
            out[1] = sn > 0.f ? sa / sn : 0.f;
        }
    }
    if (dlogits) {
        const int r0 = blockIdx.x * CE_FIN_ROWS, r1 = min(R, r0 + CE_FIN_ROWS);
        for (int i = r0; i < r1; ++i) {
            const float k = (w ? w[i] : 1.f) / W;
            for (int c = threadIdx.x; c < C; c += 256) dlogits[(long)i * ldd + c] *= k;
        }
    }
}

// Gradient of the weighted CE wrt CONTINUOUS row weights (cross_entropy_loss.py:52-54: result * normalize(weights, p=1)):
//   L = sum_i w_i CE_i / S,  S = max(sum_i |w_i|, 1e-12)  ->  dL/dw_j = (CE_j - L sgn(w_j)) / S  (CE_j / 1e-12 under the clamp)
__global__ __launch_bounds__(256) void bpb_ce_weight_grad_kernel(const float* __restrict__ row_loss, const float* __restrict__ w,
                                                                 const float* __restrict__ gloss, int R, float* __restrict__ dw)
{
    __shared__ float red[256];
    float sw = 0.f, sl = 0.f;
    for (int i = threadIdx.x; i < R; i += 256) {
        sw += fabsf(w[i]);
        sl += w[i] * row_loss[i];
    }
    sw = block_sum(sw, red);
    sl = block_sum(sl, red);
    const bool clamped = sw < 1e-12f;
    const float S = clamped ? 1e-12f : sw;
    const float L = sl / S, g = gloss[0];
    for (int i = threadIdx.x; i < R; i += 256) {
        const float sg = w[i] > 0.f ? 1.f : (w[i] < 0.f ? -1.f : 0.f);
        dw[i] = g * (row_loss[i] - (clamped ? 0.f : L * sg)) / S;
    }
}

// ---- pixel-wise part CE (body part attention loss) ------------------------------------------------
// scores [N][K1][HW] (NCHW), external masks [N][K1][Hm][Wm]; target = argmax_k bilinear(align_corners)(masks)
// (part_based_engine.py:118-124); loss = mean over pixels of label-smoothed CE; dscores = (p - t)/(N*HW).
__global__ __launch_bounds__(256) void bpb_pixel_ce_kernel(const float* __restrict__ scores, const float* __restrict__ masks,
                                                           const long* __restrict__ targets, int N, int K1, int H, int W, int Hm, int Wm, float sh, float sw,
                                                           float eps, float* __restrict__ dscores, double* __restrict__ partial)
{
    __shared__ float red[256];
    const int HW = H * W;
    const long total = (long)N * HW;
    float lsum = 0.f, asum = 0.f;
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long n = i / HW;
        const int p = (int)(i - n * HW), h = p / W, w = p - h * W;
        const float fh = sh * h, fw = sw * w;
        const int h0 = (int)fh, w0 = (int)fw;
        const int h1 = h0 + (h0 < Hm - 1), w1 = w0 + (w0 < Wm - 1);
        const float lh1 = fh - h0, lw1 = fw - w0, lh0 = 1.f - lh1, lw0 = 1.f - lw1;
        int y = 0;
        if (targets) {   // the reference engine's call form: int64 [N][H][W] part indices (body_part_attention_loss.py:31-52)
            // a label outside [0, K1) (an ignore index, a mask with more parts than the model) raises in the reference
            // (nn.CrossEntropyLoss); without a host sync the loud equivalent is a NaN loss and NaN gradients for the batch
            const long yt = targets[i];
            bad = bad || yt < 0 || yt >= K1;
            y = (yt < 0 || yt >= K1) ? 0 : (int)yt;
        } else {
            float best = -INFINITY;
            for (int k = 0; k < K1; ++k) {
                const float* m = masks + ((n * K1 + k) * Hm) * (long)Wm;
                const float v = lh0 * (lw0 * m[h0 * Wm + w0] + lw1 * m[h0 * Wm + w1]) + lh1 * (lw0 * m[h1 * Wm + w0] + lw1 * m[h1 * Wm + w1]);
                if (v > best) { best = v; y = k; }
            }
        }
        float l[16], mx = -INFINITY, sl = 0.f;
        int am = 0;
        for (int k = 0; k < K1; ++k) {
            l[k] = scores[(n * K1 + k) * HW + p];
            if (l[k] > mx) { mx = l[k]; am = k; }
            sl += l[k];
        }
        float se = 0.f;
        for (int k = 0; k < K1; ++k) se += expf(l[k] - mx);
        const float lse = mx + logf(se);
        lsum += lse - (1.f - eps) * l[y] - eps / (float)K1 * sl;
        asum += (am == y) ? 1.f : 0.f;
        if (dscores)
            for (int k = 0; k < K1; ++k) {
                const float t = (k == y ? 1.f - eps : 0.f) + eps / (float)K1;
                dscores[(n * K1 + k) * HW + p] = bad ? NAN : (expf(l[k] - lse) - t) / (float)total;
            }
    }
    if (bad) lsum = NAN;
    const float bl = block_sum(lsum, red), ba = block_sum(asum, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2 + 0] = (double)bl;
        partial[blockIdx.x * 2 + 1] = (double)ba;
    }
}

__global__ __launch_bounds__(256) void bpb_pixel_ce_finish_kernel(const double* __restrict__ partial, int nblocks, double total,
                                                                  float* __restrict__ out)
{
    // 256 lanes add the per-block partials b = lane, lane + 256, ... ; the lane sums are combined by a fixed-shape LDS tree
    // (deterministic).  One thread walking <= 1024 partials took 44 us.
    __shared__ double red[2][256];
    double l = 0.0, a = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        l += partial[b * 2];
        a += partial[b * 2 + 1];
    }
    red[0][threadIdx.x] = l;
    red[1][threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) {
            red[0][threadIdx.x] += red[0][threadIdx.x + o];
            red[1][threadIdx.x] += red[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(red[0][0] / total);
        out[1] = (float)(red[1][0] / total);
    }
}

// ---- part-based batch-hard triplet ----------------------------------------------------------------
// (1) per-part pairwise distances, part_averaged_triplet_loss.py:77-93.  emb element (i,k,d) at i*se_n + k*se_k + d.
//     dist[k][i][j] = sqrt(relu(sq_i - 2 a_i.a_j + sq_j) + [==0]*eps) * (1 - [==0])
__global__ __launch_bounds__(256) void bpb_triplet_dist_kernel(const float* __restrict__ emb, long se_n, long se_k, int N, int K,
                                                               int D, float epsilon, float* __restrict__ dist)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // a_i [D], dots [N]
    const int i = blockIdx.x, k = blockIdx.y;
    const float* ai = emb + i * se_n + k * se_k;
    for (int d = threadIdx.x; d < D; d += 256) smem[d] = ai[d];
    __syncthreads();
    float* dots = smem + D;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float* aj = emb + j * se_n + k * se_k;
        float dot = 0.f, sqj = 0.f, sqi = 0.f;
        for (int d = 0; d < D; ++d) {
            const float a = smem[d], b = aj[d];
            dot = fmaf(a, b, dot);
            sqj = fmaf(b, b, sqj);
            sqi = fmaf(a, a, sqi);
        }
        float v = sqi - 2.f * dot + sqj;
        if (j == i) v = 0.f;   // reference: sq_i is the diagonal of the same product, so the diagonal is exactly 0
        v = v > 0.f ? v : 0.f;
        const float zero = v == 0.f ? 1.f : 0.f;
        dots[j] = sqrtf(v + zero * epsilon) * (1.f - zero);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += 256) dist[((long)k * N + i) * N + j] = dots[j];
}

// (2) combine over parts + batch-hard mining + loss + gradient wrt the SQUARED-distance argument.
//     strategy: 0 averaged, 1 max, 2 min, 3 max_min (max for positives / min for negatives), 4 individual.
//     vis: nullptr | float [N][K] (vis_is_bool: entries are 0/1 and the pair mask is the product; otherwise
//     sqrt(v_i v_j), part_averaged_triplet_loss.py:57-59).  drop: optional uint8 [K][N][N] (random_max_min).
//     out[0] loss, out[1] trivial ratio, out[2] valid ratio, out[3] = number of valid triplets.
//     gsq[k][i][j] = d loss / d (squared distance before the sqrt) -- consumed by bpb_triplet_bwd_kernel.
__global__ __launch_bounds__(1024) void bpb_triplet_mine_kernel(const float* __restrict__ dist, const long* __restrict__ pids,
                                                                const float* __restrict__ vis, int vis_is_bool,
                                                                const unsigned char* __restrict__ drop, int N, int K,
                                                                int strategy, float margin, float* __restrict__ pair,
                                                                int* __restrict__ pair_part, float* __restrict__ out,
                                                                float* __restrict__ gsq, float* __restrict__ gvis)
{
    __shared__ float red[1024];
    const int KP = strategy == 4 ? K : 1;   // number of distance matrices that are mined
    // ---- combine
    for (int e = threadIdx.x; e < N * N; e += blockDim.x) {
        const int i = e / N, j = e - i * N;
        const bool same = pids[i] == pids[j];
        if (strategy == 4) {
            for (int k = 0; k < K; ++k) {
                float m = 1.f;
                if (vis) m = vis[i * K + k] * vis[j * K + k];
                pair[((long)k * N + i) * N + j] = (vis && m == 0.f) ? -1.f : dist[((long)k * N + i) * N + j];
            }
            continue;
        }
        float wsum = 0.f, vsum = 0.f, mx = -1.f, mn = BPB_FMAX;
        int imx = 0, imn = 0, nvalid = 0;
        for (int k = 0; k < K; ++k) {
            float m = 1.f;
            if (vis) {
                m = vis[i * K + k] * vis[j * K + k];
                if (!vis_is_bool) m = sqrtf(m);
            }
            if (drop && !drop[((long)k * N + i) * N + j]) m = 0.f;
            const float d = dist[((long)k * N + i) * N + j];
            wsum += m;
            vsum += d * m;
            if (m != 0.f) {
                ++nvalid;
                if (d > mx) { mx = d; imx = k; }
                if (d < mn) { mn = d; imn = k; }
            }
        }
        float v;
        int part = -1;
        const bool masked = (vis != nullptr) || (drop != nullptr);
        if (strategy == 0) {
            v = masked ? (wsum == 0.f ? -1.f : vsum / wsum) : vsum / (float)K;
        } else if (strategy == 1) {
            v = mx; part = imx;                       // no valid part -> stays -1 (replace_values(.., -1).max)
        } else if (strategy == 2) {
            v = nvalid ? mn : (masked ? -1.f : mn); part = imn;
        } else {
            if (same) { v = mx; part = imx; } else { v = mn; part = imn; }
            if (masked && nvalid == 0) v = -1.f;
        }
        pair[e] = v;
        pair_part[e] = part;
    }
    __syncthreads();
    // ---- mining: one thread per (matrix kp, anchor i)
    float hinge_sum = 0.f, trivial = 0.f, valid = 0.f;
    for (int a = threadIdx.x; a < KP * N; a += blockDim.x) {
        const int kp = a / N, i = a - kp * N;
        const float* row = pair + ((long)kp * N + i) * N;
        float dp = -1.f, dn = BPB_FMAX;
        int jp = -1, jn = -1;
        for (int j = 0; j < N; ++j) {
            const float d = row[j];
            const bool ok = d != -1.f;
            const bool same = pids[i] == pids[j];
            const float vp = (ok && same && j != i) ? d : -1.f;
            const float vn = (ok && !same) ? d : BPB_FMAX;
            if (vp > dp) { dp = vp; jp = j; }
            if (vn < dn) { dn = vn; jn = j; }
        }
        const bool okt = (dp != -1.f) && (dn != BPB_FMAX);
        float h = 0.f;
        if (okt) {
            valid += 1.f;
            h = dp - dn + (margin > 0.f ? margin : 0.3f);
            h = h > 0.f ? h : 0.f;
            if (h == 0.f) trivial += 1.f;
            if (margin > 0.f) hinge_sum += h;
            else hinge_sum += log1pf(expf(-(dn - dp)));     // soft margin: log(1 + exp(-(dn - dp)))
        }
        // per-anchor results for the gradient pass live behind the N*N part-id table
        pair_part[N * N + a * 4 + 0] = okt ? jp : -1;
        pair_part[N * N + a * 4 + 1] = okt ? jn : -1;
        ((float*)pair_part)[N * N + a * 4 + 2] = h;
        ((float*)pair_part)[N * N + a * 4 + 3] = okt ? (dn - dp) : 0.f;
    }
    __syncthreads();
    const float V = block_sum(valid, red);
    const float HS = block_sum(hinge_sum, red);
    const float T = block_sum(trivial, red);
    if (threadIdx.x == 0) {
        out[0] = V > 0.f ? HS / V : 0.f;
        out[1] = V > 0.f ? T / V : 0.f;
        out[2] = V / (float)(KP * N);
        out[3] = V;
    }
    if (!gsq) return;
    // ---- gradient wrt squared distances.  d loss / d pair[kp][i][j] = c_i * ([j==jp] - [j==jn])
    for (long e = threadIdx.x; e < (long)K * N * N; e += blockDim.x) gsq[e] = 0.f;
    __syncthreads();
    for (int a = threadIdx.x; a < KP * N; a += blockDim.x) {
        const int kp = a / N, i = a - kp * N;
        const int jp = pair_part[N * N + a * 4 + 0], jn = pair_part[N * N + a * 4 + 1];
        if (jp < 0 || V <= 0.f) continue;
        float c;
        if (margin > 0.f) {
            c = ((float*)pair_part)[N * N + a * 4 + 2] > 0.f ? 1.f / V : 0.f;
        } else {
            const float z = ((float*)pair_part)[N * N + a * 4 + 3];   // dn - dp ; d/d(dp) log(1+e^{-z}) = sigmoid(-z)
            c = (1.f / (1.f + expf(z))) / V;
        }
        if (c == 0.f) continue;
        for (int s = 0; s < 2; ++s) {
            const int j = s == 0 ? jp : jn;
            const float gpair = s == 0 ? c : -c;
            // distribute to parts
            if (strategy == 4) {
                const float d = dist[((long)kp * N + i) * N + j];
                if (d > 0.f) gsq[((long)kp * N + i) * N + j] += gpair / (2.f * d);
            } else if (strategy == 0) {
                float wsum = 0.f;
                const bool masked = (vis != nullptr);
                for (int k = 0; k < K; ++k) {
                    float m = 1.f;
                    if (vis) { m = vis[i * K + k] * vis[j * K + k]; if (!vis_is_bool) m = sqrtf(m); }
                    wsum += m;
                }
                for (int k = 0; k < K; ++k) {
                    float m = 1.f;
                    if (vis) { m = vis[i * K + k] * vis[j * K + k]; if (!vis_is_bool) m = sqrtf(m); }
                    const float wk = masked ? (wsum == 0.f ? 0.f : m / wsum) : 1.f / (float)K;
                    const float d = dist[((long)k * N + i) * N + j];
                    if (d > 0.f && wk != 0.f) gsq[((long)k * N + i) * N + j] += gpair * wk / (2.f * d);
                }
            } else {
                const int k = pair_part[i * N + j];
                if (k >= 0) {
                    const float d = dist[((long)k * N + i) * N + j];
                    if (d > 0.f) gsq[((long)k * N + i) * N + j] += gpair / (2.f * d);
                }
            }
        }
    }
    // ---- gradient wrt CONTINUOUS visibility scores (part-averaged combination only: part_averaged_triplet_loss.py:53-59 builds
    // the pair mask m_k = sqrt(v_ik v_jk), tensortools.py:12-21 the weighted mean D = sum_k d_k m_k / sum_k m_k).  For a mined
    // pair with upstream gradient g:  dL/dm_k = g (d_k - D) / sum m,  dm_k/dv_ik = m_k / (2 v_ik).  One thread per (sample, part)
    // walks all anchors in a fixed order (deterministic, no atomics); d loss / d vis is returned UNSCALED by the upstream gradient.
    if (gvis && strategy == 0 && vis && !vis_is_bool) {
        __syncthreads();
        for (int t = threadIdx.x; t < N * K; t += blockDim.x) {
            const int n = t / K, kk = t - n * K;
            float acc = 0.f;
            for (int i = 0; i < N; ++i) {
                const int jp = pair_part[N * N + i * 4 + 0], jn = pair_part[N * N + i * 4 + 1];
                if (jp < 0 || V <= 0.f) continue;
                float c;
                if (margin > 0.f) {
                    c = ((float*)pair_part)[N * N + i * 4 + 2] > 0.f ? 1.f / V : 0.f;
                } else {
                    const float z = ((float*)pair_part)[N * N + i * 4 + 3];
                    c = (1.f / (1.f + expf(z))) / V;
                }
                if (c == 0.f) continue;
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int j = s2 == 0 ? jp : jn;
                    if (n != i && n != j) continue;
                    const float gpair = s2 == 0 ? c : -c;
                    float wsum = 0.f;
                    for (int k2 = 0; k2 < K; ++k2) wsum += sqrtf(vis[i * K + k2] * vis[j * K + k2]);
                    if (wsum == 0.f) continue;
                    const float m = sqrtf(vis[i * K + kk] * vis[j * K + kk]);
                    const float dm = gpair * (dist[((long)kk * N + i) * N + j] - pair[i * N + j]) / wsum;
                    const float vn = vis[n * K + kk];
                    if (vn > 0.f) acc += dm * m / (2.f * vn) * ((n == i && n == j) ? 2.f : 1.f);
                }
            }
            gvis[t] = acc;
        }
    }
}

// (3) demb[i][k][:] (+)= gscale * 2 * sum_j (gsq[k][i][j] + gsq[k][j][i]) * (a_i - a_j)
__global__ __launch_bounds__(256) void bpb_triplet_bwd_kernel(const float* __restrict__ emb, long se_n, long se_k,
                                                              const float* __restrict__ gsq, const float* __restrict__ gscale,
                                                              float gmul, int N, int K, int D, float* __restrict__ demb,
                                                              long sd_n, long sd_k, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // c_j [N]
    const int i = blockIdx.x, k = blockIdx.y;
    for (int j = threadIdx.x; j < N; j += 256) smem[j] = gsq[((long)k * N + i) * N + j] + gsq[((long)k * N + j) * N + i];
    __syncthreads();
    const float gs = (gscale ? gscale[0] : 1.f) * gmul;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float ai = emb[i * se_n + k * se_k + d];
        float s = 0.f;
        for (int j = 0; j < N; ++j) {
            const float c = smem[j];
            if (c != 0.f) s += c * (ai - emb[j * se_n + k * se_k + d]);
        }
        float* o = demb + i * sd_n + k * sd_k + d;
        const float v = 2.f * gs * s;
        *o = accumulate ? *o + v : v;
    }
}

// y = alpha_dev[0] * alpha * x  (+ y)   -- applies the upstream scalar gradient without a host sync
__global__ __launch_bounds__(256) void bpb_scale_kernel(const float* __restrict__ x, const float* __restrict__ alpha_dev,
                                                        float alpha, float* __restrict__ y, long n, int accumulate)
{
    const float a = (alpha_dev ? alpha_dev[0] : 1.f) * alpha;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) y[i] = accumulate ? y[i] + a * x[i] : a * x[i];
}

struct BpbScalarTerms {
    const float* p[8];
    float w[8];
};
__global__ void bpb_weighted_sum_kernel(BpbScalarTerms t, int n, float* __restrict__ out)
{
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += t.w[i] * t.p[i][0];
        out[0] = s;
    }
}
__global__ void bpb_scalar_fanout_kernel(BpbScalarTerms t, int n, const float* __restrict__ gloss, float* __restrict__ out)
{
    if ((int)threadIdx.x < n) out[threadIdx.x] = gloss[0] * t.w[threadIdx.x];
}

extern "C" {

// scratch: row_loss[R], row_ok[R] floats.  out: [loss, accuracy].
int bpb_ce_label_smooth(const float* logits, long ld, const long* targets, int target_div, const float* w,
                        int acc_on_selected, int R, int C, float eps, float* row_loss, float* row_ok, float* dlogits,
                        long ldd, float* out, hipStream_t stream)
{
    BPB_REQUIRE(R >= 1 && C >= 1 && target_div >= 1, "bpb_ce_label_smooth: bad sizes");
    hipLaunchKernelGGL(bpb_ce_rows_kernel, dim3(R), dim3(256), 0, stream, logits, ld, targets, target_div, w, R, C, eps,
                       row_loss, row_ok, dlogits, ldd);
    hipLaunchKernelGGL(bpb_ce_finish_kernel, dim3(dlogits ? bpb_cdiv(R, CE_FIN_ROWS) : 1), dim3(256), 0, stream, row_loss, row_ok, w, acc_on_selected, R, C,
                       dlogits, ldd, out);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_ce_weight_grad(const float* row_loss, const float* w, const float* gloss, int R, float* dw, hipStream_t stream)
{
    BPB_REQUIRE(R >= 1 && w != nullptr, "bpb_ce_weight_grad: bad arguments");
    hipLaunchKernelGGL(bpb_ce_weight_grad_kernel, dim3(1), dim3(256), 0, stream, row_loss, w, gloss, R, dw);
    BPB_LAUNCH_OK();
    return 0;
}

// partial: nblocks*2 doubles (nblocks <= 1024).  out: [loss, accuracy].
int bpb_pixel_ce(const float* scores, const float* masks, const long* targets, int N, int K1, int H, int W, int Hm, int Wm,
                 float eps, float* dscores, double* partial, int nblocks, float* out, hipStream_t stream)
{
    BPB_REQUIRE(K1 >= 2 && K1 <= 16 && nblocks >= 1 && nblocks <= 1024, "bpb_pixel_ce: bad sizes");
    BPB_REQUIRE((masks != nullptr) != (targets != nullptr), "bpb_pixel_ce: pass either float masks or int64 targets");
    if (targets) { Hm = H; Wm = W; }
    const float sh = H > 1 ? (float)(Hm - 1) / (float)(H - 1) : 0.f;
    const float sw = W > 1 ? (float)(Wm - 1) / (float)(W - 1) : 0.f;
    hipLaunchKernelGGL(bpb_pixel_ce_kernel, dim3(nblocks), dim3(256), 0, stream, scores, masks, targets, N, K1, H, W, Hm, Wm, sh, sw,
                       eps, dscores, partial);
    hipLaunchKernelGGL(bpb_pixel_ce_finish_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, (double)N * H * W, out);
    BPB_LAUNCH_OK();
    return 0;
}

// Forward (+ gradient wrt squared distances when gsq != nullptr; + d loss / d vis [N][K] when gvis != nullptr: continuous
// visibility scores with the part-averaged combination only, zero-filled by the caller otherwise).
// workspace: dist K*N*N floats, pair K*N*N floats, pair_part (N*N + 4*K*N) ints, gsq K*N*N floats.
int bpb_part_triplet(const float* emb, long se_n, long se_k, const long* pids, const float* vis, int vis_is_bool,
                     const unsigned char* drop, int N, int K, int D, int strategy, float margin, float epsilon, float* dist,
                     float* pair, int* pair_part, float* gsq, float* out, float* gvis, hipStream_t stream)
{
    BPB_REQUIRE(N >= 2 && K >= 1 && D >= 1 && strategy >= 0 && strategy <= 4, "bpb_part_triplet: bad arguments");
    BPB_REQUIRE((D + N) * 4 <= 64 * 1024, "bpb_part_triplet: embedding row too large for LDS");
    hipLaunchKernelGGL(bpb_triplet_dist_kernel, dim3(N, K), dim3(256), (D + N) * 4, stream, emb, se_n, se_k, N, K, D, epsilon,
                       dist);
    hipLaunchKernelGGL(bpb_triplet_mine_kernel, dim3(1), dim3(1024), 0, stream, dist, pids, vis, vis_is_bool, drop, N, K,
                       strategy, margin, pair, pair_part, out, gsq, gvis);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_part_triplet_bwd(const float* emb, long se_n, long se_k, const float* gsq, const float* gscale, float gmul, int N,
                         int K, int D, float* demb, long sd_n, long sd_k, int accumulate, hipStream_t stream)
{
    hipLaunchKernelGGL(bpb_triplet_bwd_kernel, dim3(N, K), dim3(256), N * 4, stream, emb, se_n, se_k, gsq, gscale, gmul, N, K,
                       D, demb, sd_n, sd_k, accumulate);
    BPB_LAUNCH_OK();
    return 0;
}

// loss = sum_i w_i * term_i over up to 8 device scalars (GiLt_loss.py:45-76 `loss += weight * term`, part_based_engine.py:126
// `loss += bpa_weight * bpa_loss`): one launch instead of a multiply per term, a stack and a reduction; fixed order.
// fan-out (backward): out[i] = gloss * w_i.
int bpb_weighted_sum(const float* const* h_terms, const float* h_weights, int n, float* out, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 8, "bpb_weighted_sum: %d terms (1..8)", n);
    BpbScalarTerms t;
    for (int i = 0; i < 8; ++i) { t.p[i] = i < n ? h_terms[i] : nullptr; t.w[i] = i < n ? h_weights[i] : 0.f; }
    hipLaunchKernelGGL(bpb_weighted_sum_kernel, dim3(1), dim3(64), 0, stream, t, n, out);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_scalar_fanout(const float* gloss, const float* h_weights, int n, float* out, hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 8, "bpb_scalar_fanout: %d terms (1..8)", n);
    BpbScalarTerms t;
    for (int i = 0; i < 8; ++i) { t.p[i] = nullptr; t.w[i] = i < n ? h_weights[i] : 0.f; }
    hipLaunchKernelGGL(bpb_scalar_fanout_kernel, dim3(1), dim3(64), 0, stream, t, n, gloss, out);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_scale(const float* x, const float* alpha_dev, float alpha, float* y, long n, int accumulate, hipStream_t stream)
{
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(bpb_scale_kernel, dim3((int)g), dim3(256), 0, stream, x, alpha_dev, alpha, y, n, accumulate);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
