// Input side of the path: raw human-parsing confidence maps -> the (K+1)-channel soft part masks the model consumes.
//
// Replaces the per-sample CPU transforms torchreid/data/masks_transforms/mask_transform.py:20-85 as chained by
// torchreid/data/transforms.py:133-158:  channel grouping (max or clamped sum over the source channels of each part,
// MaskGroupingTransform :31-38)  ->  background channel (AddBackgroundMask :58-75: 'sum', 'threshold', 'diff_from_max')
// ->  soft-max(weight * masks) over the K+1 channels or normalisation by their sum (:76-79)  ->  nearest resize to
// (H/scale, W/scale) (ResizeMasks :45-52).  Every step before the resize is per-pixel, so only the pixels the nearest
// resize samples are computed: one pass, 1/scale^2 of the input is read, one [N][K+1][Ho][Wo] tensor is written.
#include "bpb_common.h"

#define BPB_MASK_MAXK 64

__global__ __launch_bounds__(256) void bpb_mask_preprocess_kernel(const float* __restrict__ raw, const int* __restrict__ goff,
                                                                  const int* __restrict__ gch, int N, int Cin, int H, int W,
                                                                  int K, int Ho, int Wo, float sh, float sw, int combine_sum,
                                                                  int bg, float softmax_weight, float threshold,
                                                                  float* __restrict__ out)
{
    const long total = (long)N * Ho * Wo;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const int wo = (int)(i % Wo);
        const long r = i / Wo;
        const int ho = (int)(r % Ho), n = (int)(r / Ho);
        // torch 'nearest': src = min(floor(dst * in/out), in - 1), the scale held in fp32
        const int hs = min((int)floorf(ho * sh), H - 1), ws = min((int)floorf(wo * sw), W - 1);
        const float* px = raw + ((long)n * Cin * H + hs) * W + ws;
        float m[BPB_MASK_MAXK + 1];
        float mx = 0.f, sum = 0.f;
        for (int k = 0; k < K; ++k) {
            float v;
            if (goff) {
                const int b = goff[k], e = goff[k + 1];
                v = combine_sum ? 0.f : -INFINITY;
                for (int j = b; j < e; ++j) {
                    const float x = px[(long)gch[j] * H * W];
                    v = combine_sum ? v + x : fmaxf(v, x);
                }
                v = fminf(fmaxf(v, 0.f), 1.f);
            } else {
                v = px[(long)k * H * W];
            }
            m[k + 1] = v;
            mx = k == 0 ? v : fmaxf(mx, v);
            sum += v;
        }
        float b0;
        if (bg == 0) b0 = fminf(fmaxf(1.f - sum, 0.f), 1.f);
        else if (bg == 1) b0 = mx < threshold ? 1.f : 0.f;
        else b0 = fminf(fmaxf(1.f - mx, 0.f), 1.f);
        m[0] = b0;
        if (softmax_weight > 0.f) {
            float top = -INFINITY;
            for (int k = 0; k <= K; ++k) { m[k] *= softmax_weight; top = fmaxf(top, m[k]); }
            float z = 0.f;
            for (int k = 0; k <= K; ++k) { m[k] = expf(m[k] - top); z += m[k]; }
            for (int k = 0; k <= K; ++k) m[k] /= z;
        } else {
            float z = 0.f;
            for (int k = 0; k <= K; ++k) z += m[k];
            for (int k = 0; k <= K; ++k) m[k] /= z;       // 0/0 = NaN exactly like the reference's masks / masks.sum(dim=0)
        }
        for (int k = 0; k <= K; ++k) out[(((long)n * (K + 1) + k) * Ho + ho) * Wo + wo] = m[k];
    }
}

extern "C" int bpb_mask_preprocess(const float* raw, const int* group_offsets, const int* group_channels, int N, int Cin, int H,
                                   int W, int K, int Ho, int Wo, int combine_sum, int bg_strategy, float softmax_weight,
                                   float threshold, float* out, hipStream_t stream)
{
    BPB_REQUIRE(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "bpb_mask_preprocess: bad sizes");
    BPB_REQUIRE(K >= 1 && K <= BPB_MASK_MAXK, "bpb_mask_preprocess: K=%d out of range [1,%d]", K, BPB_MASK_MAXK);
    BPB_REQUIRE(group_offsets != nullptr || K == Cin, "bpb_mask_preprocess: without a grouping K must equal the channel count");
    BPB_REQUIRE((group_offsets == nullptr) == (group_channels == nullptr), "bpb_mask_preprocess: grouping needs both tables");
    BPB_REQUIRE(bg_strategy >= 0 && bg_strategy <= 2, "bpb_mask_preprocess: background strategy %d", bg_strategy);
    const long total = (long)N * Ho * Wo;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(bpb_mask_preprocess_kernel, dim3(grid), dim3(256), 0, stream, raw, group_offsets, group_channels, N, Cin, H,
                       W, K, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo, combine_sum, bg_strategy, softmax_weight,
                       threshold, out);
    BPB_LAUNCH_OK();
    return 0;
}
