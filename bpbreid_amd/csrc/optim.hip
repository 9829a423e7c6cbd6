// Fused multi-tensor Adam on the flat parameter arena (torch.optim.Adam semantics with coupled L2 weight
// decay, as the reference configures it: torchreid/optim/optimizer.py:113-119, default_config.py:125-127,153-155).
// One launch updates every parameter that received a gradient this step; `blocks` maps each 1024-element
// block to an arena offset so that parameters without gradient are skipped exactly like torch skips
// `grad is None` ("DDP-specific trap").
#include "bpb_common.h"

__global__ void bpb_incr_kernel(int* __restrict__ x) { x[0] += 1; }

__global__ __launch_bounds__(256) void bpb_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const long* __restrict__ blk_off,
                                                       const int* __restrict__ blk_len, float lr, float beta1, float beta2,
                                                       float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                                       const int* __restrict__ step_dev, const float* __restrict__ lr_dev)
{
    if (lr_dev) lr = lr_dev[0];   // device-resident learning rate: a scheduler step needs no re-capture of a hipGraph
    const long off = blk_off[blockIdx.x];
    const int len = blk_len[blockIdx.x];
    if (step_dev) {   // device-resident step counter: a captured (hipGraph) launch still gets the right bias correction
        const float t = (float)step_dev[0];
        bc1 = 1.f - powf(beta1, t);
        bc2_sqrt = sqrtf(1.f - powf(beta2, t));
    }
    const float step = lr / bc1;
    for (int i = threadIdx.x; i < len; i += 256) {
        const long e = off + i;
        float gr = g[e] * gscale + wd * p[e];
        const float mm = beta1 * m[e] + (1.f - beta1) * gr;
        const float vv = beta2 * v[e] + (1.f - beta2) * gr * gr;
        m[e] = mm;
        v[e] = vv;
        p[e] -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

__global__ __launch_bounds__(256) void bpb_fill_kernel(float* __restrict__ x, float value, long n)
{
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] = value;
}

__global__ void bpb_add_i64_kernel(long* __restrict__ x, long n, long v)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) x[i] += v;
}

__global__ __launch_bounds__(256) void bpb_copy2d_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, int rows,
                                                         int cols)
{
    const long total = (long)rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        const long r = i / cols, c = i - r * cols;
        dst[r * ldd + c] = src[r * lds + c];
    }
}

extern "C" {

// step_index is 1-based.  gscale multiplies the gradient first (1/world_size after a summing all-reduce).
// step_dev (optional): device int32 step counter; it is incremented on the stream first and then used for the bias
// correction, so the same launch sequence can be replayed from a hipGraph.  Otherwise step_index (1-based) is used.
// lr_dev (optional): device float holding the learning rate (overrides `lr`), for the same reason.
int bpb_adam_step(float* p, const float* g, float* m, float* v, const long* blk_off, const int* blk_len, int nblocks,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step_index, float gscale,
                  int* step_dev, const float* lr_dev, hipStream_t stream)
{
    BPB_REQUIRE(nblocks >= 1 && (step_dev || step_index >= 1), "bpb_adam_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step_index);
    const float bc2 = 1.f - powf(beta2, (float)step_index);
    if (step_dev) hipLaunchKernelGGL(bpb_incr_kernel, dim3(1), dim3(1), 0, stream, step_dev);
    hipLaunchKernelGGL(bpb_adam_kernel, dim3(nblocks), dim3(256), 0, stream, p, g, m, v, blk_off, blk_len, lr, beta1, beta2,
                       eps, weight_decay, bc1, sqrtf(bc2), gscale, step_dev, lr_dev);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_fill(float* x, float value, long n, hipStream_t stream)
{
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(bpb_fill_kernel, dim3((int)g), dim3(256), 0, stream, x, value, n);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_add_i64(long* x, long n, long v, hipStream_t stream)
{
    BPB_REQUIRE(x != nullptr && n >= 1, "bpb_add_i64: bad arguments");
    hipLaunchKernelGGL(bpb_add_i64_kernel, dim3(bpb_cdiv(n, 256)), dim3(256), 0, stream, x, n, v);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_copy2d(const float* src, long lds, float* dst, long ldd, int rows, int cols, hipStream_t stream)
{
    BPB_REQUIRE(src != nullptr && dst != nullptr && rows >= 1 && cols >= 1 && lds >= cols && ldd >= cols, "bpb_copy2d: bad arguments");
    long g = ((long)rows * cols + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(bpb_copy2d_kernel, dim3((int)g), dim3(256), 0, stream, src, lds, dst, ldd, rows, cols);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
