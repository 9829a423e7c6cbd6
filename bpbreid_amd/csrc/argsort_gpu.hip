// Row-wise stable argsort of a distance matrix that lives in HBM: the ranked gallery indices of every query
// (torchreid/metrics/rank.py:110 `indices = np.argsort(distmat, axis=1)`; the reference's ranking visualisation consumes the
// index matrix).  CMC / mAP do not need it (csrc/rank_gpu.hip ranks by counting); this serves callers that ask for the indices
// without a 164 MB round trip through the host sort (csrc/rank.cpp: 50 ms at 2048 x 20 000).
// One segmented LSD radix sort over the Q rows (rocPRIM through hipcub: (key, value) = (distance, gallery index), 32 key bits):
// radix sorting is stable, the values enter in ascending order, so equal distances keep the lower gallery index first -- the
// order of np.argsort(kind='stable') and of csrc/rank.cpp.  (-0.0 sorts before +0.0 here; numpy treats them as equal.  The
// distances of this path are sums of squares, clamped at 0, or the max + 1 fill value: never -0.0.)
#include "bpb_common.h"
#include <hipcub/hipcub.hpp>

__global__ __launch_bounds__(256) void bpb_argsort_init_kernel(int* __restrict__ vals, int* __restrict__ offsets, long total, int Q, int G)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) vals[i] = (int)(i % G);
    if (i <= Q) offsets[i] = (int)(i * G);
}

static size_t argsort_temp_bytes(int Q, int G)
{
    size_t temp = 0;
    const float* kin = nullptr;
    float* kout = nullptr;
    const int* vin = nullptr;
    int* vout = nullptr;
    const int* off = nullptr;
    (void)hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, temp, kin, kout, vin, vout, (int)((long)Q * G), Q, off, off + 1, 0, 32, nullptr);
    return temp;
}

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" {

// bytes of caller-provided device workspace for bpb_argsort_rows_gpu
int bpb_argsort_rows_gpu_workspace(int Q, int G, long* bytes_out)
{
    BPB_REQUIRE(Q >= 1 && G >= 1 && (double)Q * G < 2147483648.0 && bytes_out, "bpb_argsort_rows_gpu: Q=%d G=%d", Q, G);
    const size_t n = (size_t)Q * G;
    *bytes_out = (long)(up256(n * 4) + up256(n * 4) + up256(((size_t)Q + 1) * 4) + up256(argsort_temp_bytes(Q, G)));
    return 0;
}

// idx_out[q][r] = gallery index of rank r of query q (int32, [Q][G]); dist is not modified
int bpb_argsort_rows_gpu(const float* dist, int Q, int G, int* idx_out, void* ws, long ws_bytes, hipStream_t stream)
{
    long need = 0;
    if (int rc = bpb_argsort_rows_gpu_workspace(Q, G, &need)) return rc;
    BPB_REQUIRE(ws != nullptr && ws_bytes >= need, "bpb_argsort_rows_gpu: workspace of %ld bytes, %ld needed", ws_bytes, need);
    const size_t n = (size_t)Q * G;
    char* p = (char*)ws;
    float* keys_out = (float*)p;
    p += up256(n * 4);
    int* vals_in = (int*)p;
    p += up256(n * 4);
    int* offsets = (int*)p;
    p += up256(((size_t)Q + 1) * 4);
    size_t temp = argsort_temp_bytes(Q, G);
    hipLaunchKernelGGL(bpb_argsort_init_kernel, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, stream, vals_in, offsets, (long)n, Q, G);
    const hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairs((void*)p, temp, dist, keys_out, (const int*)vals_in, idx_out, (int)n, Q,
                                                                     (const int*)offsets, (const int*)offsets + 1, 0, 32, stream);
    if (e != hipSuccess) return bpb_set_error((int)e, "bpb_argsort_rows_gpu: %s", hipGetErrorString(e));
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
