// CMC / mAP of the market1501 protocol on the GPU (torchreid/metrics/rank.py:97-159; rank_cylib/rank_cy.pyx:154-241), for a
// query x gallery distance matrix that is already in HBM (the distance / re-ranking kernels leave it there).
//
// No sort: the protocol only needs the RANKS of the few gallery entries that match the query's identity.  For query q, with
// "kept" = gallery entries that do not share BOTH identity and camera with q (rank.py:122-125) and "match" = kept entries of q's
// identity, the 0-based rank of match m in the kept ranking is
//     r_m = #{ kept i : d_i < d_m  or  (d_i == d_m and i < m) }                     (stable order = lowest gallery index first)
// one workgroup per query counts it for every match; first hit = min r_m -> CMC, AP = (1/M) sum_t t / (r_(t) + 1) over the
// matches in rank order (precision at every relevant position, rank.py:142-147).  Identical numbers to the host routine
// (csrc/rank.cpp, stable_sort on the same keys): integer ranks, AP in fp64 in rank order, fixed-order final sums.
#include "bpb_common.h"

#define RK_MAX_MATCH 2048

__global__ __launch_bounds__(256) void rk_query_kernel(const float* __restrict__ dist, const long* __restrict__ q_pids,
                                                       const long* __restrict__ g_pids, const long* __restrict__ q_cam,
                                                       const long* __restrict__ g_cam, int G, double* __restrict__ ap,
                                                       int* __restrict__ first_rank, int* __restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keptbits[];   // ceil(G / 64) words: entry g is "kept"
    __shared__ int midx[RK_MAX_MATCH];
    __shared__ int mrank[RK_MAX_MATCH];
    __shared__ int red[256];
    __shared__ int nm_s;
    const int q = blockIdx.x;
    const float* row = dist + (size_t)q * G;
    const long qp = q_pids[q], qc = q_cam[q];
    // ---- matches in ascending gallery order (ordered compaction, chunks of 256)
    if (threadIdx.x == 0) nm_s = 0;
    __syncthreads();
    for (int base = 0; base < G; base += 256) {
        const int g = base + threadIdx.x;
        const bool same_pid = g < G && g_pids[g] == qp;
        const bool same_cam = g < G && g_cam[g] == qc;
        const int is_m = (same_pid && !same_cam) ? 1 : 0;
        const unsigned long long kb = __ballot(g < G && !(same_pid && same_cam));
        if ((threadIdx.x & 63) == 0 && g < G) keptbits[g >> 6] = kb;     // (g of lane 0 is a multiple of 64; the table has ceil(G/64) words)
        red[threadIdx.x] = is_m;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int add = (int)threadIdx.x >= o ? red[threadIdx.x - o] : 0;
            __syncthreads();
            red[threadIdx.x] += add;
            __syncthreads();
        }
        const int pos = nm_s + red[threadIdx.x] - is_m;
        if (is_m && pos < RK_MAX_MATCH) midx[pos] = g;
        __syncthreads();
        if (threadIdx.x == 255) nm_s += red[255];
        __syncthreads();
    }
    const int nm_all = nm_s;
    if (nm_all > RK_MAX_MATCH) {
        if (threadIdx.x == 0) { *overflow = 1; ap[q] = -1.0; first_rank[q] = -1; }
        return;
    }
    if (nm_all == 0) {                                   // query identity absent from the gallery (rank.py:131-133)
        if (threadIdx.x == 0) { ap[q] = -1.0; first_rank[q] = -1; }
        return;
    }
    // ---- rank of every match: each thread counts over its strided share of the gallery, block-wide integer sum
    for (int t = 0; t < nm_all; ++t) {
        const int m = midx[t];
        const float dm = row[m];
        int cnt = 0;
        for (int g = threadIdx.x; g < G; g += 256) {
            const float d = row[g];
            const bool kept = (keptbits[g >> 6] >> (g & 63)) & 1ull;
            cnt += (kept && (d < dm || (d == dm && g < m))) ? 1 : 0;
        }
        red[threadIdx.x] = cnt;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) mrank[t] = red[0];
        __syncthreads();
    }
    // ---- AP over the matches in rank order (the ranks are distinct): position of a match = number of matches with a smaller rank
    __shared__ int sorted_rank[RK_MAX_MATCH];
    int fmin = 0x7fffffff;
    for (int t = threadIdx.x; t < nm_all; t += 256) {
        const int r = mrank[t];
        int pos = 0;
        for (int u = 0; u < nm_all; ++u) pos += mrank[u] < r ? 1 : 0;
        sorted_rank[pos] = r;
        fmin = min(fmin, r);
    }
    red[threadIdx.x] = fmin;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double s = 0.0;                                  // hits / kept at every relevant position, accumulated in rank order in
        for (int k = 0; k < nm_all; ++k) s += (double)(k + 1) / (double)(sorted_rank[k] + 1);     // fp64 like the host routine
        ap[q] = s / (double)nm_all;
        first_rank[q] = red[0];
    }
}

// cmc[r] = #{valid q : first_rank[q] <= r} / nvalid ; mAP = mean of the valid APs (fixed order)
__global__ __launch_bounds__(256) void rk_finish_kernel(const double* __restrict__ ap, const int* __restrict__ first_rank, int Q,
                                                        int max_rank, float* __restrict__ cmc, double* __restrict__ map_out,
                                                        int* __restrict__ nvalid_out)
{
    __shared__ int nv;
    if (threadIdx.x == 0) {
        int n = 0;
        double s = 0.0;
        for (int q = 0; q < Q; ++q)
            if (ap[q] >= 0.0) { ++n; s += ap[q]; }
        nv = n;
        *nvalid_out = n;
        *map_out = n > 0 ? s / (double)n : 0.0;
    }
    __syncthreads();
    const int n = nv;
    for (int r = threadIdx.x; r < max_rank; r += 256) {
        int c = 0;
        for (int q = 0; q < Q; ++q) c += (first_rank[q] >= 0 && first_rank[q] <= r) ? 1 : 0;
        cmc[r] = n > 0 ? (float)((double)c / (double)n) : 0.f;
    }
}

extern "C" {

// All pointers are device pointers.  work: Q doubles (AP per query) ; iwork: Q + 2 ints.  Outputs: cmc [max_rank] floats,
// map_out [1] double, nvalid_out = iwork + Q (int), overflow flag = iwork + Q + 1 (a query with more than 2048 matches).
int bpb_eval_rank_gpu(const float* distmat, const long* q_pids, const long* g_pids, const long* q_camids, const long* g_camids,
                      int Q, int G, int max_rank, double* work, int* iwork, float* cmc, double* map_out, hipStream_t stream)
{
    BPB_REQUIRE(Q >= 1 && G >= 1 && max_rank >= 1 && max_rank <= G, "bpb_eval_rank_gpu: bad sizes");
    BPB_REQUIRE(bpb_cdiv(G, 64) * 8 <= 32 * 1024, "bpb_eval_rank_gpu: gallery of %d entries exceeds the LDS bit table (262144)", G);
    (void)hipMemsetAsync(iwork + Q, 0, 2 * sizeof(int), stream);
    hipLaunchKernelGGL(rk_query_kernel, dim3(Q), dim3(256), bpb_cdiv(G, 64) * 8, stream, distmat, q_pids, g_pids, q_camids, g_camids, G, work, iwork,
                       iwork + Q + 1);
    hipLaunchKernelGGL(rk_finish_kernel, dim3(1), dim3(256), 0, stream, work, iwork, Q, max_rank, cmc, map_out, iwork + Q);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
