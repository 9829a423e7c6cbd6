// Row-wise stable argsort of a distance matrix that lives in HBM: the ranked gallery indices of every query
// (torchreid/metrics/rank.py:110 `indices = np.argsort(distmat, axis=1)`; the reference's ranking visualisation consumes the
// index matrix).  CMC / mAP do not need it (csrc/rank_gpu.hip ranks by counting); this serves callers that ask for the indices
// without a 164 MB round trip through the host sort (csrc/rank.cpp: 50 ms at 2048 x 20 000).
//
// One workgroup of 1024 threads sorts one row: LSD radix sort of (order-preserving key bits, gallery index) pairs, four passes of
// 8 bits.  A pass = digit histogram of the row (LDS atomics), exclusive scan of the 256 counts, then a STABLE scatter in chunks of
// 1024 consecutive elements: a lane finds the lanes of its wave with the same digit by eight ballots (its rank among them = the
// population count below it), the sixteen waves of the chunk are ordered through a [wave][digit] count table, and the running
// digit bases advance by the chunk's totals.  Radix sorting is stable and the indices enter in ascending order, so equal distances
// keep the lower gallery index first -- the order of np.argsort(kind='stable') and of csrc/rank.cpp.  (-0.0 sorts before +0.0
// here; numpy treats them as equal.  The distances of this path are sums of squares, clamped at 0, or the max + 1 fill value:
// never -0.0.)  Rows are independent: Q workgroups, two of them resident per CU; the pairs ping-pong between two workspace
// buffers (L2 / HBM), the last pass writes the indices only.  Round 3 called rocPRIM's segmented sort through hipcub here.
#include "bpb_common.h"

constexpr int AS_TPB = 1024, AS_WAVES = AS_TPB / 64, AS_BINS = 256;

__device__ __forceinline__ unsigned as_key_bits(float f)
{
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);       // ascending unsigned order == ascending float order
}

__global__ __launch_bounds__(AS_TPB) void bpb_argsort_rows_kernel(const float* __restrict__ dist, unsigned* __restrict__ keyA, int* __restrict__ valA,
                                                                unsigned* __restrict__ keyB, int* __restrict__ valB, int* __restrict__ out, int G)
{
    __shared__ unsigned base[AS_BINS];                 // histogram -> running exclusive digit bases of the pass
    __shared__ unsigned wcnt[AS_WAVES][AS_BINS];       // per-wave digit counts of the current chunk -> exclusive prefix over the waves
    __shared__ unsigned scan[AS_BINS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row = (size_t)blockIdx.x * G;
    const float* drow = dist + row;
    unsigned* kA = keyA + row;
    unsigned* kB = keyB + row;
    int* vA = valA + row;
    int* vB = valB + row;
    int* orow = out + row;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        const unsigned* ksrc = (pass & 1) ? kA : kB;       // pass 0 reads dist; 1: A -> B; 2: B -> A; 3: A -> out
        const int* vsrc = (pass & 1) ? vA : vB;
        unsigned* kdst = (pass & 1) ? kB : kA;
        int* vdst = (pass & 1) ? vB : vA;
        auto load_key = [&](int i) { return pass == 0 ? as_key_bits(drow[i]) : ksrc[i]; };
        // ---- 1. digit histogram of the row
        if (tid < AS_BINS) base[tid] = 0;
        __syncthreads();
        for (int i = tid; i < G; i += AS_TPB) atomicAdd(&base[(load_key(i) >> shift) & 255u], 1u);
        __syncthreads();
        // ---- 2. exclusive scan of the 256 counts (Hillis-Steele in LDS)
        unsigned mine = tid < AS_BINS ? base[tid] : 0;
        if (tid < AS_BINS) scan[tid] = mine;
        __syncthreads();
        for (int off = 1; off < AS_BINS; off <<= 1) {
            unsigned add = 0;
            if (tid < AS_BINS && tid >= off) add = scan[tid - off];
            __syncthreads();
            if (tid < AS_BINS) scan[tid] += add;
            __syncthreads();
        }
        if (tid < AS_BINS) base[tid] = scan[tid] - mine;
        __syncthreads();
        // ---- 3. stable scatter, 1024 consecutive elements at a time
        for (int c0 = 0; c0 < G; c0 += AS_TPB) {
            const int i = c0 + tid;
            const bool valid = i < G;
            unsigned key = 0;
            int val = 0;
            if (valid) {
                key = load_key(i);
                val = pass == 0 ? i : vsrc[i];
            }
            const unsigned d = (key >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bb = __ballot(bit);
                peers &= bit ? bb : ~bb;
            }
            const unsigned rank_in_wave = (unsigned)__popcll(peers & below);
            for (int j = tid; j < AS_WAVES * AS_BINS; j += AS_TPB) (&wcnt[0][0])[j] = 0;
            __syncthreads();
            if (valid && rank_in_wave == 0) wcnt[wave][d] = (unsigned)__popcll(peers);
            __syncthreads();
            unsigned total = 0;
            if (tid < AS_BINS) {
#pragma unroll
                for (int w = 0; w < AS_WAVES; ++w) {
                    const unsigned t = wcnt[w][tid];
                    wcnt[w][tid] = total;
                    total += t;
                }
            }
            __syncthreads();
            unsigned pos = 0;
            if (valid) pos = base[d] + wcnt[wave][d] + rank_in_wave;
            __syncthreads();
            if (tid < AS_BINS) base[tid] += total;
            if (valid) {
                if (pass == 3) orow[pos] = val;
                else {
                    kdst[pos] = key;
                    vdst[pos] = val;
                }
            }
            __syncthreads();
        }
        // (a pass reads what the previous one wrote through global memory: same workgroup, made visible by the barrier + the
        //  workgroup-scope release / acquire of __syncthreads; no other workgroup touches this row)
        __threadfence_block();
    }
}

static inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" {

// bytes of caller-provided device workspace for bpb_argsort_rows_gpu: two (key, index) pair buffers
int bpb_argsort_rows_gpu_workspace(int Q, int G, long* bytes_out)
{
    BPB_REQUIRE(Q >= 1 && G >= 1 && (double)Q * G < 2147483648.0 && bytes_out, "bpb_argsort_rows_gpu: Q=%d G=%d", Q, G);
    const size_t n = (size_t)Q * G;
    *bytes_out = (long)(4 * up256(n * 4));
    return 0;
}

// idx_out[q][r] = gallery index of rank r of query q (int32, [Q][G]); dist is not modified
int bpb_argsort_rows_gpu(const float* dist, int Q, int G, int* idx_out, void* ws, long ws_bytes, hipStream_t stream)
{
    long need = 0;
    if (int rc = bpb_argsort_rows_gpu_workspace(Q, G, &need)) return rc;
    BPB_REQUIRE(ws != nullptr && ws_bytes >= need, "bpb_argsort_rows_gpu: workspace of %ld bytes, %ld needed", ws_bytes, need);
    const size_t n = (size_t)Q * G, seg = up256(n * 4);
    char* p = (char*)ws;
    hipLaunchKernelGGL(bpb_argsort_rows_kernel, dim3(Q), dim3(AS_TPB), 0, stream, dist, (unsigned*)p, (int*)(p + seg), (unsigned*)(p + 2 * seg),
                       (int*)(p + 3 * seg), idx_out, G);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
