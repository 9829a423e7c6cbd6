// Eval-time part-based query x gallery distance with visibility-masked combination.
//
// Replaces torchreid/metrics/distance.py:87-247: per-part Euclidean (no epsilon, relu + sqrt, :230-236) or
// cosine distance [P,Q,G], pair mask q_vis[p,q] * g_vis[p,g] (sqrt for continuous scores, :199), masked mean /
// max over parts with -1 for pairs without a common visible part, then -1 -> max+1 (:171-176, :214-216).
// The reference loops over gallery chunks of 500 with a .cpu() per chunk; here one launch covers a gallery
// shard: a 128x128 (query, gallery) tile (64x64 in the generic fallback kernel) runs the P batched GEMMs back to back
// on the fp32 MFMA pipe and folds each part's distance straight into the masked sum / max in registers.
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// sq[r][p] = sum_d f[r][p][d]^2
__global__ __launch_bounds__(256) void bpb_rownorm_kernel(const float* __restrict__ f, float* __restrict__ sq, long rows, int D)
{
    const long r = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = f[r * D + d]; s = fmaf(v, v, s); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) sq[r] = s;
}

// mode: 0 no visibility, 1 boolean visibility (vis values 0/1), 2 continuous visibility.  strat: 0 mean, 1 max.
// parts_out[p][q][g] : per-part distance (mode 1: -1 where the pair mask is 0); dist_out[q][g]: combined (-1 invalid).
// maxbits: atomicMax over the int bit pattern of every non-negative value written to parts_out (max+1 fill).
__global__ __launch_bounds__(256) void bpb_part_distance_kernel(const float* __restrict__ qf, const float* __restrict__ gf,
                                                                const float* __restrict__ qsq, const float* __restrict__ gsq,
                                                                const float* __restrict__ qvis, const float* __restrict__ gvis,
                                                                int Q, int G, int P, int D, int mode, int strat, int cosine,
                                                                float* __restrict__ parts_out, float* __restrict__ dist_out,
                                                                int* __restrict__ maxbits)
{
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int tiles_g = (G + 63) >> 6;
    const int tq = blockIdx.x / tiles_g, tg = blockIdx.x % tiles_g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int g = tg * 64 + wn + l31;                 // this lane's gallery column
    float sumv[16], sumw[16], mxv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { sumv[r] = 0.f; sumw[r] = 0.f; mxv[r] = -1.f; }
    float lmax = 0.f;
    const long PD = (long)P * D;
    for (int p = 0; p < P; ++p) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < D; k0 += 16) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = threadIdx.x & 15, m = (threadIdx.x >> 4) + 16 * i;
                const int gq = tq * 64 + m, gg = tg * 64 + m, gk = k0 + k;
                As[k][m] = (gq < Q && gk < D) ? qf[gq * PD + (long)p * D + gk] : 0.f;
                Bs[k][m] = (gg < G && gk < D) ? gf[gg * PD + (long)p * D + gk] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) acc = MFMA32(As[kk + half][wm + l31], Bs[kk + half][wn + l31], acc);
        }
        const float gs = (g < G) ? gsq[(long)g * P + p] : 0.f;
        const float gv = (mode != 0 && g < G) ? gvis[(long)g * P + p] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = tq * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (q < Q && g < G) {
                float d;
                if (cosine) d = 1.f - acc[r];
                else {
                    d = qsq[(long)q * P + p] - 2.f * acc[r] + gs;
                    d = sqrtf(d > 0.f ? d : 0.f);
                }
                float m = 1.f;
                if (mode != 0) {
                    m = qvis[(long)q * P + p] * gv;
                    if (mode == 2) m = sqrtf(m);
                }
                float pv = d;
                if (mode == 1 && m == 0.f) pv = -1.f;
                if (parts_out) parts_out[((long)p * Q + q) * G + g] = pv;
                if (pv > lmax) lmax = pv;
                sumv[r] += d * m;
                sumw[r] += m;
                if (m != 0.f && d > mxv[r]) mxv[r] = d;
                if (mode == 0 && strat == 1 && d > mxv[r]) mxv[r] = d;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = tq * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (q < Q && g < G) {
            float v;
            if (mode == 0) v = strat == 1 ? mxv[r] : sumv[r] / (float)P;
            else if (mode == 1 && strat == 1) v = mxv[r];                      // stays -1 when no part is shared
            else v = sumw[r] == 0.f ? -1.f : sumv[r] / sumw[r];                 // masked mean (mode 2 always mean)
            dist_out[(long)q * G + g] = v;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && lmax > 0.f) atomicMax(maxbits, __float_as_int(lmax));
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiled variant (D % 32 == 0, operands < 2 GiB): 128 (query) x 128 (gallery) outputs per workgroup, 4 waves of 64 x 64
// (four 32x32 accumulators each), 32-wide K chunks streamed global -> LDS by `buffer_load ... lds` DMA into a double buffer
// (one barrier per chunk, out-of-range rows zero-filled by the descriptor), operands read as 16-byte fragments with the
// same channel permutation as the convolution kernel ([row][8 data + 1 pad slots]: conflict-free).  Each part's distance
// is folded into ONE register array (masked sum or masked max).  Arithmetic per element is identical to the 64x64 kernel
// above (same fp32 operation order) -- the MFMA k-order differs (8-channel groups), which is round-off only.
//
// Round 3 (profiles/r03_pmc_sq_distance.txt: 6.1 VALU and 0.5 VMEM instructions per MFMA, one workgroup per CU, 0.33 of the
// peak): the per-part epilogue no longer does 64-bit index arithmetic and 128 scattered loads per lane -- the query-side norms
// of the tile are staged in LDS once per part, the visibility of boolean masks is ONE bit mask per row (pair mask = a bit test,
// number of shared parts = a population count), stores address a per-part base pointer with 32-bit lane offsets -- and the
// staging regions are packed (no piece padding) so that two workgroups share a CU.
// Workgroup = 8 waves (512 threads): wave (wq, wg) owns the 32 query rows wq x 64 gallery columns wg of the 128 x 128 tile --
// 64 accumulator + 64 combination registers per lane would not leave room for two workgroups per CU with four 64 x 64 waves.
template <int STRAT>
__global__ __launch_bounds__(512, 2) void bpb_part_distance_tiled_kernel(const float* __restrict__ qf, const float* __restrict__ gf,
                                                                           const float* __restrict__ qsq, const float* __restrict__ gsq,
                                                                           const float* __restrict__ qvis, const float* __restrict__ gvis,
                                                                           int Q, int G, int P, int D, int mode, int cosine,
                                                                           float* __restrict__ parts_out, float* __restrict__ dist_out,
                                                                           int* __restrict__ maxbits, unsigned q_bytes, unsigned g_bytes)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SLOTS = 1152;                       // 128 rows x 9 slots = 2.25 DMA pieces of 512 lanes (the third: waves 0, 1)
    constexpr int BUF = 2 * SLOTS * 16;               // bytes of one {A, B} buffer
    constexpr unsigned OOB = 0x80000000u;
    const int tiles_g = (G + 127) >> 7;
    const int tq = blockIdx.x / tiles_g, tg = blockIdx.x % tiles_g;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wq = wave >> 1, wg = wave & 1;
    const int q0 = tq * 128, g0 = tg * 128;
    const long PD = (long)P * D;
    float* sq_s = (float*)((char*)smem + 2 * BUF);               // [128] squared norms of the tile's query rows, current part
    unsigned* qb_s = (unsigned*)((char*)smem + 2 * BUF + 512);  // [128] visibility bit mask of the tile's query rows (mode 1)

    __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qf, 0, (int)q_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gf, 0, (int)g_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    unsigned aofs[3], bofs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = k * 512 + (int)threadIdx.x;
        const int row = idx / 9, v = idx - row * 9;
        const bool ok = idx < SLOTS && v < 8;
        aofs[k] = (ok && q0 + row < Q) ? (unsigned)(((long)(q0 + row) * PD + v * 4) * 4) : OOB;
        bofs[k] = (ok && g0 + row < G) ? (unsigned)(((long)(g0 + row) * PD + v * 4) * 4) : OOB;
    }
    auto dma_issue = [&](int p, int k0, int buf) {
        const int inc = (p * D + k0) * 4;             // rides in the instruction's scalar offset: the lane offsets are loop invariants
        char* base = (char*)smem + buf * BUF + wave * 1024;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 2 && wave >= 2) continue;        // the third piece holds 128 slots: only waves 0, 1 have lanes in it
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(base + k * 8192), 16, (int)aofs[k], inc, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr_t)(base + SLOTS * 16 + k * 8192), 16, (int)bofs[k], inc, 0, 0);
        }
    };
    const int a_lane = ((wq * 32 + l31) * 9 + half) * 16;          // byte offset of this lane's A fragment (kg = 0)
    const int b_lane = SLOTS * 16 + ((wg * 64 + l31) * 9 + half) * 16;

    // this lane's two gallery columns and the visibility bit masks (boolean visibility: the pair mask of part p is bit p of
    // qbits & gbits, the number of shared visible parts its population count)
    int gcol[2];
    unsigned gbits[2] = {0u, 0u};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        gcol[nt] = g0 + wg * 64 + nt * 32 + l31;
        if (mode == 1 && gcol[nt] < G)
            for (int pp = 0; pp < P; ++pp) gbits[nt] |= (gvis[(long)gcol[nt] * P + pp] != 0.f ? 1u : 0u) << pp;
    }
    if (mode == 1 && threadIdx.x < 128) {
        unsigned b = 0u;
        if (q0 + (int)threadIdx.x < Q)
            for (int pp = 0; pp < P; ++pp) b |= (qvis[(long)(q0 + threadIdx.x) * P + pp] != 0.f ? 1u : 0u) << pp;
        qb_s[threadIdx.x] = b;
    }

    f32x16 comb[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) comb[nt][r] = STRAT == 1 ? -1.f : 0.f;
    float lmax = 0.f;
    const int nch = D >> 5;
    const int nwork = P * nch;
    dma_issue(0, 0, 0);
    f32x16 acc[2];
    int p = 0, ch = 0;
    // The gallery norms of a part are fetched at the part's FIRST chunk, sixteen chunks ahead of their use.  (Round 4 loaded the norm
    // of the second 32-column sub-tile between the two sub-tiles' stores of the per-part matrix: a load waited for behind 32 stores
    // drains them.  Same-box A/B at Q = 2048, G = 20 000, P = 9, profiles/r05_ab_distance.txt: 4.84 -> 4.63 ms with the matrix,
    // 3.96 -> 3.81 ms without.  A counted wait -- s_waitcnt vmcnt(63) + raw s_barrier behind the 64 stores of an interior tile, so
    // that the burst drains under the next chunk's MFMAs -- was measured too: 5.26 / 4.55 ms, the asm waits and the second loop head
    // cost the MFMA loop more than the burst; not kept.)
    float gsv[2] = {0.f, 0.f};
    for (int w = 0; w < nwork; ++w) {
        __syncthreads();                                   // chunk w has landed; the other buffer is free
        if (w + 1 < nwork) {
            const bool lastc = ch + 1 == nch;
            dma_issue(lastc ? p + 1 : p, lastc ? 0 : (ch + 1) * 32, (w + 1) & 1);
        }
        if (ch == 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            if (!cosine && threadIdx.x < 128) sq_s[threadIdx.x] = q0 + (int)threadIdx.x < Q ? qsq[(long)(q0 + threadIdx.x) * P + p] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) gsv[nt] = (gcol[nt] < G && !cosine) ? gsq[(long)gcol[nt] * P + p] : 0.f;
        }
        const char* sb = (const char*)smem + (w & 1) * BUF;
        f32x4 a[2], b[2][2];                                // [ping-pong]([sub-tile])
        auto fetch = [&](int kg, f32x4& af, f32x4 (&bf)[2]) {
            af = *(const f32x4*)(sb + a_lane + kg * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) bf[t] = *(const f32x4*)(sb + b_lane + t * (32 * 9 * 16) + kg * 32);
        };
        auto mma = [&](const f32x4& af, const f32x4 (&bf)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = MFMA32(af[i], bf[nt][i], acc[nt]);
        };
        fetch(0, a[0], b[0]);
        fetch(1, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(2, a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        fetch(3, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[0], b[0]);
        __builtin_amdgcn_sched_barrier(0);
        mma(a[1], b[1]);
        if (++ch == nch) {                                  // ---- this part is complete: distances, mask, fold
            __syncthreads();                                // the norms staged at the part's first chunk (nch may be 1)
            // per-part base of the output block of this tile: the lane offsets below stay 32-bit.  (Round 4 measured the per-part block
            // through an LDS transpose -- 8 sixteen-byte stores per lane and part instead of 32 four-byte ones: 4.86 -> 4.92 ms, no gain:
            // the cost of the [P,Q,G] output is not store issue; stores count in vmcnt on gfx950, so the barrier that waits for the
            // next chunk's DMA also waits for the part's 1.47 GB / P store burst.)
            float* pbase = parts_out ? parts_out + ((long)p * Q + q0) * G : nullptr;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int g = gcol[nt];
                const bool gok = g < G;
                const float gs = gsv[nt];
                const float gv = (mode == 2 && gok) ? gvis[(long)g * P + p] : 1.f;
#pragma unroll
                for (int rq4 = 0; rq4 < 4; ++rq4) {
                    const int rowb = wq * 32 + 8 * rq4 + 4 * half;                 // four consecutive query rows of the tile
                    const f32x4 qs4 = cosine ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(sq_s + rowb);
                    unsigned qb4[4] = {0u, 0u, 0u, 0u};
                    if (mode == 1) {
                        const uint4 t4 = *(const uint4*)(qb_s + rowb);
                        qb4[0] = t4.x; qb4[1] = t4.y; qb4[2] = t4.z; qb4[3] = t4.w;
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int r = rq4 * 4 + jj;
                        const int row = rowb + jj;
                        if (q0 + row < Q && gok) {
                            float d;
                            if (cosine) d = 1.f - acc[nt][r];
                            else {
                                d = qs4[jj] - 2.f * acc[nt][r] + gs;
                                d = sqrtf(d > 0.f ? d : 0.f);
                            }
                            float m = 1.f;
                            if (mode == 1) m = ((qb4[jj] & gbits[nt]) >> p) & 1u ? 1.f : 0.f;
                            else if (mode == 2) m = sqrtf(qvis[(long)(q0 + row) * P + p] * gv);
                            float pv = d;
                            if (mode == 1 && m == 0.f) pv = -1.f;
                            if (pbase) pbase[(unsigned)row * (unsigned)G + (unsigned)g] = pv;
                            if (pv > lmax) lmax = pv;
                            if (STRAT == 1) { if (m != 0.f && d > comb[nt][r]) comb[nt][r] = d; }
                            else comb[nt][r] += d * m;
                        }
                    }
                }
            }
            ch = 0;
            ++p;
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int g = gcol[nt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int q = q0 + row;
            if (q < Q && g < G) {
                float v;
                if (STRAT == 1) v = comb[nt][r];                               // stays -1 when no part is shared
                else if (mode == 0) v = comb[nt][r] / (float)P;
                else {
                    float sw = 0.f;                                             // sum_p m_p
                    if (mode == 1) sw = (float)__popc(qb_s[row] & gbits[nt]);   // (a sum of P zeros and ones: exact in any order)
                    else
                        for (int pp = 0; pp < P; ++pp) sw += sqrtf(qvis[(long)q * P + pp] * gvis[(long)g * P + pp]);   // same order as the P loop
                    v = sw == 0.f ? -1.f : comb[nt][r] / sw;
                }
                dist_out[(long)q * G + g] = v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && lmax > 0.f) atomicMax(maxbits, __float_as_int(lmax));
}

// -1 -> max + 1  (distance.py:171-176 for boolean masks: both matrices; :214-216 for continuous: distmat only)
__global__ __launch_bounds__(256) void bpb_fill_invalid_kernel(float* __restrict__ x, long n, const int* __restrict__ maxbits)
{
    const float fillv = __int_as_float(maxbits[0]) + 1.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
        if (x[i] == -1.f) x[i] = fillv;
}

// Row-wise L2 normalisation of the test embeddings: y = x / max(||x||_2, eps)  (F.normalize(p=2, dim=-1) of
// torchreid/engine/engine.py:558).  One wave per row, lanes strided over the feature dimension, fixed summation order.
__global__ __launch_bounds__(256) void bpb_l2_normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int D,
                                                                    float eps)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d] * xr[d];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.f / fmaxf(sqrtf(s), eps);
    float* yr = y + row * D;
    for (int d = lane; d < D; d += 64) yr[d] = xr[d] * inv;
}

extern "C" {

// scratch: qsq [Q*P], gsq [G*P] floats, maxbits 1 int (zeroed by the callee).  vis arrays are float [rows][P].
// finalize != 0 applies the -1 -> max+1 replacement using the maximum of THIS call; a caller that shards the
// gallery passes finalize = 0, all-reduces (max) `maxbits` and calls bpb_part_distance_fill itself.
// parts_out may be NULL: the [P,Q,G] per-part matrix is then not written (the q-q / g-g calls of the re-ranking and callers
// that only rank need the combined matrix only: 1.47 GB less traffic at Q = 2048, G = 20 000, P = 9).
int bpb_part_distance(const float* qf, const float* gf, const float* qvis, const float* gvis, int Q, int G, int P, int D,
                      int mode, int strat, int cosine, float* qsq, float* gsq, int* maxbits, float* parts_out,
                      float* dist_out, int finalize, hipStream_t stream)
{
    BPB_REQUIRE(Q >= 1 && G >= 1 && P >= 1 && D >= 1, "bpb_part_distance: bad sizes");
    BPB_REQUIRE(mode >= 0 && mode <= 2 && (strat == 0 || strat == 1), "bpb_part_distance: bad mode/strategy");
    BPB_REQUIRE(!(mode == 2 && strat == 1), "bpb_part_distance: continuous visibility supports 'mean' only (distance.py:200)");
    (void)hipMemsetAsync(maxbits, 0, sizeof(int), stream);
    hipLaunchKernelGGL(bpb_rownorm_kernel, dim3(bpb_cdiv((long)Q * P, 4)), dim3(256), 0, stream, qf, qsq, (long)Q * P, D);
    hipLaunchKernelGGL(bpb_rownorm_kernel, dim3(bpb_cdiv((long)G * P, 4)), dim3(256), 0, stream, gf, gsq, (long)G * P, D);
    const double qb = (double)Q * P * D * 4.0, gb = (double)G * P * D * 4.0;
    // (boolean visibility rides as one 32-bit mask per row in the tiled kernel: P <= 32; a block of 128 rows x G of the per-part
    // output is addressed with 32-bit lane offsets)
    if (D % 32 == 0 && qb < 2147483648.0 && gb < 2147483648.0 && (Q >= 128 || G >= 128) && P <= 32 && (double)G * 128.0 * 4.0 < 4294967296.0) {
        static bool attr_done = false;
        const int lds = 2 * 2 * 1152 * 16 + 1024;          // two {A, B} buffers + the staged query norms / visibility bits
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)bpb_part_distance_tiled_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)bpb_part_distance_tiled_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_done = true;
        }
        const int tiles = bpb_cdiv(Q, 128) * bpb_cdiv(G, 128);
        if (strat == 1)
            hipLaunchKernelGGL(bpb_part_distance_tiled_kernel<1>, dim3(tiles), dim3(512), lds, stream, qf, gf, qsq, gsq, qvis, gvis,
                               Q, G, P, D, mode, cosine, parts_out, dist_out, maxbits, (unsigned)qb, (unsigned)gb);
        else
            hipLaunchKernelGGL(bpb_part_distance_tiled_kernel<0>, dim3(tiles), dim3(512), lds, stream, qf, gf, qsq, gsq, qvis, gvis,
                               Q, G, P, D, mode, cosine, parts_out, dist_out, maxbits, (unsigned)qb, (unsigned)gb);
    } else {
        const int tiles = bpb_cdiv(Q, 64) * bpb_cdiv(G, 64);
        hipLaunchKernelGGL(bpb_part_distance_kernel, dim3(tiles), dim3(256), 0, stream, qf, gf, qsq, gsq, qvis, gvis, Q, G, P, D,
                           mode, strat, cosine, parts_out, dist_out, maxbits);
    }
    if (finalize && mode != 0) {
        hipLaunchKernelGGL(bpb_fill_invalid_kernel, dim3(1024), dim3(256), 0, stream, dist_out, (long)Q * G, maxbits);
        if (mode == 1 && parts_out)
            hipLaunchKernelGGL(bpb_fill_invalid_kernel, dim3(2048), dim3(256), 0, stream, parts_out, (long)P * Q * G, maxbits);
    }
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_part_distance_fill(float* x, long n, const int* maxbits, hipStream_t stream)
{
    hipLaunchKernelGGL(bpb_fill_invalid_kernel, dim3(1024), dim3(256), 0, stream, x, n, maxbits);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_l2_normalize_rows(const float* x, float* y, long rows, int D, float eps, hipStream_t stream)
{
    BPB_REQUIRE(rows >= 0 && D >= 1, "bpb_l2_normalize_rows: rows=%ld D=%d", rows, D);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(bpb_l2_normalize_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, y, rows, D, eps);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
