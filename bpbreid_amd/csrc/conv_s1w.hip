// Data gradient of STRIDE-2 3x3 convolutions on the lean kernel family: the parity classes as windowed stride-1 problems.
//   torchreid/models/hrnet.py:240-250 (exchange down paths), :459-481 (transitions), :319-323 (stem conv2);
//   torchreid/models/resnet.py:31-49 (the strided 3x3 of a Bottleneck) -- what autograd runs as conv backward-input.
//
// y = conv3x3(x, stride 2, pad 1):  dx[i][j] = sum_{r, s} dy[(i + 1 - r) / 2][(j + 1 - s) / 2] . W[r][s]^T over the (r, s) that make the
// divisions exact.  For an input row of even index only r = 1 contributes (dy row i / 2); for an odd one r = 2 and r = 0 (dy rows
// (i - 1) / 2 and (i + 1) / 2): a 1- or 2-tap window of dy per direction.  The four (row, column) parity classes are therefore four
// dense stride-1 problems on dy with windows 1x1, 1x2, 2x1, 2x2 whose outputs interleave in dx:
//     dx[n][2a + ph][2b + pw][:] (+)= sum_{u < RH, v < RW} dy[n][a + u][b + v][:] . W_{wt[u * RW + v]}
// The general kernel (conv_igemm.hip) ran the classes as four problems at 47 TFLOP/s (run-time tap iterator, 75-dword descriptor);
// this one is bpb_conv_s1_kernel's structure -- DMA-staged dy and weight tiles, double buffer, k-loop fully unrolled, two-level sums --
// with ONE workgroup computing all four classes of its 128 class pixels from one staged (TH + 1) x (TW + 1) tile of dy: nine taps per
// staged pixel like a 3x3 forward convolution, four accumulator sets per wave.
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define M24(a, b) __umul24((unsigned)(a), (unsigned)(b))

__device__ __forceinline__ unsigned s1w_fdiv(unsigned x, unsigned d, unsigned magic)
{
    return d == 1 ? x : __umulhi(x, magic);
}

// Window position w = u * 2 + v of dy (rows a + u, columns b + v), filter tap r * 3 + s and parity class ph * 2 + pw of the nine
// (window, tap) products of a class pixel:  r = ph + 1 - 2u, s = pw + 1 - 2v.
__device__ constexpr int S1W_WIN[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
__device__ constexpr int S1W_TAP[9] = {4, 5, 7, 8, 3, 6, 1, 2, 0};
__device__ constexpr int S1W_CLS[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};

// One workgroup = 4 waves x 32 class pixels (a, b) x 32 output channels x the FOUR parity classes: the (TH + 1) x (TW + 1) tile of dy
// is staged once and serves all nine taps (round 4, first form: one launch problem per class, each staging its own copy of the
// tile -- 1 to 4 taps per staged pixel, 71-76 TFLOP/s).  Four accumulator sets per wave, like the 64 x 64 wave tiles of conv_s1.
template <int KG>
__global__ __launch_bounds__(256, 2) void bpb_conv_s1w_kernel(const BpbConvS1wProb* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int T = 9, CK = 8 * KG, NS = T * KG;       // NS steps per chunk: one B fragment (tap, k-group) each
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbConvS1wProb P = probs[pi];
    bid -= P.blk_begin;
    if (P.xr) {   // XCD-aware tile map (conv_s1.hip): the blocks of one XCD walk a contiguous range of this problem's tiles
        const int nb = P.n_mtiles * P.n_ntiles, q = nb >> 3, r = nb & 7, f = bid & 7;
        bid = f * q + min(f, r) + (bid >> 3);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int lTW = P.lTW, lTH = P.lTH;
    const int TWm = (1 << lTW) - 1, THm = (1 << lTH) - 1;
    const int mtile = (int)s1w_fdiv((unsigned)bid, P.n_ntiles, P.magic_nt);
    const int ntile = bid - mtile * P.n_ntiles;
    const int t2 = (int)s1w_fdiv((unsigned)mtile, P.tiles_b, P.magic_tb);
    const int tb = mtile - t2 * P.tiles_b;
    const int tn = (int)s1w_fdiv((unsigned)t2, P.tiles_a, P.magic_ta);
    const int ta = t2 - tn * P.tiles_a;
    const int n0 = tn << P.lTI, a0 = ta << lTH, b0 = tb << lTW;
    const int LD = P.LD, HWd = P.HW, HH = P.HH;
    const int Cin = P.Cin, Cout = P.Cout, cin4 = Cin >> 2;

    int pixoff;   // byte offset of this lane's pixel inside the staged tile, + the k half
    {
        const int m = wave * 32 + l31;
        const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
        pixoff = (int)M24(M24(M24(ti, HH) + th, HWd) + tw, LD) * 4 + half * 16;
    }
    const int cout_l = ntile * 32 + l31;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    constexpr int lvpp = KG == 1 ? 1 : 2;                     // log2(CK / 4)
    constexpr int qn = CK >> 2;
    const int npix = (1 << P.lTI) * HH * HWd;
    const int nch = Cin / CK;
    const int spp = LD >> 2;
    const int halo_slots = npix * spp;
    const int halo_reg = (halo_slots + 3) & ~3;
    constexpr int nB = T * qn * 32;
    const int bufbytes = (halo_reg + nB) * 16;

    // ---- DMA piece offsets (pixels beyond dy: an out-of-range offset that the buffer descriptor zero-fills)
    constexpr unsigned DMA_OOB = 0x80000000u;
    constexpr int DMA_HS = 8, DMA_WS = (nB + 255) >> 8;
    const int nhs = (halo_slots + 255) >> 8;
    const bool hlast = (nhs - 1) * 256 + (int)threadIdx.x < halo_slots;
    const bool wlast = (DMA_WS - 1) * 256 + (int)threadIdx.x < nB;
    unsigned hofs[DMA_HS], wofs[DMA_WS];
#pragma unroll
    for (int k = 0; k < DMA_HS; ++k) {
        unsigned vo = DMA_OOB;
        if (k < nhs) {
            const int idx = k * 256 + (int)threadIdx.x;
            const unsigned hp = s1w_fdiv((unsigned)idx, spp, P.magic_spp);
            const int v = idx - (int)M24(hp, spp);
            const unsigned t = s1w_fdiv(hp, HWd, P.magic_hw);
            const int hc = hp - M24(t, HWd);
            const unsigned ti = s1w_fdiv(t, HH, P.magic_hh);
            const int hr = t - M24(ti, HH);
            const int n = n0 + (int)ti, ih = a0 + hr, iw = b0 + hc;           // the windows start AT the class pixel: no padding
            if (idx < halo_slots && v < qn && n < P.N && ih < P.Hi && iw < P.Wi)
                vo = ((M24(M24(n, P.Hi) + ih, P.Wi) + iw) * (unsigned)Cin + v * 4) * 4u;
        }
        hofs[k] = vo;
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < DMA_WS; ++k) {
        unsigned vo = DMA_OOB;
        const int bi = k * 256 + (int)threadIdx.x;
        const int n = bi & 31;
        const int r = bi >> 5;
        const int q = r & (qn - 1);
        const int t = r >> lvpp;                                               // the tile holds the nine taps in filter order
        if (bi < nB) {
            const int co = min(ntile * 32 + n, Cout - 1);                      // columns >= Cout are never stored
            vo = (((unsigned)(t * cin4 + q) * Cout + co) * 4) * 4u;
        }
        wofs[k] = vo;
        __builtin_amdgcn_sched_barrier(0);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_issue = [&](int cb, int buf) {
        char* base = (char*)smem + buf * bufbytes + wave * 1024;
        const unsigned incx = (unsigned)(cb * 4);
#pragma unroll
        for (int k = 0; k < DMA_HS; ++k)
            if (k < nhs && (k + 1 < nhs || hlast))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + k * 4096), 16, (int)(hofs[k] + incx), 0, 0, 0);
        char* wb = base + halo_reg * 16;
        const unsigned incw = (unsigned)((cb >> 2) * Cout * 16);
#pragma unroll
        for (int k = 0; k < DMA_WS; ++k)
            if (k + 1 < DMA_WS || wlast)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(wb + k * 4096), 16, (int)(wofs[k] + incw), 0, 0, 0);
    };

    // ---- channel-chunk loop: DMA of chunk c + 1 under the MFMAs of chunk c, one barrier per chunk.  Per k-group the four window
    // fragments of dy (A) are read once and meet nine weight fragments (B): 36 MFMAs per 13 ds_read_b128, all addresses one VGPR + an
    // immediate.  Two-level sums as in conv_s1 (the chunk's products accumulate into `cacc`).
    int apix[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) apix[w] = pixoff + ((w >> 1) * HWd + (w & 1)) * LD * 4;
    int bptr = halo_reg * 16 + half * 512 + l31 * 16;            // + tap * (qn * 512) + k-group * 1024: immediates
    dma_issue(0, 0);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        if (c + 1 < nch) dma_issue((c + 1) * CK, (c + 1) & 1);
        const char* lds = (const char*)smem;
        f32x16 cacc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[q][r] = 0.f;
        f32x4 fa[2][4], fb[2];
#pragma unroll
        for (int w = 0; w < 4; ++w) fa[0][w] = *(const f32x4*)(lds + apix[w]);
        fb[0] = *(const f32x4*)(lds + bptr + S1W_TAP[0] * (qn * 512));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kg = s / T, it = s % T;
            if (s + 1 < NS) {        // the next step's B fragment, and at the end of a k-group the next group's A fragments
                const int kg1 = (s + 1) / T, it1 = (s + 1) % T;
                fb[(s + 1) & 1] = *(const f32x4*)(lds + bptr + S1W_TAP[it1] * (qn * 512) + kg1 * 1024);
                if (it == T - 1) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) fa[kg1 & 1][w] = *(const f32x4*)(lds + apix[w] + kg1 * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) cacc[S1W_CLS[it]] = MFMA32(fa[kg & 1][S1W_WIN[it]][i], fb[s & 1][i], cacc[S1W_CLS[it]]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int delta = (c & 1) ? -bufbytes : bufbytes;
#pragma unroll
        for (int w = 0; w < 4; ++w) apix[w] += delta;
        bptr += delta;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] += cacc[q][r];
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: column = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // Class (ph, pw) of class pixel (a, b) lands at dx[n][2a + ph][2b + pw]; invalid pixels / channels become out-of-range offsets.
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, (int)P.y_bytes, 0x00020000);
    constexpr unsigned PIX_OOB = 0x80000000u;
    const bool cv = cout_l < Cout;
    const int pstride = Cout * 4;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        unsigned offs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
            const int n = n0 + ti, i = 2 * (a0 + th) + ph, j = 2 * (b0 + tw) + pw;
            const bool pv = cv && (n < P.N) && (i < P.H) && (j < P.W);
            offs[r] = pv ? M24(M24(M24(n, P.H) + i, P.W) + j, pstride) + (unsigned)(cout_l * 4) : PIX_OOB;
        }
        if (P.accumulate) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (int)offs[r], 0, 0));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cls][r] += old[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[cls][r];      // (a scalar copy: __builtin_bit_cast applied to the vector ELEMENT expression reads element 0 for every r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)offs[r], 0, 0);
        }
    }
}

static int conv_s1w_lds_bytes(const BpbConvS1wProb& p)
{
    const int npix = (1 << p.lTI) * p.HH * p.HW;
    const int halo_reg = (npix * (p.LD / 4) + 3) & ~3;
    const int nB = 9 * (p.CK / 4) * 32;
    return 2 * (halo_reg + nB) * 16;
}

extern "C" {

int bpb_conv_s1w_init(void)
{
#define BPB_ATTR(K)                                                                                                  \
    {                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        if (e != hipSuccess) return bpb_set_error((int)e, "bpb_conv_s1w_init: %s", hipGetErrorString(e));             \
    }
    BPB_ATTR((bpb_conv_s1w_kernel<1>)) BPB_ATTR((bpb_conv_s1w_kernel<2>))
#undef BPB_ATTR
    return 0;
}

// Grouped launch: one problem per strided convolution (descriptors in device memory, `h_probs` = host copy for validation); all
// problems of a launch share the channel chunk CK.  Replaces conv backward-input of stride-2 3x3 pad-1 convolutions.
int bpb_conv_s1w(const BpbConvS1wProb* d_probs, const BpbConvS1wProb* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_s1w: nprobs=%d out of range", nprobs);
    int nblk = 0, lds = 0;
    const int ck = h_probs[0].CK;
    for (int i = 0; i < nprobs; ++i) {
        const BpbConvS1wProb& p = h_probs[i];
        BPB_REQUIRE(p.CK == ck && (ck == 8 || ck == 16) && p.Cin % ck == 0 && p.LD == ck + 4, "bpb_conv_s1w: channel chunk CK=%d (LD=%d) for Cin=%d", p.CK,
                    p.LD, p.Cin);
        BPB_REQUIRE(p.Cout % 4 == 0, "bpb_conv_s1w: Cout=%d must be a multiple of 4", p.Cout);
        BPB_REQUIRE((1 << (p.lTI + p.lTH + p.lTW)) == 128, "bpb_conv_s1w: the M tile must hold 128 class pixels");
        BPB_REQUIRE(p.HH == (1 << p.lTH) + 1 && p.HW == (1 << p.lTW) + 1, "bpb_conv_s1w: staged extent mismatch");
        BPB_REQUIRE(p.A == (p.H + 1) / 2 && p.B == (p.W + 1) / 2 && p.A >= 1 && p.B >= 1, "bpb_conv_s1w: class domain %dx%d does not follow from the output %dx%d",
                    p.A, p.B, p.H, p.W);
        // (window rows / columns beyond dy read zero: the last odd row of an even-height input has no (i + 1) / 2 partner)
        BPB_REQUIRE(p.Hi == (p.H - 1) / 2 + 1 && p.Wi == (p.W - 1) / 2 + 1, "bpb_conv_s1w: dy %dx%d is not the stride-2 output of a %dx%d input", p.Hi,
                    p.Wi, p.H, p.W);
        BPB_REQUIRE(p.x_bytes > 0 && p.w_bytes > 0 && p.y_bytes > 0 && p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u && p.y_bytes < 0x80000000u,
                    "bpb_conv_s1w: tensors addressed through a buffer descriptor must be < 2 GiB");
        BPB_REQUIRE((double)p.N * p.H * p.W < 16777216.0 && (double)p.N * p.Hi * p.Wi < 16777216.0 && p.Cout * 4 < 16777216,
                    "bpb_conv_s1w: 24-bit index arithmetic overflow");
        BPB_REQUIRE(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "bpb_conv_s1w: x/w must be 16-byte aligned");
        BPB_REQUIRE(p.tiles_a == bpb_cdiv(p.A, 1 << p.lTH) && p.tiles_b == bpb_cdiv(p.B, 1 << p.lTW) &&
                        p.n_mtiles == bpb_cdiv(p.N, 1 << p.lTI) * p.tiles_a * p.tiles_b && p.n_ntiles == bpb_cdiv(p.Cout, 32),
                    "bpb_conv_s1w: tile counts mismatch");
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_s1w: blk_begin mismatch");
        const int npix = (1 << p.lTI) * p.HH * p.HW;
        BPB_REQUIRE(((npix * (p.LD / 4) + 255) >> 8) <= 8, "bpb_conv_s1w: more than 8 DMA pieces per thread for the staged tile (%d pixels)", npix);
        nblk += p.n_mtiles * p.n_ntiles;
        const int l = conv_s1w_lds_bytes(p);
        lds = l > lds ? l : lds;
    }
    BPB_REQUIRE(lds <= 160 * 1024, "bpb_conv_s1w: needs %d B of LDS", lds);
    if (nblk == 0) return 0;
    const BpbBlkBegins bb = bpb_blk_begins(h_probs, nprobs);
    if (ck == 8) hipLaunchKernelGGL((bpb_conv_s1w_kernel<1>), dim3(nblk), dim3(256), lds, stream, d_probs, bb);
    else hipLaunchKernelGGL((bpb_conv_s1w_kernel<2>), dim3(nblk), dim3(256), lds, stream, d_probs, bb);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
