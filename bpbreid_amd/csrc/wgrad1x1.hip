// Weight gradient of 1x1 convolutions:  dW[ci][co] = sum_p x[p][ci] * dy[p][co]   (p over all N*A*B output pixels; a strided
// convolution reads input pixel (n, a*sa, b*sa))
// Replaces conv backward-weight of torchreid/models/resnet.py:119-127 (Bottleneck 1x1 convs: 53 of ResNet-50's convolutions),
// hrnet.py:104-110 (layer1 Bottlenecks) and the 1x1 convs of the HRNet head (hrnet.py:319-350) on the path.
//
// GEMM view: M = ci, N = co, K = pixels -- a short, wide product with an enormous K (131 072 pixels at 64 x 32 x batch 64), so
// the kernel is HBM-bound by construction: x and dy should each be read ONCE.  The first-generation kernel
// (bpb_conv_wgrad_kernel<1,2>) tiles 32 ci x 64 co per workgroup and re-reads x Cout/64 times and dy Cin/32 times (40-48 TFLOP/s;
// 26 % of ResNet-50's step).  Here a workgroup owns a (64 << lwm) x (256 >> lwm) channel tile -- 64 x 256, 128 x 128 or
// 256 x 64, chosen per problem to minimise the re-reads -- and streams 32-pixel tiles of both operands through LDS by
// buffer_load ... lds DMA (double-buffered, planar [16-channel plane][pixel][16]: conflict-free ds_read_b32 without padding, as
// in wgrad16.hip).  Each of the 4 waves keeps a 64 x 64 accumulator (4 x 4 v_mfma_f32_16x16x4_f32 tiles, 64 VGPRs): 8 LDS reads
// per 16 MFMAs.  Split-K over pixel ranges writes slabs [split][Cin][Cout] that bpb_wgrad_reduce_multi sums in a fixed order.
#include "bpb_common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256, 1) void bpb_wgrad1x1_kernel(const BpbWgrad1x1Prob* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MPIX = 32, NKS = MPIX / 4, PLANE = MPIX * 64;      // bytes of one 16-channel plane of a pixel tile
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbWgrad1x1Prob P = probs[pi];
    bid -= P.blk_begin;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int lwm = P.lwm;                                   // 2^lwm waves along ci, 4 >> lwm along co
    const int wmi = wave & ((1 << lwm) - 1), wni = wave >> lwm;
    const int CIP = 4 << lwm, COP = 16 >> lwm;              // 16-channel planes of x / dy per workgroup
    // block -> (split, ci tile, co tile), co fastest
    const int cot = bid % P.n_cotiles;
    const int r1 = bid / P.n_cotiles;
    const int cit = r1 % P.n_citiles;
    const int split = r1 / P.n_citiles;
    const int Cin = P.Cin, Cout = P.Cout;
    const int ci0 = cit * (64 << lwm), co0 = cot * (256 >> lwm);
    const int bufbytes = (CIP + COP) * PLANE;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA pieces (256 x 16 B): slot idx = plane * 128 + pixel * 4 + quarter; a piece (2 planes) is all x or all dy
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr int DMA_P = 10;
    const int npieces = (CIP + COP) >> 1;
    unsigned rel[DMA_P], pk[DMA_P];           // pk = pixel in tile | never-valid << 31
#pragma unroll
    for (int k = 0; k < DMA_P; ++k) {
        const int idx = k * 256 + (int)threadIdx.x;
        const int plane = idx >> 7, pix = (idx & 127) >> 2, quarter = idx & 3;
        const bool isx = plane < CIP;
        const int c = isx ? ci0 + plane * 16 + quarter * 4 : co0 + (plane - CIP) * 16 + quarter * 4;
        const int C = isx ? Cin : Cout;
        rel[k] = (unsigned)(pix * C + c) * 4u;
        pk[k] = (unsigned)pix | ((unsigned)c & 0x7fffu) << 8 | ((k < npieces && c < C) ? 0u : 0x80000000u);
    }
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, (int)P.dy_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int xpieces = CIP >> 1;
    auto dma_issue = [&](int ptile, int buf) {
        const int p0 = ptile * MPIX;
        const int remaining = P.npix - p0;
        const unsigned xoff = (unsigned)p0 * (unsigned)Cin * 4u, yoff = (unsigned)p0 * (unsigned)Cout * 4u;
        char* base = (char*)smem + buf * bufbytes + wave * 1024;
#pragma unroll
        for (int k = 0; k < DMA_P; ++k) {
            if (k < npieces) {
                const bool ok = (int)pk[k] >= 0 && (int)(pk[k] & 255u) < remaining;
                if (k < xpieces) {
                    unsigned off = xoff + rel[k];
                    if (P.sa != 1) {            // strided 1x1 (ResNet downsample): output pixel (n, a, b) reads input pixel (n, a*sa, b*sa)
                        const unsigned q = (unsigned)p0 + (pk[k] & 255u);
                        const unsigned n = __umulhi(q, P.magic_ab), r = q - n * (unsigned)(P.A * P.B);
                        const unsigned a = __umulhi(r, P.magic_b), b = r - a * (unsigned)P.B;
                        off = (((n * (unsigned)P.Hi + a * (unsigned)P.sa) * (unsigned)P.Wi + b * (unsigned)P.sa) * (unsigned)Cin +
                               ((pk[k] >> 8) & 0x7fffu)) * 4u;
                    }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + k * 4096), 16, (int)(ok ? off : OOB), 0, 0, 0);
                } else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr_t)(base + k * 4096), 16, (int)(ok ? yoff + rel[k] : OOB), 0, 0, 0);
            }
        }
    };

    const int per = (P.n_ptiles + P.nsplit - 1) / P.nsplit;
    const int t_begin = split * per, t_end = min(P.n_ptiles, t_begin + per);
    // this lane's operand addresses inside buffer 0: k-step ks and 16-channel block i / j are immediates
    int xa = (wmi * 4) * PLANE + kq * 64 + l15 * 4;
    int ya = (CIP + wni * 4) * PLANE + kq * 64 + l15 * 4;
    if (t_begin < t_end) dma_issue(t_begin, 0);
    for (int pt = t_begin; pt < t_end; ++pt) {
        __syncthreads();   // this tile has landed (the barrier drains vmcnt) and the other buffer is free again
        const int cur = (pt - t_begin) & 1;
        if (pt + 1 < t_end) dma_issue(pt + 1, cur ^ 1);
        const char* lds = (const char*)smem;
        auto fetch = [&](int ks, float (&a)[4], float (&b)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const float*)(lds + xa + i * PLANE + ks * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const float*)(lds + ya + j * PLANE + ks * 256);
        };
        auto mma = [&](const float (&a)[4], const float (&b)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA16(a[i], b[j], acc[i][j]);
        };
        float a0[4], b0[4], a1[4], b1[4];
        fetch(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < NKS; ks += 2) {
            fetch(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < NKS) fetch(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int delta = cur ? -bufbytes : bufbytes;   // the next tile lives in the other buffer
        xa += delta;
        ya += delta;
    }

    // ---- straight to the slab.  C/D layout of the 16x16 MFMA: col = lane & 15 (co), row = 4 * (lane >> 4) + reg (ci)
    bpb_gf ws = (bpb_gf)P.ws;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + (wni * 4 + j) * 16 + l15;
        if (co < Cout) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + (wmi * 4 + i) * 16 + kq * 4 + r;
                    if (ci < Cin) ws[((size_t)split * Cin + ci) * Cout + co] = acc[i][j][r];
                }
        }
    }
}

extern "C" {

int bpb_wgrad1x1_init(void)
{
    hipError_t e = hipFuncSetAttribute((const void*)bpb_wgrad1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return bpb_set_error((int)e, "bpb_wgrad1x1_init: %s", hipGetErrorString(e));
    return 0;
}

// Grouped launch of 1x1 stride-1 weight-gradient problems (slab layout of bpb_conv_wgrad with T = 1).
int bpb_conv_wgrad1x1(const BpbWgrad1x1Prob* d_probs, const BpbWgrad1x1Prob* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_wgrad1x1: nprobs=%d out of range", nprobs);
    int nblk = 0, lds = 0;
    for (int i = 0; i < nprobs; ++i) {
        const BpbWgrad1x1Prob& p = h_probs[i];
        BPB_REQUIRE(p.lwm >= 0 && p.lwm <= 2, "bpb_conv_wgrad1x1: lwm=%d", p.lwm);
        BPB_REQUIRE(p.Cin % 4 == 0 && p.Cout % 4 == 0 && p.npix >= 1 && p.Cin < 32768, "bpb_conv_wgrad1x1: Cin/Cout must be multiples of 4");
        BPB_REQUIRE(p.sa == 1 || (p.sa >= 2 && p.A >= 2 && p.B >= 2 && (long)p.A * p.B >= 2 && p.npix % (p.A * p.B) == 0 &&
                                  (p.A - 1) * p.sa < p.Hi && (p.B - 1) * p.sa < p.Wi),
                    "bpb_conv_wgrad1x1: strided problem needs the output extent A x B (>= 2 each) and the input extent");
        BPB_REQUIRE(p.n_citiles == bpb_cdiv(p.Cin, 64 << p.lwm) && p.n_cotiles == bpb_cdiv(p.Cout, 256 >> p.lwm) &&
                        p.n_ptiles == bpb_cdiv(p.npix, 32) && p.nsplit >= 1 && p.nsplit <= p.n_ptiles,
                    "bpb_conv_wgrad1x1: tile counts mismatch");
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_wgrad1x1: blk_begin mismatch");
        BPB_REQUIRE(p.x_bytes > 0 && p.dy_bytes > 0 && p.x_bytes < 0xFFFFFFF0u && p.dy_bytes < 0xFFFFFFF0u,
                    "bpb_conv_wgrad1x1: tensors addressed through a buffer descriptor must be < 4 GiB");
        BPB_REQUIRE(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.dy & 15) == 0, "bpb_conv_wgrad1x1: x/dy must be 16-byte aligned");
        nblk += p.nsplit * p.n_citiles * p.n_cotiles;
        const int l = 2 * ((4 << p.lwm) + (16 >> p.lwm)) * 2048;
        lds = l > lds ? l : lds;
    }
    if (nblk == 0) return 0;
    hipLaunchKernelGGL(bpb_wgrad1x1_kernel, dim3(nblk), dim3(256), lds, stream, d_probs, bpb_blk_begins(h_probs, nprobs));
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
