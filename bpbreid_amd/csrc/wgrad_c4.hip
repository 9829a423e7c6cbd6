// Weight gradient of the stem convolutions: 3 input channels (4 in the NHWC4 image), 3x3 or 7x7 filter, stride 2.
//   torchreid/models/hrnet.py:319-320 (conv1 3 -> 64, 3x3 / 2), torchreid/models/resnet.py:211-213 (conv1 3 -> 64, 7x7 / 2, pad 3),
//   backward: dW[co][ci][r][s] = sum_{n,a,b} x[n][a * sa + r + ih0][b * sa + s + iw0][ci] * dy[n][a][b][co].
//
// GEMM view: M = (tap, ci), N = co, K = all output pixels (524 288 at 64 x 256 x 128).  The first-generation kernel
// (bpb_conv_wgrad_kernel<9,1>) puts ci on the 32 MFMA rows -- 4 of 32 used -- and gives every group of nine taps a workgroup of its
// own that re-stages x and dy: 11 TFLOP/s, 0.21 ms for the 3x3 stem and 1.8 ms (8 % of the ResNet-50 step) for the 7x7 one.
// Here an MFMA row is a (tap, ci) pair: an M tile = 8 taps x 4 channels, wave w owns M tiles {w, w + 4, ...} for ALL pixels of the
// workgroup's tile range and both 32-channel halves of co -- no cross-wave reduction, x and dy staged once per pixel tile
// (buffer_load ... lds double buffer) and shared by the four waves.  The A operand is a gather: lane (tap, ci) reads
// x_halo[pixel + tap offset][ci] with a per-lane constant offset.  Results go to the split-K slab layout [split][tap][4][Cout] that
// bpb_wgrad_reduce_multi sums into OIHW.
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define M24(a, b) __umul24((unsigned)(a), (unsigned)(b))

__device__ __forceinline__ unsigned c4_fdiv(unsigned x, unsigned d, unsigned magic)
{
    return d == 1 ? x : __umulhi(x, magic);
}

template <int MTW>      // M tiles (8 taps x 4 channels) per wave: 1 (T <= 32 taps), 2 (T <= 64)
__global__ __launch_bounds__(256, 2) void bpb_wgrad_c4_kernel(const BpbWgradProb* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbWgradProb P = probs[pi];
    bid -= P.blk_begin;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int cot = bid % P.n_cotiles, split = bid / P.n_cotiles;
    const int co0 = cot * 64, Cout = P.Cout;
    const int HWd = P.HW, HH = P.HH, sa = P.sa;
    const int lTW = P.lTW, lTH = P.lTH, lTI = P.lTI;
    const int TWm = (1 << lTW) - 1, THm = (1 << lTH) - 1;
    const int halo_slots = (1 << lTI) * HH * HWd;         // one 16-byte slot per staged pixel (4 channels)
    const int halo_reg = (halo_slots + 3) & ~3;
    constexpr int dy_slots = 64 * 16;                     // 64 pixels x 64 output channels
    const int bufbytes = (halo_reg + dy_slots) * 16;

    f32x16 acc[MTW][2];
#pragma unroll
    for (int j = 0; j < MTW; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;

    // this lane's (tap, ci) rows: byte offset of the tap inside the staged image + the channel
    int tapoff[MTW];
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
        const int tap = min((wave + 4 * j) * 8 + (l31 >> 2), P.T - 1);      // rows of taps beyond the filter are never stored
        const int r = tap / P.S, s = tap - r * P.S;
        tapoff[j] = ((r * HWd + s) * 4 + (l31 & 3)) * 4;
    }
    const bool live0 = wave * 8 < P.T;                                    // (wave-uniform) does this wave own any tap at all?
    const bool live1 = MTW > 1 && (wave + 4) * 8 < P.T;

    const int per = (P.n_mtiles + P.nsplit - 1) / P.nsplit;
    const int mt_begin = split * per, mt_end = min(P.n_mtiles, mt_begin + per);
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, (int)P.dy_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr unsigned OOB = 0x80000000u;
    auto dma_issue = [&](int mtile, int buf) {
        const int tb = mtile % P.tiles_b, t2 = mtile / P.tiles_b;
        const int ta = t2 % P.tiles_a, tn = t2 / P.tiles_a;
        const int n0 = tn << lTI, a0 = ta << lTH, b0 = tb << lTW;
        char* base = (char*)smem + buf * bufbytes + wave * 1024;
        for (int s0 = 0; s0 < halo_slots; s0 += 256) {                     // out-of-image pixels: an out-of-range offset (zero fill)
            const int idx = s0 + (int)threadIdx.x;
            const unsigned t = c4_fdiv((unsigned)idx, HWd, P.magic_hw);
            const int hc = idx - (int)M24(t, HWd);
            const unsigned ti = c4_fdiv(t, HH, P.magic_hh);
            const int hr = t - M24(ti, HH);
            const int n = n0 + (int)ti, ih = a0 * sa + hr + P.ih0, iw = b0 * sa + hc + P.iw0;
            unsigned vo = OOB;
            if (n < P.N && (unsigned)ih < (unsigned)P.Hi && (unsigned)iw < (unsigned)P.Wi) vo = (M24(M24(n, P.Hi) + ih, P.Wi) + iw) * 16u;
            if (idx < halo_slots) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + s0 * 16), 16, (int)vo, 0, 0, 0);
        }
#pragma unroll
        for (int s0 = 0; s0 < dy_slots; s0 += 256) {
            const int idx = s0 + (int)threadIdx.x;
            const int v = idx & 15, m = idx >> 4;
            const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
            const int n = n0 + ti, a = a0 + th, b = b0 + tw;
            const int co = co0 + v * 4;
            unsigned vo = OOB;
            if (n < P.N && a < P.A && b < P.B && co < Cout) vo = ((M24(M24(n, P.A) + a, P.B) + b) * (unsigned)Cout + co) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr_t)(base + (halo_reg + s0) * 16), 16, (int)vo, 0, 0, 0);
        }
    };

    if (mt_begin < mt_end) dma_issue(mt_begin, 0);
    for (int mtile = mt_begin; mtile < mt_end; ++mtile) {
        __syncthreads();   // this tile has landed (the barrier drains vmcnt) and the other buffer is free
        const int bufoff = ((mtile - mt_begin) & 1) * bufbytes;
        if (mtile + 1 < mt_end) dma_issue(mtile + 1, (mtile + 1 - mt_begin) & 1);
        if (!live0) continue;                                              // (3x3: waves 2, 3 only help with the staging)
        const char* sx = (const char*)smem + bufoff;
        const char* sdy = sx + halo_reg * 16;
        // 32 k-steps of 2 pixels: lane half h holds pixel 2 * ks + h
        auto xoff = [&](int ks) {
            const int m = 2 * ks + half;
            const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
            return (int)M24(M24(M24(ti, HH) + M24(th, sa), HWd) + M24(tw, sa), 16);
        };
        float a[2][MTW], b[2][2];
        auto fetch = [&](int ks, float (&af)[MTW], float (&bf)[2]) {
            const int xo = xoff(ks);
#pragma unroll
            for (int j = 0; j < MTW; ++j) af[j] = *(const float*)(sx + xo + tapoff[j]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bf[nt] = *(const float*)(sdy + (2 * ks + half) * 256 + (nt * 32 + l31) * 4);
        };
        fetch(0, a[0], b[0]);
#pragma unroll 4
        for (int ks = 0; ks < 32; ++ks) {
            if (ks + 1 < 32) fetch(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[0][nt] = MFMA32(a[ks & 1][0], b[ks & 1][nt], acc[0][nt]);
                if (MTW > 1 && live1) acc[MTW - 1][nt] = MFMA32(a[ks & 1][MTW - 1], b[ks & 1][nt], acc[MTW - 1][nt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- slab rows of this wave: C/D layout of the 32x32 MFMA: column = lane & 31 (co), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    // = (tap within the M tile) * 4 + ci
    bpb_gf ws = (bpb_gf)P.ws;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = co0 + nt * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int tap = (wave + 4 * j) * 8 + (row >> 2), ci = row & 3;
                const float v = acc[j][nt][r];
                if (tap < P.T && co < Cout) ws[(((size_t)split * P.T + tap) * 4 + ci) * Cout + co] = v;
            }
        }
    }
}

extern "C" {

int bpb_wgrad_c4_init(void)
{
#define BPB_ATTR(K)                                                                                                  \
    {                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        if (e != hipSuccess) return bpb_set_error((int)e, "bpb_wgrad_c4_init: %s", hipGetErrorString(e));             \
    }
    BPB_ATTR((bpb_wgrad_c4_kernel<1>)) BPB_ATTR((bpb_wgrad_c4_kernel<2>))
#undef BPB_ATTR
    return 0;
}

// Grouped launch of stem weight-gradient problems (Cin = 4: the NHWC4 image; 64-pixel tiles; T <= 64 taps).  Same descriptor and
// slab layout as bpb_conv_wgrad ([nsplit][T][4][Cout]); replaces conv backward-weight of hrnet.py:319-320 / resnet.py:211-213.
int bpb_conv_wgrad_c4(const BpbWgradProb* d_probs, const BpbWgradProb* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_wgrad_c4: nprobs=%d out of range", nprobs);
    int nblk = 0, lds = 0, mtw = 0;
    for (int i = 0; i < nprobs; ++i) {
        const BpbWgradProb& p = h_probs[i];
        BPB_REQUIRE(p.Cin == 4 && p.Cout % 4 == 0, "bpb_conv_wgrad_c4: Cin=%d must be 4 (the NHWC4 image), Cout=%d a multiple of 4", p.Cin, p.Cout);
        BPB_REQUIRE(p.lTI + p.lTH + p.lTW == 6, "bpb_conv_wgrad_c4: M tile must be 64 pixels");
        BPB_REQUIRE(p.T >= 1 && p.T <= 64 && p.S >= 1 && p.T % p.S == 0, "bpb_conv_wgrad_c4: T=%d S=%d", p.T, p.S);
        BPB_REQUIRE(p.HH == ((1 << p.lTH) - 1) * p.sa + p.T / p.S && p.HW == ((1 << p.lTW) - 1) * p.sa + p.S, "bpb_conv_wgrad_c4: staged extent mismatch");
        BPB_REQUIRE(p.n_cotiles == bpb_cdiv(p.Cout, 64) && p.nsplit >= 1 && p.n_mtiles >= 1, "bpb_conv_wgrad_c4: tile counts mismatch");
        BPB_REQUIRE(p.tiles_a == bpb_cdiv(p.A, 1 << p.lTH) && p.tiles_b == bpb_cdiv(p.B, 1 << p.lTW) &&
                        p.n_mtiles == bpb_cdiv(p.N, 1 << p.lTI) * p.tiles_a * p.tiles_b, "bpb_conv_wgrad_c4: M tile count mismatch");
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_wgrad_c4: blk_begin mismatch");
        BPB_REQUIRE(p.x_bytes > 0 && p.dy_bytes > 0 && p.x_bytes < 0x80000000u && p.dy_bytes < 0x80000000u,
                    "bpb_conv_wgrad_c4: tensors addressed through a buffer descriptor must be < 2 GiB");
        BPB_REQUIRE((double)p.N * p.Hi * p.Wi < 16777216.0 && (double)p.N * p.A * p.B < 16777216.0, "bpb_conv_wgrad_c4: 24-bit index arithmetic overflow");
        const int this_mtw = p.T <= 32 ? 1 : 2;
        BPB_REQUIRE(mtw == 0 || mtw == this_mtw, "bpb_conv_wgrad_c4: filters of <= 32 and > 32 taps cannot share a group");
        mtw = this_mtw;
        nblk += p.nsplit * p.n_cotiles;
        const int halo_reg = ((1 << p.lTI) * p.HH * p.HW + 3) & ~3;
        const int l = 2 * (halo_reg + 64 * 16) * 16;
        lds = l > lds ? l : lds;
    }
    BPB_REQUIRE(lds <= 160 * 1024, "bpb_conv_wgrad_c4: needs %d B of LDS", lds);
    if (nblk == 0) return 0;
    if (mtw == 1) hipLaunchKernelGGL((bpb_wgrad_c4_kernel<1>), dim3(nblk), dim3(256), lds, stream, d_probs, bpb_blk_begins(h_probs, nprobs));
    else hipLaunchKernelGGL((bpb_wgrad_c4_kernel<2>), dim3(nblk), dim3(256), lds, stream, d_probs, bpb_blk_begins(h_probs, nprobs));
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
