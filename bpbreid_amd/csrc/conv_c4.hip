// Forward of the stem convolutions: 3 input channels (4 in the NHWC4 image), 3x3 or 7x7 filter, stride 2, padding R / 2, 64 output
// channels -- torchreid/models/hrnet.py:319-320 (conv1 3 -> 64, 3x3 / 2), torchreid/models/resnet.py:211-213 (conv1 3 -> 64, 7x7 / 2).
//
// GEMM view: M = output pixels, N = 64 channels, K = (tap, ci) with the three REAL channels only (the general kernel ran these with
// K = 4 per tap behind a run-time tap iterator: 25-37 TFLOP/s, 0.35 ms for the 7x7 stem).  fp32 MFMA 32x32x2: one step multiplies two
// k values, lane half 0 its first, lane half 1 its second.  Step s = (tap pair j, channel i): half 0 takes (tap 2j, ci i), half 1
// (tap 2j + 1, ci i) -- the summation order of the general kernel's 3-channel path (whose fourth step per pair added the zero channel),
// so the results are bit-identical to it; 15 / 75 steps for the 3x3 / 7x7 filter (an odd tap count ends on a zero row of the weight
// matrix).  The A operand of (tap, ci) for output pixel m is x_halo[pixel(m) + tap offset][ci], a 4-byte LDS read at
// base[class(s)] + constant(s): the base registers hold the lane's pixel offset plus (lane half) * the distance from tap 2j to tap
// 2j + 1 (next pixel: 16 bytes; first tap of the next filter row: one staged row minus R - 1 pixels; phantom tap: 0) -- the k-loop
// is fully unrolled, every address is a register plus an immediate.  The B operand is the weight matrix [2 * steps][64] kept in LDS
// for the workgroup's whole life.
//
// A workgroup walks a contiguous range of 8 x 16-pixel output tiles (the staged (14 + R) x (30 + R) input pixels arrive by
// buffer_load ... lds, double-buffered under the MFMA loop of the previous tile), wave w owns 32 pixels x 64 channels.  BatchNorm
// statistics (training) are summed in fp64 over ALL tiles of the workgroup and written once: `nblk` partial rows instead of one per
// tile.  Eval plan: bias (the folded BatchNorm shift) and ReLU in the epilogue.
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define M24(a, b) __umul24((unsigned)(a), (unsigned)(b))

struct BpbConvC4Args {
    const float* x;
    const float* w;      // forward packing [T][1][64][4]
    float* y;
    const float* bias;
    double* stats;       // [nblk][2][64] or null
    int N, Hi, Wi, H, W, relu;
    int tiles_a, tiles_b, n_mtiles, per;     // per = tiles per workgroup
    unsigned x_bytes, y_bytes;
};

template <int R>
__global__ __launch_bounds__(256, 2) void bpb_conv_c4_kernel(BpbConvC4Args P)
{
    constexpr int T = R * R, PAD = R / 2, TH = 8, TW = 16, HH = 2 * TH + R - 2, HWd = 2 * TW + R - 2, NPIX = HH * HWd;
    constexpr int KS = ((T + 1) / 2) * 3;                 // 15 / 75 MFMA steps: (tap pair, channel); the phantom tap of an odd T is a zero row
    constexpr int HALO_REG = (NPIX + 3) & ~3;             // 16-byte slots per staged image
    constexpr int D_ROW = (HWd - (R - 1)) * 16;           // tap 2j -> tap 2j + 1 when tap 2j ends a filter row
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem + 2 * HALO_REG * 4;                  // [2 * KS][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mt_begin = blockIdx.x * P.per, mt_end = min(P.n_mtiles, mt_begin + P.per);

    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    constexpr unsigned OOB = 0x80000000u;
    auto tile_origin = [&](int mtile, int& n, int& a0, int& b0) {
        const int tb = mtile % P.tiles_b, t2 = mtile / P.tiles_b;
        n = t2 / P.tiles_a;
        a0 = (t2 - n * P.tiles_a) * TH;
        b0 = tb * TW;
    };
    auto dma_issue = [&](int mtile, int buf) {
        int n, a0, b0;
        tile_origin(mtile, n, a0, b0);
        char* base = (char*)smem + buf * (HALO_REG * 16) + wave * 1024;
#pragma unroll
        for (int s0 = 0; s0 < NPIX; s0 += 256) {           // out-of-image pixels: an out-of-range offset (zero fill)
            const int idx = s0 + (int)threadIdx.x;
            const int hr = idx / HWd, hc = idx - hr * HWd;
            const int ih = a0 * 2 + hr - PAD, iw = b0 * 2 + hc - PAD;
            unsigned vo = OOB;
            if ((unsigned)ih < (unsigned)P.Hi && (unsigned)iw < (unsigned)P.Wi) vo = (M24(M24(n, P.Hi) + ih, P.Wi) + iw) * 16u;
            if (idx < NPIX) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + s0 * 16), 16, (int)vo, 0, 0, 0);
        }
    };
    if (mt_begin < mt_end) dma_issue(mt_begin, 0);

    // ---- the weight matrix [2 * step + lane half][64] = W[tap 2j + half][ci i] from the forward packing wf[tap][co][4]; phantom tap: zero
    {
        bpb_gcf gw = (bpb_gcf)P.w;
#pragma unroll 8
        for (int idx = threadIdx.x; idx < 2 * KS * 64; idx += 256) {
            const int row = idx >> 6, co = idx & 63;
            const int st = row >> 1, tap = 2 * (st / 3) + (row & 1), ci = st % 3;
            wl[idx] = tap < T ? gw[(tap * 64 + co) * 4 + ci] : 0.f;
        }
    }

    // this lane's pixel inside the staged image, and the A bases of the three tap 2j -> tap 2j + 1 distances
    const int m = wave * 32 + l31;
    const int pixbase = (((m >> 4) * 2) * HWd + (m & 15) * 2) * 16;
    int abase[3];
    abase[0] = pixbase + half * 16;         // tap 2j + 1 = the next pixel of the filter row
    abase[1] = pixbase + half * D_ROW;      // ... = the first tap of the next filter row
    abase[2] = pixbase;                     // ... = the phantom tap (zero weights): any finite value
    const int bbase = 2 * HALO_REG * 16 + (half * 64 + l31) * 4;
    auto a_cls = [](int s) {
        const int t0 = 2 * (s / 3);
        return t0 + 1 >= T ? 2 : (t0 + 1) % R != 0 ? 0 : 1;
    };
    auto a_imm = [](int s) {
        const int t0 = 2 * (s / 3), ci = s % 3;
        return ((t0 / R) * HWd + t0 % R) * 16 + ci * 4;
    };

    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, (int)P.y_bytes, 0x00020000);
    const bpb_gcf gbias = (bpb_gcf)P.bias;
    float bias_v[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bias_v[nt] = gbias ? gbias[nt * 32 + l31] : 0.f;
    const bool relu = P.relu != 0, do_stats = P.stats != nullptr;
    double ssum[2] = {0.0, 0.0}, ssq[2] = {0.0, 0.0};      // fp64 from the first element on, as in conv_s1

    for (int mtile = mt_begin; mtile < mt_end; ++mtile) {
        __syncthreads();   // this tile has landed (the barrier drains vmcnt; the first one also covers the weight matrix), the other buffer is free
        const int bufoff = ((mtile - mt_begin) & 1) * (HALO_REG * 16);
        if (mtile + 1 < mt_end) dma_issue(mtile + 1, (mtile + 1 - mt_begin) & 1);
        const char* lds = (const char*)smem;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        float fa[2], fb[2][2];
        fa[0] = *(const float*)(lds + bufoff + abase[a_cls(0)] + a_imm(0));
        fb[0][0] = *(const float*)(lds + bbase);
        fb[0][1] = *(const float*)(lds + bbase + 128);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
                fa[(s + 1) & 1] = *(const float*)(lds + bufoff + abase[a_cls(s + 1)] + a_imm(s + 1));
                fb[(s + 1) & 1][0] = *(const float*)(lds + bbase + (s + 1) * 512);
                fb[(s + 1) & 1][1] = *(const float*)(lds + bbase + (s + 1) * 512 + 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA32(fa[s & 1], fb[s & 1][0], acc[0]);
            acc[1] = MFMA32(fa[s & 1], fb[s & 1][1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        // the four rows of a register quad are four consecutive pixels of one image row (tile width 16)
        int n, a0, b0;
        tile_origin(mtile, n, a0, b0);
        unsigned offs[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int mm = wave * 32 + 8 * rq + 4 * half;
            const int a = a0 + (mm >> 4), b = b0 + (mm & 15);
            const unsigned qoff = (M24(M24(n, P.H) + a, P.W) + b) * 256u + (unsigned)(l31 * 4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) offs[rq * 4 + jj] = (a < P.H && b + jj < P.W) ? qoff + (unsigned)(jj * 256) : OOB;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[nt][r] + bias_v[nt];
                if (relu) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)(offs[r] + (unsigned)(nt * 128)), 0, 0);
                if (do_stats) {
                    const double dv = offs[r] != OOB ? (double)v : 0.0;
                    ssum[nt] += dv;
                    ssq[nt] += dv * dv;
                }
            }
    }
    if (do_stats) {   // one partial row per workgroup, combined in a fixed order (deterministic: no atomics)
        __syncthreads();                                   // every wave is done with the staging buffers
        double* red = (double*)smem;                       // [wave][64][2]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const double s = ssum[nt] + __shfl_xor(ssum[nt], 32);
            const double q = ssq[nt] + __shfl_xor(ssq[nt], 32);
            if (half == 0) {
                red[((wave * 2 + nt) * 32 + l31) * 2 + 0] = s;
                red[((wave * 2 + nt) * 32 + l31) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            double s = 0.0, q = 0.0;
            for (int w = 0; w < 4; ++w) {
                s += red[(w * 64 + (int)threadIdx.x) * 2 + 0];
                q += red[(w * 64 + (int)threadIdx.x) * 2 + 1];
            }
            P.stats[((size_t)blockIdx.x * 2 + 0) * 64 + threadIdx.x] = s;
            P.stats[((size_t)blockIdx.x * 2 + 1) * 64 + threadIdx.x] = q;
        }
    }
}

template <int R>
static constexpr int conv_c4_lds()
{
    constexpr int NPIX = (16 + R - 2) * (32 + R - 2), HALO_REG = (NPIX + 3) & ~3, KS = ((R * R + 1) / 2) * 3;
    return 2 * HALO_REG * 16 + 2 * KS * 64 * 4;
}

extern "C" {

int bpb_conv_c4_init(void)
{
#define BPB_ATTR(K)                                                                                                  \
    {                                                                                                                \
        hipError_t e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
        if (e != hipSuccess) return bpb_set_error((int)e, "bpb_conv_c4_init: %s", hipGetErrorString(e));              \
    }
    BPB_ATTR((bpb_conv_c4_kernel<3>)) BPB_ATTR((bpb_conv_c4_kernel<7>))
#undef BPB_ATTR
    return 0;
}

// Stem convolution forward: y[N, H, W, 64] = conv_RxR(x[N, Hi, Wi, 4 (3 real channels)]), stride 2, padding R / 2, H = (Hi - 1) / 2 + 1.
// `w` is the forward packing of bpb_pack_weights; `stats` (training) receives nblk rows of (sum, sum of squares) per channel;
// `bias` / `relu`: the eval plan's folded BatchNorm.  Replaces aten::conv2d of hrnet.py:319-320 / resnet.py:211-213.
int bpb_conv_c4(const float* x, const float* w, float* y, const float* bias, double* stats, int N, int Hi, int Wi, int R, int Cout,
                int relu, int nblk, hipStream_t stream)
{
    BPB_REQUIRE(R == 3 || R == 7, "bpb_conv_c4: filter size %d (3 or 7)", R);
    BPB_REQUIRE(Cout == 64, "bpb_conv_c4: Cout=%d must be 64", Cout);
    BPB_REQUIRE(N >= 1 && Hi >= 1 && Wi >= 1 && nblk >= 1, "bpb_conv_c4: empty problem");
    BpbConvC4Args a;
    a.x = x, a.w = w, a.y = y, a.bias = bias, a.stats = stats;
    a.N = N, a.Hi = Hi, a.Wi = Wi, a.relu = relu;
    a.H = (Hi - 1) / 2 + 1, a.W = (Wi - 1) / 2 + 1;
    a.tiles_a = bpb_cdiv(a.H, 8), a.tiles_b = bpb_cdiv(a.W, 16);
    const double nm = (double)N * a.tiles_a * a.tiles_b;
    BPB_REQUIRE((double)N * Hi * Wi < 16777216.0 && (double)N * a.H * a.W < 16777216.0, "bpb_conv_c4: 24-bit index arithmetic overflow");
    a.n_mtiles = (int)nm;
    BPB_REQUIRE(nblk <= a.n_mtiles, "bpb_conv_c4: %d workgroups for %d tiles", nblk, a.n_mtiles);
    a.per = bpb_cdiv(a.n_mtiles, nblk);
    BPB_REQUIRE(bpb_cdiv(a.n_mtiles, a.per) == nblk, "bpb_conv_c4: nblk=%d leaves workgroups without a tile (%d tiles, %d per workgroup): the statistics rows would be undefined",
                nblk, a.n_mtiles, a.per);
    const double xb = (double)N * Hi * Wi * 16.0, yb = (double)N * a.H * a.W * 256.0;
    BPB_REQUIRE(xb < 2147483648.0 && yb < 2147483648.0, "bpb_conv_c4: tensors addressed through a buffer descriptor must be < 2 GiB");
    BPB_REQUIRE(((uintptr_t)x & 15) == 0, "bpb_conv_c4: x must be 16-byte aligned");
    a.x_bytes = (unsigned)xb, a.y_bytes = (unsigned)yb;
    if (R == 3) hipLaunchKernelGGL((bpb_conv_c4_kernel<3>), dim3(nblk), dim3(256), conv_c4_lds<3>(), stream, a);
    else hipLaunchKernelGGL((bpb_conv_c4_kernel<7>), dim3(nblk), dim3(256), conv_c4_lds<7>(), stream, a);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
