// Implicit-GEMM convolution on the fp32 MFMA pipe of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, on the BPBReID hot path, every nn.Conv2d the reference dispatches to cuDNN/MIOpen
// (torchreid/models/hrnet.py:61-64,104-110,184,223,240-250,319-323,459-481 and
//  torchreid/models/resnet.py:31-49,211-216): forward, data-gradient and weight-gradient.
//
// Why fp32 MFMA: the contract is 1e-4 parity with an fp32 CPU oracle through ~320 conv+BN
// layers.  v_mfma_f32_32x32x2_f32 is bit-exactly an fp32 fma chain (157 TF peak); bf16 inputs
// cannot hold the tolerance.  The roofline for these kernels is therefore
// the fp32-matrix peak, 157.3 TFLOP/s.
//
// Data layout: activations NHWC fp32.  One workgroup = 4 waves = a (4 * MT * 32 >> lwn)-pixel x ((32 * NT) << lwn)-channel
// output tile (MT, NT template parameters, lwn per problem).  Per chunk of CK input channels the (TI x TH x TW) pixel
// tile's input halo ([halo pixel][CK + 4 pad] -> conflict-free 16-byte fragment reads) and the weight tile
// ([tap][CK/4][N tile][4], from the pre-packed [tap][Cin/4][Cout][4] array) are brought into LDS by buffer_load ... lds
// DMA, double-buffered against the MFMA loop of the previous chunk; the halo is reused by every filter tap (9x for 3x3).
// MFMA operand packing: for one 16-byte LDS read lanes 0-31 hold channels c..c+3 of pixel (lane&31) and lanes 32-63
// channels c+4..c+7, which is exactly the A fragment of four consecutive 32x32x2 MFMAs (B likewise); the reads of
// k-group j+1 are issued before the MFMAs of k-group j.  DESIGN.md section 4 has the full anatomy and section 5 the
// measurements (what bounds the kernel, what was tried).
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate): exact when both OPERANDS are < 2^24 and the product < 2^32.
// Used for halo-local coordinates and for (image, row, column) -> pixel index; the pixel index itself may exceed 2^24,
// so its product with the channel count stays a 32-bit multiply.
#define M24(a, b) __umul24((unsigned)(a), (unsigned)(b))

__device__ __forceinline__ unsigned bpb_fdiv(unsigned x, unsigned d, unsigned magic)
{
    return d == 1 ? x : __umulhi(x, magic);
}

template <int NT, bool C4, int MTr>
__global__ __launch_bounds__(256, 2) void bpb_conv_igemm_kernel(const BpbConvProb* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbConvProb P = probs[pi];   // by value: the whole descriptor sits in SGPRs, no reloads in the loops
    bid -= P.blk_begin;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int lTW = P.lTW, lTH = P.lTH, lTI = P.lTI;
    const int TWm = (1 << lTW) - 1, THm = (1 << lTH) - 1;
    // One workgroup walks `tpb` consecutive M tiles of one N tile: the halo of tile i+1 streams into LDS while the MFMA loop
    // of tile i runs and while its epilogue drains, so the load / compute / store phases of the workgroups on a CU no longer
    // run in lock-step (with one tile per workgroup and every workgroup launched at once they do).
    const int ntile = bid % P.n_ntiles, mgroup = bid / P.n_ntiles;
    const int mt_first = mgroup * P.tpb, mt_last = min(P.n_mtiles, mt_first + P.tpb);
    const int LD = P.LD, HWd = P.HW, HH = P.HH, sa = P.sa;
    const int Cin = P.Cin, Cout = P.Cout, cin4 = Cin >> 2;
    const int Rt = P.Rt, St = P.St, ntaps = Rt * St;
    // Workgroup tile = (WM * MT * 32 pixels) x (WN * NT * 32 channels) with WM * WN = 4 waves.  MT in {1,2} and WN in {1,2}
    // are per-problem (runtime, wave-uniform): small tiles give the deep, low-resolution HRNet branches enough
    // workgroups to occupy 256 CUs (a 256-channel 8x4 map at batch 64 is only eight 256-pixel tiles).
    const int lwn = P.lwn;                       // MTr (template) = P.mt_r
    const int wm = wave >> lwn, wni = wave & ((1 << lwn) - 1);
    const int NTC = (NT * 32) << lwn;            // output channels per workgroup
    const int lNTC = (NT == 1 ? 5 : 6) + lwn;

    int pixoff[MTr];   // byte offset of this lane's pixel (per 32-pixel sub-tile) inside the halo tile
#pragma unroll
    for (int mt = 0; mt < MTr; ++mt) {
        const int m = (wm * MTr + mt) * 32 + l31;
        const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
        pixoff[mt] = (int)M24(M24(M24(ti, HH) + M24(th, sa), HWd) + M24(tw, sa), LD) * 4 + (C4 ? 0 : half * 16);
    }
    const int cout_l = ntile * NTC + wni * NT * 32 + l31;

    f32x16 acc[MTr][NT];
#pragma unroll
    for (int mt = 0; mt < MTr; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int CK = P.CK;                       // power of two in {4, 8, 16, 32}; divides Cin
    const int lvpp = 31 - __clz(CK) - 2;       // log2(CK / 4)
    const int npix = (1 << lTI) * HH * HWd;
    const int KG = C4 ? 1 : (CK >> 3);         // 8-channel k-groups per tap inside one chunk
    const int nj = C4 ? ((ntaps + 1) >> 1) : ntaps * KG;
    const int nch = Cin / CK;                  // channel chunks per tile
    // LDS map (16-byte "slots"):
    //   halo image(s)  [halo pixel][spp]     spp = LD/4 = CK/4 data slots (+1 pad slot unless Cin == 4); 1 or 2 buffers
    //   weight tile(s) [tap][CK/4][NTC]      wres: one tile per channel chunk, loaded once per workgroup (resident);
    //                                        otherwise one per buffer, re-staged with every halo image
    //   stats scratch  4 KiB                 (never a DMA target)
    // Both MFMA operands come from LDS (in-order ds_read returns -> counted lgkmcnt waits).
    const int qn = CK >> 2;
    const int spp = LD >> 2;
    const int halo_slots = npix * spp;
    const int halo_pad = (halo_slots + 255) & ~255;
    const int nB = (ntaps + (C4 ? 1 : 0)) * qn * NTC;   // (C4: one extra zero tap for the phantom half)
    const int b_pad = (nB + 255) & ~255;
    const bool dma = P.dma != 0, wres = P.wres != 0;
    const int nbuf = dma ? 2 : 1;
    const int wbase = nbuf * halo_pad * 16;                       // byte offset of the weight region
    const int redbase = wbase + (wres ? nch : nbuf) * b_pad * 16; // byte offset of the stats scratch
    const int tapB = qn * NTC * 16;            // bytes per tap in the weight region
    const int boff_lane = (C4 ? 0 : half * NTC * 16) + (wni * NT * 32 + l31) * 16;

    // byte offset of slot `idx` of chunk `cb` inside x / w; DMA_OOB = "out of range" (buffer loads return 0 there).  x and w
    // are < 2 GiB (host check), so DMA_OOB plus any per-chunk increment is still beyond the descriptor's range.
    constexpr unsigned DMA_OOB = 0x80000000u;
    auto halo_voff = [&](int idx, int cb, int n0, int a0, int b0) -> unsigned {
        const unsigned hp = bpb_fdiv((unsigned)idx, spp, P.magic_spp);
        const int v = idx - (int)M24(hp, spp);
        if (idx >= halo_slots || v >= qn) return DMA_OOB;
        const unsigned t = bpb_fdiv(hp, HWd, P.magic_hw);
        const int hc = hp - M24(t, HWd);
        const unsigned ti = bpb_fdiv(t, HH, P.magic_hh);
        const int hr = t - M24(ti, HH);
        const int n = n0 + (int)ti, ih = a0 * sa + hr + P.ih0, iw = b0 * sa + hc + P.iw0;
        if (n < P.N && (unsigned)ih < (unsigned)P.Hi && (unsigned)iw < (unsigned)P.Wi)
            return ((M24(M24(n, P.Hi) + ih, P.Wi) + iw) * (unsigned)Cin + cb + v * 4) * 4u;   // (pixel index may exceed 2^24)
        return DMA_OOB;
    };
    auto b_voff = [&](int bi, int cb) -> unsigned {
        const int n = bi & (NTC - 1);
        const int r = bi >> lNTC;
        const int q = r & (qn - 1);
        const int t = r >> lvpp;
        if (bi >= nB || t >= ntaps) return DMA_OOB;
        const int ti_ = t / St, tj_ = t - ti_ * St;
        const int widx = P.w0 + P.wrs * ti_ + P.wss * tj_;
        const int co = min(ntile * NTC + n, Cout - 1);     // columns >= Cout are never stored
        return (((unsigned)(widx * cin4 + (cb >> 2) + q) * Cout + co) * 4) * 4u;
    };
    // (a) asynchronous path: buffer_load ... lds (global -> LDS DMA, no staging registers), double-buffered so that the
    //     image of work item i+1 (next channel chunk, or first chunk of the next tile) streams in while the MFMA loop of
    //     item i runs.  One barrier per item.
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto tile_origin = [&](int mtile, int& n0, int& a0, int& b0) {
        const int tb = mtile % P.tiles_b, t2 = mtile / P.tiles_b;
        const int ta = t2 % P.tiles_a, tn = t2 / P.tiles_a;
        n0 = tn << lTI; a0 = ta << lTH; b0 = tb << lTW;
    };
    // The offsets of a thread's DMA pieces depend on the chunk only through a constant increment (cb channels of x,
    // cb/4 channel quads of w): they are computed once per tile (~30 VALU instructions each) and kept in registers when
    // the image has at most DMA_HS + DMA_WS pieces per thread; a chunk then costs one add per piece.
    constexpr int DMA_HS = 12, DMA_WS = 12;
    const int nhs = halo_pad >> 8, nws = b_pad >> 8;
    const bool pre = dma && P.tpb == 1 && nhs <= DMA_HS && nws <= DMA_WS;   // (recomputing per tile in the loop costs 70 VGPRs)
    unsigned hofs[DMA_HS], wofs[DMA_WS];
    auto compute_hofs = [&](int mtile) {
        int n0, a0, b0;
        tile_origin(mtile, n0, a0, b0);
#pragma unroll
        for (int k = 0; k < DMA_HS; ++k) {
            hofs[k] = k < nhs ? halo_voff(k * 256 + (int)threadIdx.x, 0, n0, a0, b0) : DMA_OOB;
            __builtin_amdgcn_sched_barrier(0);   // one piece at a time: interleaving all twelve costs ~100 VGPRs
        }
    };
    if (pre) {
        compute_hofs(mt_first);
#pragma unroll
        for (int k = 0; k < DMA_WS; ++k) {
            wofs[k] = k < nws ? b_voff(k * 256 + (int)threadIdx.x, 0) : DMA_OOB;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    auto dma_weights = [&](int cb, int slot) {
        char* base = (char*)smem + wbase + slot * b_pad * 16 + wave * 1024;
        if (pre) {
            const unsigned inc = (unsigned)((cb >> 2) * Cout * 16);
#pragma unroll
            for (int k = 0; k < DMA_WS; ++k)
                if (k < nws)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + k * 4096), 16, (int)(wofs[k] + inc), 0, 0, 0);
        } else {
            for (int b0s = 0; b0s < b_pad; b0s += 256)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + b0s * 16), 16,
                                                         (int)b_voff(b0s + (int)threadIdx.x, cb), 0, 0, 0);
        }
    };
    auto dma_issue = [&](int mtile, int cb, int buf) {
        char* base = (char*)smem + buf * halo_pad * 16 + wave * 1024;     // wave-uniform; lanes land at +16*lane
        if (pre) {
            const unsigned inc = (unsigned)(cb * 4);
#pragma unroll
            for (int k = 0; k < DMA_HS; ++k)
                if (k < nhs)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + k * 4096), 16, (int)(hofs[k] + inc), 0, 0, 0);
        } else {
            int n0, a0, b0;
            tile_origin(mtile, n0, a0, b0);
            for (int b0s = 0; b0s < halo_pad; b0s += 256)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + b0s * 16), 16,
                                                         (int)halo_voff(b0s + (int)threadIdx.x, cb, n0, a0, b0), 0, 0, 0);
        }
        if (!wres) dma_weights(cb, buf);
    };
    // (b) synchronous path (fallback when the double-buffered image does not fit in LDS)
    auto sync_weights = [&](int cb, int slot) {
        for (int bi = threadIdx.x; bi < b_pad; bi += 256) {
            const unsigned vo = b_voff(bi, cb);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (vo != DMA_OOB) val = BPB_GLD4((const char BPB_GLOBAL*)P.w + vo);
            *(f32x4*)((char*)smem + wbase + (slot * b_pad + bi) * 16) = val;
        }
    };
    auto sync_stage = [&](int mtile, int cb) {
        int n0, a0, b0;
        tile_origin(mtile, n0, a0, b0);
        for (int idx = threadIdx.x; idx < halo_pad; idx += 256) {
            const unsigned vo = halo_voff(idx, cb, n0, a0, b0);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (vo != DMA_OOB) val = BPB_GLD4((const char BPB_GLOBAL*)P.x + vo);
            *(f32x4*)((char*)smem + idx * 16) = val;
        }
        if (!wres) sync_weights(cb, 0);
    };

    // ---- epilogue of one M tile: C/D layout of 32x32 MFMA: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, P.N * P.Ho * P.Wo * Cout * 4, 0x00020000);
    const bpb_gcf gbias = (bpb_gcf)P.bias;
    float bias_v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias_v[nt] = (gbias && cout_l + nt * 32 < Cout) ? gbias[cout_l + nt * 32] : 0.f;
    const bool accum = P.accumulate != 0;
    auto epilogue = [&](int mtile) {
        int n0, a0, b0;
        tile_origin(mtile, n0, a0, b0);
        // opaque copy: keeps the per-row address arithmetic inside the work loop (hoisted, it costs ~100 VGPRs)
        int hq = half;
        asm volatile("" : "+v"(hq));
        double ssum[NT], ssq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ssum[nt] = 0.0;
            ssq[nt] = 0.0;
        }
        // Stores (and the loads of the accumulate mode) go through a buffer descriptor: an out-of-range offset is dropped
        // by the hardware, so tile edges need no exec-mask branches.  Offsets are 32-bit; the output is <= 1 GiB (checked
        // on the host), an invalid pixel contributes 0x80000000 and an invalid channel 0x40000000 to the offset, so every
        // invalid combination lands beyond the descriptor's range without any select.
        const unsigned PIX_OOB = 0x80000000u, CH_OOB = 0x40000000u;
        const int pstride = P.osw * Cout * 4;
        unsigned cofs[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) cofs[nt] = (cout_l + nt * 32 < Cout) ? (unsigned)(nt * 128) : CH_OOB;
#pragma unroll
        for (int mt = 0; mt < MTr; ++mt) {
            unsigned offs[16];
            if (lTW >= 2) {
                // tile width >= 4: the four accumulator rows (r & 3) of a register quad are four consecutive pixels of one
                // image row -> one address computation per quad, then constant strides
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int m = (wm * MTr + mt) * 32 + 8 * rq + 4 * hq;
                    const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
                    const int n = n0 + ti, a = a0 + th, b = b0 + tw;
                    const bool pq = (n < P.N) && (a < P.A);
                    const unsigned qoff = (unsigned)(((n * P.Ho + (a * P.osh + P.ooh)) * P.Wo + (b * P.osw + P.oow)) * Cout + cout_l) * 4u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) offs[rq * 4 + j] = (pq && b + j < P.B) ? qoff + (unsigned)(j * pstride) : PIX_OOB;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (wm * MTr + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hq;
                    const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
                    const int n = n0 + ti, a = a0 + th, b = b0 + tw;
                    const bool pv = (n < P.N) && (a < P.A) && (b < P.B);
                    offs[r] = pv ? (unsigned)(((n * P.Ho + (a * P.osh + P.ooh)) * P.Wo + (b * P.osw + P.oow)) * Cout + cout_l) * 4u : PIX_OOB;
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float old[16];
                if (accum) {   // all sixteen loads in flight before the first add
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (int)(offs[r] + cofs[nt]), 0, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = offs[r] + cofs[nt];
                    float v = acc[mt][nt][r] + bias_v[nt];
                    if (accum) v += old[r];
                    if (P.relu) v = fmaxf(v, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)off, 0, 0);
                    const double dv = off < CH_OOB ? (double)v : 0.0;
                    ssum[nt] += dv;
                    ssq[nt] += dv * dv;
                    acc[mt][nt][r] = 0.f;
                }
            }
        }
        if (P.stats) {   // per-tile BatchNorm partials, combined in fp64 (deterministic: no atomics)
            double* red = (double*)((char*)smem + redbase);   // [wave][NT*32][2]; readers of the previous tile's partials
                                                              // are separated from these writes by the work-loop barrier
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double s = ssum[nt] + __shfl_xor(ssum[nt], 32);
                double q = ssq[nt] + __shfl_xor(ssq[nt], 32);
                if (half == 0) {
                    red[((wave * NT + nt) * 32 + l31) * 2 + 0] = s;
                    red[((wave * NT + nt) * 32 + l31) * 2 + 1] = q;
                }
            }
            __syncthreads();
            if (threadIdx.x < NTC) {
                const int cw = threadIdx.x / (NT * 32);            // which wave column owns this channel
                const int nt = (threadIdx.x >> 5) % NT, c = threadIdx.x & 31;
                const int co = ntile * NTC + threadIdx.x;
                if (co < Cout) {
                    double s = 0.0, q = 0.0;
                    for (int wr = 0; wr < (4 >> lwn); ++wr) {           // fixed order over the wave rows
                        const int w = (wr << lwn) + cw;
                        s += red[((w * NT + nt) * 32 + c) * 2 + 0];
                        q += red[((w * NT + nt) * 32 + c) * 2 + 1];
                    }
                    // write-through (sc1) stores: visible to the finalising workgroup on any XCD without a release fence
                    double* gs = (double*)P.stats;
                    __hip_atomic_store(&gs[((size_t)mtile * 2 + 0) * Cout + co], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&gs[((size_t)mtile * 2 + 1) * Cout + co], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };

    // ---- work loop over (M tile, channel chunk) items
    if (wres) {
        for (int c = 0; c < nch; ++c) {
            if (dma) dma_weights(c * CK, c);
            else sync_weights(c * CK, c);
        }
    }
    if (dma && mt_first < mt_last) dma_issue(mt_first, 0, 0);
    int mtile = mt_first, chunk = 0;
    const int nwork = (mt_last - mt_first) * nch;
    for (int w = 0; w < nwork; ++w) {
        __syncthreads();   // dma: item w has landed (the barrier drains vmcnt) and the other buffer is free again
        const int cb = chunk * CK;
        int hoff = 0, wslot = wres ? chunk : 0;
        if (dma) {
            hoff = (w & 1) * halo_pad * 16;
            if (!wres) wslot = w & 1;
            if (w + 1 < nwork) {
                const bool last_chunk = chunk + 1 == nch;
                dma_issue(last_chunk ? mtile + 1 : mtile, last_chunk ? 0 : cb + CK, (w + 1) & 1);
            }
        } else {
            sync_stage(mtile, cb);
            __syncthreads();
        }
        const char* sA = (const char*)smem + hoff;
        const char* sB = (const char*)smem + wbase + wslot * b_pad * 16 + boff_lane;

        // scalar iteration state over (tap row i, tap col jj, k-group kg); no table, no global memory access
        int it_j = 0, it_kg = 0, it_c4 = 0;
        int ldsoff_s = ((P.dh0 * HWd + P.dw0) * LD) * 4, bo_s = 0;        // running offsets of the next k-group (non-C4)
        const int stepj = P.dws * LD * 4 - KG * 32, stepi = (P.dhs * HWd - St * P.dws) * LD * 4;
        auto fetch = [&](f32x4 (&a)[MTr], f32x4 (&b)[NT]) {
            int ldsoff, bo;
            if (C4) {   // Cin == 4: lanes 0-31 take tap 2j, lanes 32-63 tap 2j+1 (phantom tap -> the zero slot)
                int t = 2 * it_c4 + half;
                bo = t * tapB;
                if (t >= ntaps) t = 2 * it_c4;
                const int ti_ = t / St, tj_ = t - ti_ * St;
                ldsoff = (((P.dh0 + P.dhs * ti_) * HWd + (P.dw0 + P.dws * tj_)) * LD) * 4;
                ++it_c4;
            } else {
                ldsoff = ldsoff_s;
                bo = bo_s;
                // branch-free advance over (k-group, tap column, tap row): the weight tile is contiguous in that order
                // (tapB == KG * 2 * NTC * 16), the halo offset takes one of three strides
                bo_s += 2 * NTC * 16;
                ++it_kg;
                const bool wk = it_kg == KG;
                it_kg = wk ? 0 : it_kg;
                it_j += wk ? 1 : 0;
                const bool wj = it_j == St;
                it_j = wj ? 0 : it_j;
                ldsoff_s += 32 + (wk ? stepj : 0) + (wj ? stepi : 0);
            }
#pragma unroll
            for (int mt = 0; mt < MTr; ++mt) a[mt] = *(const f32x4*)(sA + pixoff[mt] + ldsoff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4*)(sB + bo + nt * 32 * 16);
        };
        auto mma = [&](const f32x4 (&a)[MTr], const f32x4 (&b)[NT]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MTr; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA32(a[mt][i], b[nt][i], acc[mt][nt]);
        };
        // ping-pong operand sets: the LDS reads of k-group j+1 are in flight while the 8*NT MFMAs of k-group j run
        f32x4 a0[MTr], b0[NT], a1[MTr], b1[NT];
        if (nj > 0) {                // nj == 0: empty tap set (a parity class of a strided 1x1 dgrad) -> zeros
            fetch(a0, b0);
            int j = 0;
            for (; j + 2 < nj; j += 2) {   // steady state: straight-line, counted lgkmcnt waits
                fetch(a1, b1);
                __builtin_amdgcn_sched_barrier(0);   // keep the ds_reads of group j+1 ahead of the MFMAs of group j
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j + 1 < nj) {
                fetch(a1, b1);
                mma(a0, b0);
                mma(a1, b1);
            } else {
                mma(a0, b0);
            }
        }
        if (++chunk == nch) {
            epilogue(mtile);
            chunk = 0;
            ++mtile;
        }
    }

    // ---- fused BatchNorm finalisation: the last workgroup of this problem to get here reduces the partials of ALL tiles
    // (fixed order: 8 row lanes per channel, lanes combined 0..7 -> deterministic) and writes scale / shift / saved
    // statistics / running statistics.  It is already resident, so unlike a separate tiny launch it never queues behind
    // the big kernels of the other streams for a CU.
    if (P.stats && P.bnf) {
        const BpbBnFinalizeArgs F = *P.bnf;
        // Hand-off (cdna_hip_programming.md, Guideline 16, write-through form): the partials were stored sc1, every wave
        // drains its stores, then ONE lane takes a relaxed agent-scope ticket; the last arriver issues ONE agent-scope
        // acquire and reads the partials with plain loads.  No __threadfence(): a whole-L2 write-back per workgroup made the step 70 % slower.
        volatile int* s_ticket = (volatile int*)((char*)smem + redbase);   // (dynamic LDS: no static allocation in this kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            *s_ticket = __hip_atomic_fetch_add(F.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = *s_ticket;
        __syncthreads();                       // everyone has read the ticket before the scratch is reused below
        const int nblk = ((P.n_mtiles + P.tpb - 1) / P.tpb) * P.n_ntiles;
        if (ticket == nblk - 1) {
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // ONE acquire, then plain pipelined loads
            __syncthreads();
            double* red = (double*)((char*)smem + redbase);        // [2][8][32] doubles = the 4 KiB scratch
            const double BPB_GLOBAL* gs = (const double BPB_GLOBAL*)P.stats;
            const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
            for (int cb = 0; cb < Cout; cb += 32) {
                const int c = cb + cl;
                double sm = 0.0, sq = 0.0;
                if (c < Cout)
#pragma unroll 8
                    for (int row = rl; row < P.n_mtiles; row += 8) {
                        sm += gs[((size_t)row * 2 + 0) * Cout + c];
                        sq += gs[((size_t)row * 2 + 1) * Cout + c];
                    }
                red[(0 * 8 + rl) * 32 + cl] = sm;
                red[(1 * 8 + rl) * 32 + cl] = sq;
                __syncthreads();
                if (rl == 0 && c < Cout) {
                    sm = 0.0;
                    sq = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        sm += red[(0 * 8 + i) * 32 + cl];
                        sq += red[(1 * 8 + i) * 32 + cl];
                    }
                    const double mean = sm / F.count;
                    double var = sq / F.count - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)F.eps));
                    const float g = F.gamma ? F.gamma[c] : 1.f, b = F.beta ? F.beta[c] : 0.f;
                    const float sc = g * invstd;
                    F.scale[c] = sc;
                    F.shift[c] = b - (float)mean * sc;
                    F.mean[c] = (float)mean;
                    F.invstd[c] = invstd;
                    if (F.running_mean) {
                        const double unbiased = F.count > 1.0 ? var * F.count / (F.count - 1.0) : var;
                        F.running_mean[c] = (1.f - F.momentum) * F.running_mean[c] + F.momentum * (float)mean;
                        F.running_var[c] = (1.f - F.momentum) * F.running_var[c] + F.momentum * (float)unbiased;
                    }
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) __hip_atomic_store(F.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
    }
}


// ---------------------------------------------------------------------------------------
// Weight gradient:  dW[t][ci][co] = sum_{n,a,b} x[n, a*sa+dh_t+ih0, b*sa+dw_t+iw0, ci] * dy[n,a,b,co]
// GEMM view: M = ci (32-row tile), N = co (32*NTW), K = pixels.  NHWC makes both fragments
// "K-major with unit-stride M/N": lane l supplies x[pixel k(l>>5)][ci = l&31] and dy[pixel][co = l&31]
// -- one conflict-free ds_read_b32 each, the dy fragment shared by all taps.
// One workgroup owns (tap group <= 9, ci tile, co tile, a range of 128-pixel tiles); its 4 waves split
// the pixels, accumulate in registers over the whole range, reduce through LDS and write ONE
// partial slab; bpb_wgrad_reduce_kernel sums the slabs in a fixed order (deterministic) into OIHW.
// ---------------------------------------------------------------------------------------

template <int TG, int NTW>   // TG = taps per group (9 for spatial filters, 1 for 1x1), NTW = 32-wide co sub-tiles
__global__ __launch_bounds__(256, 2) void bpb_conv_wgrad_kernel(const BpbWgradProb* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbWgradProb P = probs[pi];
    bid -= P.blk_begin;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // block id -> (split, tap group, ci tile, co tile); co fastest so neighbours share the x halo in L2
    const int cot = bid % P.n_cotiles;
    int r1 = bid / P.n_cotiles;
    const int cit = r1 % P.n_citiles;
    r1 /= P.n_citiles;
    const int tg = r1 % P.n_tapgroups;
    const int split = r1 / P.n_tapgroups;
    const int t0 = tg * TG;
    const int nt_here = min(TG, P.T - t0);
    const int Cin = P.Cin, Cout = P.Cout;
    const int ci0 = cit * 32, co0 = cot * 32 * NTW;
    const int ckc = min(32, Cin - ci0);            // valid input channels in this tile (multiple of 4)
    const int LD = P.LD, HWd = P.HW, HH = P.HH, sa = P.sa;
    const int lTW = P.lTW, lTH = P.lTH, lTI = P.lTI;
    const int TWm = (1 << lTW) - 1, THm = (1 << lTH) - 1;
    const int npix_h = (1 << lTI) * HH * HWd;
    constexpr int LDY = 32 * NTW;
    // LDS image of one pixel tile (16-byte slots): halo [halo pixel][LD/4] then dy [128][LDY/4]; regions padded to 256 slots.
    const int spp = LD >> 2;
    const int halo_slots = npix_h * spp;
    const int halo_pad = (halo_slots + 255) & ~255;
    constexpr int dy_slots = 128 * (LDY / 4);
    const int bufbytes = (halo_pad + dy_slots) * 16;
    const bool dma = P.dma != 0;

    f32x16 acc[TG][NTW];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.f;

    int tapoff[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tt = min(t0 + t, P.T - 1);
        tapoff[t] = (((tt / P.S) * HWd + (tt % P.S)) * LD) * 4;
    }

    // M-tile range of this split
    const int per = (P.n_mtiles + P.nsplit - 1) / P.nsplit;
    const int mt_begin = split * per, mt_end = min(P.n_mtiles, mt_begin + per);
    const int vpp = ckc >> 2;   // data slots per halo pixel.  When ckc < 32 the A fragment of lanes >= ckc reads past the row
                                // (pad / other pixels): those MFMA rows are never stored, rows are independent.

    auto tile_origin = [&](int mtile, int& n0, int& a0, int& b0) {
        const int tb = mtile % P.tiles_b, t2 = mtile / P.tiles_b;
        const int ta = t2 % P.tiles_a, tn = t2 / P.tiles_a;
        n0 = tn << lTI; a0 = ta << lTH; b0 = tb << lTW;
    };
    auto halo_voff = [&](int idx, int n0, int a0, int b0) -> unsigned {
        const unsigned hp = bpb_fdiv((unsigned)idx, spp, P.magic_spp);
        const int v = idx - (int)M24(hp, spp);
        if (idx >= halo_slots || v >= vpp) return 0xFFFFFFF0u;
        const unsigned t = bpb_fdiv(hp, HWd, P.magic_hw);
        const int hc = hp - M24(t, HWd);
        const unsigned ti = bpb_fdiv(t, HH, P.magic_hh);
        const int hr = t - M24(ti, HH);
        const int n = n0 + (int)ti, ih = a0 * sa + hr + P.ih0, iw = b0 * sa + hc + P.iw0;
        if (n < P.N && (unsigned)ih < (unsigned)P.Hi && (unsigned)iw < (unsigned)P.Wi)
            return ((M24(M24(n, P.Hi) + ih, P.Wi) + iw) * (unsigned)Cin + ci0 + v * 4) * 4u;
        return 0xFFFFFFF0u;
    };
    auto dy_voff = [&](int idx, int n0, int a0, int b0) -> unsigned {
        const int v = idx % (LDY / 4), m = idx / (LDY / 4);
        const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
        const int n = n0 + ti, a = a0 + th, b = b0 + tw;
        const int co = co0 + v * 4;
        if (n < P.N && a < P.A && b < P.B && co < Cout)   // Cout % 4 == 0
            return ((M24(M24(n, P.A) + a, P.B) + b) * (unsigned)Cout + co) * 4u;
        return 0xFFFFFFF0u;
    };
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)P.dy, 0, (int)P.dy_bytes, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    auto dma_issue = [&](int mtile, int buf) {
        int n0, a0, b0;
        tile_origin(mtile, n0, a0, b0);
        char* base = (char*)smem + buf * bufbytes + wave * 1024;
        for (int s0 = 0; s0 < halo_pad; s0 += 256)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + s0 * 16), 16,
                                                     (int)halo_voff(s0 + (int)threadIdx.x, n0, a0, b0), 0, 0, 0);
        for (int s0 = 0; s0 < dy_slots; s0 += 256)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr_t)(base + (halo_pad + s0) * 16), 16,
                                                     (int)dy_voff(s0 + (int)threadIdx.x, n0, a0, b0), 0, 0, 0);
    };
    auto sync_stage = [&](int mtile) {
        int n0, a0, b0;
        tile_origin(mtile, n0, a0, b0);
        for (int idx = threadIdx.x; idx < halo_pad; idx += 256) {
            const unsigned vo = halo_voff(idx, n0, a0, b0);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (vo != 0xFFFFFFF0u) val = BPB_GLD4((const char BPB_GLOBAL*)P.x + vo);
            *(f32x4*)((char*)smem + idx * 16) = val;
        }
        for (int idx = threadIdx.x; idx < dy_slots; idx += 256) {
            const unsigned vo = dy_voff(idx, n0, a0, b0);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (vo != 0xFFFFFFF0u) val = BPB_GLD4((const char BPB_GLOBAL*)P.dy + vo);
            *(f32x4*)((char*)smem + (halo_pad + idx) * 16) = val;
        }
    };

    // tile-independent LDS offsets of this lane's 16 k-steps (pixel m = wave*32 + 2*ks + half)
    int xo[16], mrow[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const int m = wave * 32 + ks * 2 + half;
        const int tw = m & TWm, th = (m >> lTW) & THm, ti = m >> (lTW + lTH);
        xo[ks] = (int)(M24(M24(M24(ti, HH) + M24(th, sa), HWd) + M24(tw, sa), LD) + l31) * 4;
        mrow[ks] = m * LDY + l31;
    }
    if (dma && mt_begin < mt_end) dma_issue(mt_begin, 0);
    for (int mtile = mt_begin; mtile < mt_end; ++mtile) {
        __syncthreads();   // dma: this tile has landed (barrier drains vmcnt) and the other buffer is free
        int bufoff = 0;
        if (dma) {
            bufoff = ((mtile - mt_begin) & 1) * bufbytes;
            if (mtile + 1 < mt_end) dma_issue(mtile + 1, (mtile + 1 - mt_begin) & 1);
        } else {
            sync_stage(mtile);
            __syncthreads();
        }
        const char* sx = (const char*)smem + bufoff;
        const float* sdy = (const float*)(sx + halo_pad * 16);
        // wave handles pixels [wave*32, wave*32+32): 16 k-steps of 2 pixels.  Straight-line, software-pipelined: the
        // TG + NTW ds_reads of step ks+1 issue before the TG*NTW MFMAs of step ks (counted lgkmcnt waits, no branches;
        // taps past the filter -- only the last group of a 7x7 -- re-read the last valid tap and are never stored).
        auto fetch = [&](int ks, float (&a)[TG], float (&b)[NTW]) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) b[nt] = sdy[mrow[ks] + nt * 32];
#pragma unroll
            for (int t = 0; t < TG; ++t) a[t] = *(const float*)(sx + xo[ks] + tapoff[t]);
        };
        auto mma = [&](const float (&a)[TG], const float (&b)[NTW]) {
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[t][nt] = MFMA32(a[t], b[nt], acc[t][nt]);
        };
        float a0[TG], b0[NTW], a1[TG], b1[NTW];
        fetch(0, a0, b0);
#pragma unroll
        for (int ks = 0; ks < 16; ks += 2) {
            fetch(ks + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 16) fetch(ks + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();

    // cross-wave reduction through LDS, one tap at a time; row = ci, col = co
    float* red = smem;   // [4 waves][16 regs][64 lanes]
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        if (t < nt_here) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][nt][r];
                __syncthreads();
                // 1024 values, 256 threads -> 4 each: (reg, lane) pairs
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = e * 256 + threadIdx.x;
                    const int r = idx >> 6, ln = idx & 63;
                    const float s = red[(0 * 16 + r) * 64 + ln] + red[(1 * 16 + r) * 64 + ln] +
                                    red[(2 * 16 + r) * 64 + ln] + red[(3 * 16 + r) * 64 + ln];
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
                    const int ci = ci0 + row, co = co0 + nt * 32 + (ln & 31);
                    if (ci < Cin && co < Cout)
                        ((bpb_gf)P.ws)[(((size_t)split * P.T + (t0 + t)) * Cin + ci) * Cout + co] = s;
                }
            }
        }
    }
}

// dW[co][ci_real][t] (OIHW, the state-dict layout) (+)= sum_split ws[split][t][ci][co]
// block = 256 threads = (256 >> lsl) consecutive slab elements (co fastest -> coalesced reads) x (1 << lsl) split lanes, lsl in
// {0, 2, 4} chosen from the number of slabs: few big slabs (256-channel layers: 2.4 MB each) are summed element-parallel at HBM
// speed, many small ones (32-channel layers: 37 KB x 256 splits) split-parallel.  Fixed summation order: deterministic.
__device__ __forceinline__ void bpb_wgrad_reduce_body(int blk, float* red, const float* __restrict__ ws, float* __restrict__ dw,
                                                      int nsplit, int T, int Cin, int Cin_real, int Cout, int accumulate, int lsl)
{
    const long total = (long)T * Cin * Cout;
    const int leb = 8 - lsl, eb = 1 << leb, nsl = 1 << lsl;
    const int el = threadIdx.x & (eb - 1), sl = threadIdx.x >> leb;
    const long e = (long)blk * eb + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // four chains: loads of four slabs in flight
    if (e < total) {
        int sp = sl;
        for (; sp + 3 * nsl < nsplit; sp += 4 * nsl) {
            s0 += ws[(size_t)sp * total + e];
            s1 += ws[(size_t)(sp + nsl) * total + e];
            s2 += ws[(size_t)(sp + 2 * nsl) * total + e];
            s3 += ws[(size_t)(sp + 3 * nsl) * total + e];
        }
        for (; sp < nsplit; sp += nsl) s0 += ws[(size_t)sp * total + e];
    }
    float s = (s0 + s1) + (s2 + s3);
    if (lsl > 0) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl == 0) {
            s = 0.f;
            for (int i = 0; i < nsl; ++i) s += red[i * eb + el];
        }
    }
    if (sl == 0 && e < total) {
        const int co = (int)(e % Cout);
        const long r = e / Cout;
        const int ci = (int)(r % Cin), t = (int)(r / Cin);
        if (ci < Cin_real) {
            const size_t o = ((size_t)co * Cin_real + ci) * T + t;
            dw[o] = accumulate ? dw[o] + s : s;
        }
    }
}

// Vector form (round 5; Cout % 4 == 0 and 16-byte aligned slabs -- every slab the planner lays out): a lane owns FOUR consecutive slab
// elements (one 16-byte load per slab: a block row is (256 >> lsl) * 16 contiguous bytes instead of (256 >> lsl) * 4), four slabs in flight
// per lane.  Same split-lane scheme and the same fixed order per element as the scalar body; blocks cover 4x the elements.
// profiles/r05_ab_wgrad_reduce_vec.txt: 3.2 -> 4.6 TB/s over the 28 grouped launches of the HRNet-W32 step (0.90 -> 0.63 ms).
__device__ __forceinline__ void bpb_wgrad_reduce_body4(int blk, f32x4* red, const float* __restrict__ ws, float* __restrict__ dw,
                                                       int nsplit, int T, int Cin, int Cin_real, int Cout, int accumulate, int lsl)
{
    const long total = (long)T * Cin * Cout;
    const int leb = 8 - lsl, eb = 1 << leb, nsl = 1 << lsl;
    const int el = threadIdx.x & (eb - 1), sl = threadIdx.x >> leb;
    const long e = ((long)blk * eb + el) * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (e < total) {
        const float* __restrict__ b = ws + e;
        int sp = sl;
        for (; sp + 7 * nsl < nsplit; sp += 8 * nsl) {          // eight slabs in flight while a lane has that many left
            const f32x4 v0 = *(const f32x4*)(b + (size_t)sp * total);
            const f32x4 v1 = *(const f32x4*)(b + (size_t)(sp + nsl) * total);
            const f32x4 v2 = *(const f32x4*)(b + (size_t)(sp + 2 * nsl) * total);
            const f32x4 v3 = *(const f32x4*)(b + (size_t)(sp + 3 * nsl) * total);
            const f32x4 v4 = *(const f32x4*)(b + (size_t)(sp + 4 * nsl) * total);
            const f32x4 v5 = *(const f32x4*)(b + (size_t)(sp + 5 * nsl) * total);
            const f32x4 v6 = *(const f32x4*)(b + (size_t)(sp + 6 * nsl) * total);
            const f32x4 v7 = *(const f32x4*)(b + (size_t)(sp + 7 * nsl) * total);
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
            s0 += v4;
            s1 += v5;
            s2 += v6;
            s3 += v7;
        }
        for (; sp + 3 * nsl < nsplit; sp += 4 * nsl) {
            const f32x4 v0 = *(const f32x4*)(b + (size_t)sp * total);
            const f32x4 v1 = *(const f32x4*)(b + (size_t)(sp + nsl) * total);
            const f32x4 v2 = *(const f32x4*)(b + (size_t)(sp + 2 * nsl) * total);
            const f32x4 v3 = *(const f32x4*)(b + (size_t)(sp + 3 * nsl) * total);
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; sp < nsplit; sp += nsl) s0 += *(const f32x4*)(b + (size_t)sp * total);
    }
    f32x4 s = (s0 + s1) + (s2 + s3);
    if (lsl > 0) {
        red[threadIdx.x] = s;
        __syncthreads();
        if (sl == 0) {
            s = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < nsl; ++i) s += red[i * eb + el];
        }
    }
    if (sl == 0 && e < total) {
        const int co = (int)(e % Cout);             // co .. co + 3 share (ci, t): Cout % 4 == 0
        const long r = e / Cout;
        const int ci = (int)(r % Cin), t = (int)(r / Cin);
        if (ci < Cin_real) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = ((size_t)(co + j) * Cin_real + ci) * T + t;
                dw[o] = accumulate ? dw[o] + s[j] : s[j];
            }
        }
    }
}

static inline int bpb_wgrad_reduce_lsl(int nsplit) { return nsplit <= 4 ? 0 : nsplit <= 32 ? 2 : 4; }
// field pad_ of a record / what the single entry point derives: lsl | 256 when the vector form applies
static inline int bpb_wgrad_reduce_form(int nsplit, int Cout, const float* ws)
{
    return bpb_wgrad_reduce_lsl(nsplit) | ((Cout % 4 == 0 && ((uintptr_t)ws & 15) == 0) ? 256 : 0);
}
static inline int bpb_wgrad_reduce_blocks(long total, int form)
{
    return (int)bpb_cdiv(total, (long)(256 >> (form & 255)) << ((form & 256) ? 2 : 0));
}

__global__ __launch_bounds__(256) void bpb_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int T,
                                                               int Cin, int Cin_real, int Cout, int accumulate, int form)
{
    __shared__ f32x4 red[256];
    if (form & 256) bpb_wgrad_reduce_body4(blockIdx.x, red, ws, dw, nsplit, T, Cin, Cin_real, Cout, accumulate, form & 255);
    else bpb_wgrad_reduce_body(blockIdx.x, (float*)red, ws, dw, nsplit, T, Cin, Cin_real, Cout, accumulate, form & 255);
}

// grouped: the slab reductions of the convolutions of one module step in one launch (blk_begin prefix)
__global__ __launch_bounds__(256) void bpb_wgrad_reduce_multi_kernel(const BpbWgradReduceDesc* __restrict__ descs, BpbBlkBegins bb)
{
    __shared__ f32x4 red[256];
    const int di = bpb_find_problem(bb, (int)blockIdx.x);
    const BpbWgradReduceDesc D = descs[di];
    if (D.pad_ & 256)
        bpb_wgrad_reduce_body4(blockIdx.x - D.blk_begin, red, D.ws, D.dw, D.nsplit, D.T, D.Cin, D.Cin_real, D.Cout, D.accumulate, D.pad_ & 255);
    else
        bpb_wgrad_reduce_body(blockIdx.x - D.blk_begin, (float*)red, D.ws, D.dw, D.nsplit, D.T, D.Cin, D.Cin_real, D.Cout, D.accumulate, D.pad_ & 255);
}

// ---------------------------------------------------------------------------------------
// Weight packing, one launch for every conv of a network (descriptor table):
//   fwd   wf[t][ci/4][co][4]  = W[co][ci][t]            (ci zero-padded to Cin_pad)
//   dgrad wd[t][co/4][ci][4]  = W[co][ci][t]            (only when wd != nullptr)
// ---------------------------------------------------------------------------------------

// A workgroup moves a tile of 16 output channels x IB input channels x all T taps through LDS: the reads run along W's rows (one
// output channel's IB * T values are contiguous), the writes along wf's output-channel axis (64 consecutive floats per (tap, channel
// quad)) and along wd's input-channel axis (4 * IB floats per (tap, output-channel quad)) -- the first form (one thread per packed
// element, 4-byte gathers at a stride of T floats) took 210 us per step for HRNet-W32's 28.5 M weights, three times the HBM time.
// IB comes with the descriptor (graph.py: IB * T <= 196, so that the block counts of both sides agree by construction).
constexpr int PACK_CB = 16, PACK_ROW = 197;

__global__ __launch_bounds__(256) void bpb_pack_weights_kernel(const BpbPackProb* __restrict__ probs, int nprobs)
{
    __shared__ float tile[PACK_CB * PACK_ROW];
    int bid = blockIdx.x;
    int lo = 0, hi = nprobs - 1;   // binary search over blk_begin (hundreds of convs)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (probs[mid].blk_begin <= bid) lo = mid; else hi = mid - 1;
    }
    const BpbPackProb P = probs[lo];
    bid -= P.blk_begin;
    const int T = P.T, IB = P.IB, Cin = P.Cin, Cinp = P.Cin_pad, Cout = P.Cout;
    const int tiles_ci = (Cinp + IB - 1) / IB;
    const int c0 = (bid / tiles_ci) * PACK_CB, i0 = (bid % tiles_ci) * IB;
    const int nci = min(IB, Cin - i0);                 // real input channels of this tile (<= 0: padding only)
    const int ncp = min(IB, Cinp - i0);                // packed (zero-padded) input channels of this tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rl = max(nci, 0) * T;
    for (int r = wave; r < PACK_CB; r += 4) {
        const int co = c0 + r;
        if (co < Cout) {
            const float* src = P.w + ((size_t)co * Cin + i0) * T;
            for (int c = lane; c < rl; c += 64) tile[r * PACK_ROW + c] = src[c];
        }
    }
    __syncthreads();
    // F(2,3) packing (P.wino bit 0: wf, bit 1: wd; T == 9): 12 "taps" [column tap s][position q] with the row-transformed filters
    //   q = 0: g0   1: (g0 + g1 + g2) / 2   2: (g0 - g1 + g2) / 2   3: g2        (g_r = W[..][r][s]; csrc/conv_s1.hip, WINO)
    // the data-gradient side transforms the MIRRORED filter g'_r = W[..][2 - r][2 - s'] (its kernel runs with wflip = 0)
    auto wino_val = [&](int co_l, int cil, int t12, bool mirror) {
        const int s_ = t12 >> 2, q = t12 & 3;
        const float* g = tile + co_l * PACK_ROW + cil * 9 + (mirror ? 2 - s_ : s_);
        const float g0 = mirror ? g[6] : g[0], g1 = g[3], g2 = mirror ? g[0] : g[6];
        return q == 0 ? g0 : q == 3 ? g2 : q == 1 ? 0.5f * ((g0 + g1) + g2) : 0.5f * ((g0 - g1) + g2);
    };
    if (P.wino & 1) {
        const int co_l = lane >> 2, e = lane & 3, co = c0 + co_l;
        const float sc = (P.scale && co < Cout) ? P.scale[co] : 1.f;
        const int nq = ncp >> 2;
        for (int pr = wave; pr < 12 * nq; pr += 4) {
            const int t = pr / nq, ql = pr - t * nq;
            const int cil = ql * 4 + e;
            if (co < Cout) {
                const float v = cil < nci ? wino_val(co_l, cil, t, false) * sc : 0.f;
                P.wf[(((size_t)t * (Cinp >> 2) + (i0 >> 2) + ql) * Cout + co) * 4 + e] = v;
            }
        }
    } else
    // ---- forward layout wf[t][ci/4][co][4]: one (tap, channel quad) = 16 co x 4 e = 64 consecutive floats
    {
        const int co_l = lane >> 2, e = lane & 3, co = c0 + co_l;
        const float sc = (P.scale && co < Cout) ? P.scale[co] : 1.f;
        const int nq = ncp >> 2;
        for (int pr = wave; pr < T * nq; pr += 4) {
            const int t = pr / nq, ql = pr - t * nq;
            const int cil = ql * 4 + e;
            if (co < Cout) {
                const float v = cil < nci ? tile[co_l * PACK_ROW + cil * T + t] * sc : 0.f;
                P.wf[(((size_t)t * (Cinp >> 2) + (i0 >> 2) + ql) * Cout + co) * 4 + e] = v;
            }
        }
    }
    // ---- data-gradient layout wd[t][co/4][ci][4]: one (tap, output-channel quad) = ncp ci x 4 e' = 4 * ncp consecutive floats
    if (P.wd) {
        const int per = ncp * 4;
        const bool wdw = (P.wino & 2) != 0;
        for (int pr = wave; pr < (wdw ? 12 : T) * (PACK_CB / 4); pr += 4) {
            const int t = pr >> 2, cq = pr & 3;
            for (int c = lane; c < per; c += 64) {
                const int cil = c >> 2, e = c & 3, co_l = cq * 4 + e, co = c0 + co_l;
                if (co < Cout) {
                    const float v = cil >= nci ? 0.f : wdw ? wino_val(co_l, cil, t, true) : tile[co_l * PACK_ROW + cil * T + t];
                    P.wd[(((size_t)t * (Cout >> 2) + (c0 >> 2) + cq) * Cinp + i0 + cil) * 4 + e] = v;
                }
            }
        }
    }
}

// ------------------------------------ C ABI ------------------------------------------
static int conv_lds_bytes(const BpbConvProb& p)
{
    const int npix = (1 << p.lTI) * p.HH * p.HW;
    const int ntaps = p.Rt * p.St + (p.Cin == 4 ? 1 : 0);
    const int halo_pad = (npix * (p.LD / 4) + 255) & ~255;
    const int b_pad = (ntaps * (p.CK / 4) * ((p.nt * 32) << p.lwn) + 255) & ~255;
    const int nbuf = p.dma ? 2 : 1;                      // two halo images for the DMA pipeline
    const int nwb = p.wres ? p.Cin / p.CK : nbuf;        // resident weight tiles, or one per buffer
    return (nbuf * halo_pad + nwb * b_pad) * 16 + 4096;  // + BatchNorm partials scratch
}

extern "C" {

// One-time per-process setup: allow the conv kernels to use up to 160 KiB of dynamic LDS.
int bpb_conv_init(void)
{
#define BPB_ATTR(K)                                                                                          \
    {                                                                                                        \
        hipError_t e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e != hipSuccess) return bpb_set_error((int)e, "bpb_conv_init: %s", hipGetErrorString(e));            \
    }
    BPB_ATTR((bpb_conv_igemm_kernel<1, false, 1>))
    BPB_ATTR((bpb_conv_igemm_kernel<2, false, 1>))
    BPB_ATTR((bpb_conv_igemm_kernel<1, true, 1>))
    BPB_ATTR((bpb_conv_igemm_kernel<2, true, 1>))
    BPB_ATTR((bpb_conv_igemm_kernel<1, false, 2>))
    BPB_ATTR((bpb_conv_igemm_kernel<2, false, 2>))
    BPB_ATTR((bpb_conv_igemm_kernel<1, true, 2>))
    BPB_ATTR((bpb_conv_igemm_kernel<2, true, 2>))
    BPB_ATTR((bpb_conv_wgrad_kernel<1, 1>))
    BPB_ATTR((bpb_conv_wgrad_kernel<9, 1>))
    BPB_ATTR((bpb_conv_wgrad_kernel<1, 2>))
    BPB_ATTR((bpb_conv_wgrad_kernel<1, 4>))
#undef BPB_ATTR
    return 0;
}

// Launch a group of conv problems (descriptors already in device memory, `h_probs` is the host copy
// used for validation and grid sizing).  Replaces aten::conv2d / conv backward-input on the path.
int bpb_conv_igemm(const BpbConvProb* d_probs, const BpbConvProb* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_igemm: nprobs=%d out of range", nprobs);
    int nblk = 0, lds = 0, nt = 0, mt = 0;
    const bool c4 = h_probs[0].Cin == 4;
    for (int i = 0; i < nprobs; ++i) {
        const BpbConvProb& p = h_probs[i];
        BPB_REQUIRE(p.Cin == 4 || p.Cin % 8 == 0, "bpb_conv_igemm: Cin=%d must be 4 or a multiple of 8", p.Cin);
        BPB_REQUIRE(p.CK >= 4 && p.CK <= 32 && (p.CK & (p.CK - 1)) == 0 && p.Cin % p.CK == 0 && (p.Cin == 4 || p.CK >= 8),
                    "bpb_conv_igemm: bad channel chunk CK=%d for Cin=%d", p.CK, p.Cin);
        BPB_REQUIRE(p.LD >= p.CK && p.LD % 4 == 0, "bpb_conv_igemm: bad LDS pitch %d", p.LD);
        BPB_REQUIRE(p.x_bytes > 0 && p.w_bytes > 0 && p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u,
                    "bpb_conv_igemm: tensors addressed through a buffer descriptor must be < 2 GiB");
        BPB_REQUIRE((double)p.N * p.Ho * p.Wo * p.Cout * 4.0 <= 1073741824.0, "bpb_conv_igemm: output tensor must be <= 1 GiB");
        BPB_REQUIRE((p.mt_r == 1 || p.mt_r == 2) && (p.lwn == 0 || p.lwn == 1), "bpb_conv_igemm: bad tile shape mt=%d lwn=%d", p.mt_r, p.lwn);
        BPB_REQUIRE((1 << (p.lTI + p.lTH + p.lTW)) == (4 >> p.lwn) * p.mt_r * 32, "bpb_conv_igemm: M tile / wave layout mismatch");
        BPB_REQUIRE(p.nt == 1 || p.nt == 2, "bpb_conv_igemm: nt=%d", p.nt);
        BPB_REQUIRE(p.Rt >= 0 && p.St >= 0 && p.Rt * p.St <= 64, "bpb_conv_igemm: tap grid %dx%d", p.Rt, p.St);
        BPB_REQUIRE(i == 0 || (p.Cin == 4) == (h_probs[0].Cin == 4), "bpb_conv_igemm: Cin==4 problems need their own group");
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_igemm: blk_begin mismatch");
        BPB_REQUIRE(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "bpb_conv_igemm: x/w must be 16-byte aligned");
        const int this_nt = p.nt;
        BPB_REQUIRE(nt == 0 || nt == this_nt, "bpb_conv_igemm: mixed N-tile widths in one group");
        nt = this_nt;
        BPB_REQUIRE(mt == 0 || mt == p.mt_r, "bpb_conv_igemm: mixed M sub-tile counts in one group");
        mt = p.mt_r;
        BPB_REQUIRE(p.n_ntiles == bpb_cdiv(p.Cout, (32 * nt) << p.lwn), "bpb_conv_igemm: n_ntiles mismatch");
        BPB_REQUIRE(p.tpb >= 1 && (p.wres == 0 || p.wres == 1), "bpb_conv_igemm: tpb=%d wres=%d", p.tpb, p.wres);
        nblk += bpb_cdiv(p.n_mtiles, p.tpb) * p.n_ntiles;
        const int l = conv_lds_bytes(p);
        lds = l > lds ? l : lds;
    }
    BPB_REQUIRE(lds <= 160 * 1024, "bpb_conv_igemm: halo tile needs %d B of LDS", lds);
    if (nblk == 0) return 0;
#define BPB_CONV_LAUNCH(NT, C4, MT) \
    hipLaunchKernelGGL((bpb_conv_igemm_kernel<NT, C4, MT>), dim3(nblk), dim3(256), lds, stream, d_probs, bpb_blk_begins(h_probs, nprobs))
#define BPB_CONV_LAUNCH_MT(NT, C4) \
    do { if (mt == 1) { BPB_CONV_LAUNCH(NT, C4, 1); } else { BPB_CONV_LAUNCH(NT, C4, 2); } } while (0)
    if (c4 && nt == 1) { BPB_CONV_LAUNCH_MT(1, true); }
    else if (c4) { BPB_CONV_LAUNCH_MT(2, true); }
    else if (nt == 1) { BPB_CONV_LAUNCH_MT(1, false); }
    else { BPB_CONV_LAUNCH_MT(2, false); }
#undef BPB_CONV_LAUNCH_MT
#undef BPB_CONV_LAUNCH
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_conv_wgrad(const BpbWgradProb* d_probs, const BpbWgradProb* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_wgrad: nprobs=%d out of range", nprobs);
    int nblk = 0, lds = 0, ntw = 0;
    for (int i = 0; i < nprobs; ++i) {
        const BpbWgradProb& p = h_probs[i];
        BPB_REQUIRE(p.Cin % 4 == 0 && p.Cout % 4 == 0, "bpb_conv_wgrad: Cin/Cout must be multiples of 4");
        BPB_REQUIRE(p.lTI + p.lTH + p.lTW == 7, "bpb_conv_wgrad: M tile must be 128 pixels");
        BPB_REQUIRE(p.LD % 4 == 0 && p.LD >= (p.Cin < 32 ? p.Cin : 32), "bpb_conv_wgrad: bad LDS pitch");
        BPB_REQUIRE(p.T >= 1 && p.S >= 1 && p.T % p.S == 0, "bpb_conv_wgrad: T=%d S=%d", p.T, p.S);
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_wgrad: blk_begin mismatch");
        const int this_ntw = p.ntw;
        BPB_REQUIRE(this_ntw == 1 || (p.T == 1 && (this_ntw == 2 || this_ntw == 4)), "bpb_conv_wgrad: ntw=%d (T=%d)", this_ntw, p.T);
        BPB_REQUIRE(ntw == 0 || ntw == this_ntw, "bpb_conv_wgrad: mixed tile widths in one group");
        BPB_REQUIRE(i == 0 || (p.T == 1) == (h_probs[0].T == 1), "bpb_conv_wgrad: 1x1 and spatial filters cannot share a group");
        ntw = this_ntw;
        BPB_REQUIRE(p.n_cotiles == bpb_cdiv(p.Cout, 32 * ntw) && p.n_citiles == bpb_cdiv(p.Cin, 32) &&
                    p.n_tapgroups == (p.T == 1 ? 1 : bpb_cdiv(p.T, 9)), "bpb_conv_wgrad: tile counts mismatch");
        nblk += p.nsplit * p.n_tapgroups * p.n_citiles * p.n_cotiles;
        const int npix = (1 << p.lTI) * p.HH * p.HW;
        const int halo_pad = (npix * (p.LD / 4) + 255) & ~255;
        int l = (halo_pad + 128 * 8 * ntw) * 16 * (p.dma ? 2 : 1);
        if (l < 16384) l = 16384;
        BPB_REQUIRE(p.x_bytes > 0 && p.dy_bytes > 0 && p.x_bytes < 0xFFFFFFF0u && p.dy_bytes < 0xFFFFFFF0u,
                    "bpb_conv_wgrad: tensors addressed through a buffer descriptor must be < 4 GiB");
        lds = l > lds ? l : lds;
    }
    BPB_REQUIRE(lds <= 160 * 1024, "bpb_conv_wgrad: needs %d B of LDS", lds);
    if (nblk == 0) return 0;
#define BPB_WG_LAUNCH(TG, NTW) \
    hipLaunchKernelGGL((bpb_conv_wgrad_kernel<TG, NTW>), dim3(nblk), dim3(256), lds, stream, d_probs, bpb_blk_begins(h_probs, nprobs))
    if (ntw == 1 && h_probs[0].T == 1) { BPB_WG_LAUNCH(1, 1); }
    else if (ntw == 1) { BPB_WG_LAUNCH(9, 1); }
    else if (ntw == 2) { BPB_WG_LAUNCH(1, 2); }
    else { BPB_WG_LAUNCH(1, 4); }
#undef BPB_WG_LAUNCH
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_wgrad_reduce(const float* ws, float* dw, int nsplit, int T, int Cin, int Cin_real, int Cout, int accumulate,
                     hipStream_t stream)
{
    const long total = (long)T * Cin * Cout;
    BPB_REQUIRE(total > 0 && nsplit >= 1 && Cin_real <= Cin, "bpb_wgrad_reduce: empty problem");
    const int form = bpb_wgrad_reduce_form(nsplit, Cout, ws);
    const int grid = bpb_wgrad_reduce_blocks(total, form);
    hipLaunchKernelGGL(bpb_wgrad_reduce_kernel, dim3(grid), dim3(256), 0, stream, ws, dw, nsplit, T, Cin, Cin_real, Cout,
                       accumulate, form);
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_wgrad_reduce_multi(const BpbWgradReduceDesc* d_descs, const BpbWgradReduceDesc* h_descs, int n, int total_blocks,
                           hipStream_t stream)
{
    BPB_REQUIRE(n >= 1 && n <= 16, "bpb_wgrad_reduce_multi: n=%d", n);
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const long total = (long)h_descs[i].T * h_descs[i].Cin * h_descs[i].Cout;
        BPB_REQUIRE(total > 0 && h_descs[i].nsplit >= 1 && h_descs[i].Cin_real <= h_descs[i].Cin && h_descs[i].blk_begin == blk,
                    "bpb_wgrad_reduce_multi: record %d", i);
        // any split-lane count 1 .. 16 is valid (the planner's choice); the vector form needs aligned slabs and Cout % 4 == 0
        const int form = bpb_wgrad_reduce_form(h_descs[i].nsplit, h_descs[i].Cout, h_descs[i].ws);
        BPB_REQUIRE((h_descs[i].pad_ & 255) <= 4 && (h_descs[i].pad_ & ~(255 | (form & 256))) == 0,
                    "bpb_wgrad_reduce_multi: record %d: form (field pad_ = lsl 0..4 [| 256 for 16-byte slabs with Cout %% 4 == 0]) is %d", i,
                    h_descs[i].pad_);
        blk += bpb_wgrad_reduce_blocks(total, h_descs[i].pad_);
    }
    BPB_REQUIRE(blk == total_blocks, "bpb_wgrad_reduce_multi: block count mismatch");
    hipLaunchKernelGGL(bpb_wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(256), 0, stream, d_descs, bpb_blk_begins(h_descs, n));
    BPB_LAUNCH_OK();
    return 0;
}

int bpb_pack_weights(const BpbPackProb* d_probs, int nprobs, int total_blocks, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && total_blocks >= 1, "bpb_pack_weights: empty");
    // (the descriptors live in device memory only; graph.py computes IB and the block counts with the kernel's formula:
    //  IB a multiple of 4, IB * T <= 196, blocks = ceil(Cout / 16) * ceil(Cin_pad / IB), Cout % 4 == 0 where wd is packed)
    hipLaunchKernelGGL(bpb_pack_weights_kernel, dim3(total_blocks), dim3(256), 0, stream, d_probs, nprobs);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
