// Pointwise (1x1, stride 1) convolution = a plain [pixels x Cin] . [Cin x Cout] GEMM on the NHWC tensors: the layer-1 Bottleneck
// convolutions of both backbones (64->64, 256->64, 64->256 at 64x32: 131 072 pixels) and ResNet-50's layer-2 ones, forward and data
// gradient:  torchreid/models/hrnet.py:104-110 (Bottleneck conv1 / conv3), :184 (downsample); torchreid/models/resnet.py:119-127.
//
// Why a kernel of its own (profiles/r04_trace_1x1/, r04_pmc_1x1_cold.txt): on bpb_conv_s1_kernel<.,.,1,.> these launches ran at
// 47-57 TFLOP/s and 2.0-2.4 TB/s -- a workgroup lived for ONE 128-pixel tile, i.e. two to eight channel chunks: descriptor + halo /
// weight offset arithmetic 6 400 cycles, the wait for the cold first chunk 5 000, the epilogue 5 000, around 7 400 cycles of MFMAs,
// with a workgroup barrier per chunk.  A 1x1 convolution has no image geometry at all, so here
//   * the weight slice [Cin][NTC] of a workgroup is loaded into LDS ONCE (buffer_load ... lds) and stays for its whole life;
//   * the workgroup is persistent and its four waves are AUTONOMOUS: a wave walks its own sequence of 32-pixel tiles, loads its A
//     fragments straight from global memory into registers in the MFMA operand layout (lane (row, half) reads the 16 bytes
//     x[row][8 kg + 4 half ..+3]: linear offsets, rows beyond the tensor read zero through the buffer descriptor), double-buffered
//     against the MFMAs of the previous 64-channel chunk -- no LDS staging of x, NO barrier after the weight load;
//   * BatchNorm partials accumulate per wave over all of its tiles (LDS scratch) and leave as ONE row per workgroup.
// K = 64 (Cin == 64): the A registers of a tile serve every 64-channel pass over the workgroup's NTC <= 256 output channels (x is
// read exactly once); K = 128 / 192 / 256: NTC = 64, the accumulators run over the chunks (two-level sums over 32-channel sub-sums
// like bpb_conv_s1's 32-channel chunks).
// Epilogue = that of bpb_conv_s1 (bias, ReLU, residual operand, accumulate, BatchNorm statistics in fp64, BatchNorm-backward
// partials of the data-gradient launches).
#include "bpb_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define M24(a, b) __umul24((unsigned)(a), (unsigned)(b))

template <bool KC1>      // KC1: Cin == 64 (one chunk; several 64-channel passes per tile) -- else NTC == 64 (one pass, Cin / 64 chunks)
__global__ __launch_bounds__(256, 2) void bpb_conv_pw_kernel(const BpbConvPwProb* __restrict__ probs, BpbBlkBegins bb)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 2;
    int bid = blockIdx.x;
    const int pi = bpb_find_problem(bb, bid);
    const BpbConvPwProb P = probs[pi];
    bid -= P.blk_begin;
    if (P.xr) {          // XCD-aware map (see bpb_conv_s1): the column blocks of one pixel group share an L2
        const int nb = P.n_mtiles << P.l_ntiles, q = nb >> 3, r = nb & 7, f = bid & 7;
        bid = f * q + min(f, r) + (bid >> 3);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ncol = bid & ((1 << P.l_ntiles) - 1), g = bid >> P.l_ntiles;
    const int Cin = P.Cin, Cout = P.Cout, NTC = P.NTC;
    const int KC = KC1 ? 1 : Cin >> 6;
    const int npass = KC1 ? NTC >> 6 : 1;
    const int wslots = (Cin >> 2) * NTC;                      // 16-byte slots of the weight slice
    double* red = (double*)((char*)smem + wslots * 16);       // [wave][NTC][2] BatchNorm partials of this workgroup
    const bool do_stats = P.stats != nullptr;

    // ---- the weight slice -> LDS, once: slot (q, n) <- w[(q * Cout + ncol * NTC + n)][0..3]
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, 0, (int)P.w_bytes, 0x00020000);
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        const int lntc = NTC == 64 ? 6 : NTC == 128 ? 7 : 8;
        const int npieces = wslots >> 8;                      // NTC >= 64, Cin >= 64: a multiple of 256 slots
        for (int k = 0; k < npieces; ++k) {
            const int idx = k * 256 + (int)threadIdx.x;
            const int n = idx & (NTC - 1), q = idx >> lntc;
            const unsigned vo = ((unsigned)(q * Cout + ncol * NTC + n)) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)((char*)smem + k * 4096 + wave * 1024), 16, (int)vo, 0, 0, 0);
        }
        if (do_stats)
            for (int i = threadIdx.x; i < 4 * NTC * 2; i += 256) red[i] = 0.0;
    }
    // per-channel epilogue constants of this workgroup's columns -> LDS: [bias | mean | invstd][NTC].  (Read with global_load inside
    // the item loop they made hipcc wait vmcnt(0) at the first MFMA of every pass -- behind the previous pass's 32 stores; the ISA
    // shows counted waits once no global_load is left between the MFMA phases.)
    float* cvals = (float*)(red + 4 * NTC * 2);
    const bool bn_bwd = do_stats && P.bnb != nullptr;
    const float* bn_out = nullptr;
    const float* bn_src = nullptr;
    {
        auto uniform_ptr = [](const void* p_) {
            const unsigned long long u = (unsigned long long)p_;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return (const float*)(((unsigned long long)hi << 32) | lo);
        };
        bpb_gcf mean_p = nullptr, invstd_p = nullptr;
        if (bn_bwd) {
            const BpbS1BnBwd* bp = P.bnb;
            bn_out = uniform_ptr(bp->out);
            bn_src = uniform_ptr(bp->src);
            mean_p = (bpb_gcf)uniform_ptr(bp->mean);
            invstd_p = (bpb_gcf)uniform_ptr(bp->invstd);
        }
        const bpb_gcf gbias = (bpb_gcf)P.bias;
        if ((int)threadIdx.x < NTC) {
            const int co = ncol * NTC + (int)threadIdx.x;
            cvals[threadIdx.x] = gbias ? gbias[co] : 0.f;
            cvals[NTC + threadIdx.x] = bn_bwd ? mean_p[co] : 0.f;
            cvals[2 * NTC + threadIdx.x] = bn_bwd ? invstd_p[co] : 0.f;
        }
    }

    // ---- this wave's tiles: t = g * 4 + wave, + n_mtiles * 4, ...
    const int ntiles = P.ntiles32;
    const int tstride = P.n_mtiles * 4;
    int t = g * 4 + wave;
    // A operands: `cur` feeds the MFMAs of the current (tile, chunk) item while the loads of the next item land in `nxt` (issued
    // behind the item's first MFMA group).  The hand-over cur <- nxt sits between the item's last MFMA and its epilogue, pinned by
    // an empty asm that consumes the registers: the compiler places its own s_waitcnt there (it knows every outstanding operation:
    // no hand-counted waits, and no chance of a register copy of an in-flight load -- which is what inline-asm loads with counted
    // waits produced around the loop-carried buffers) -- the prefetch has had the whole MFMA phase to land.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, 0, (int)P.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto rowoff = [&](int tile) -> unsigned {                 // byte offset of this lane's row of x (its k half added), or out of range
        const int p = tile * 32 + l31;
        return p < P.P ? M24(p, Cin * 4) + (unsigned)(half * 16) : OOB;
    };
    f32x4 cur[8], nxt[8];                                     // 8 k-groups of a 64-channel chunk each
#define PW_LOAD(A, ROFF, CHUNK)                                                                                                       \
    _Pragma("unroll") for (int kg = 0; kg < 8; ++kg)(A)[kg] =                                                                         \
        __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ROFF), (CHUNK) * 256 + kg * 32, 0))
    unsigned roff = rowoff(t);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) cur[kg] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < ntiles) { PW_LOAD(cur, roff, 0); }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the weight DMA; the first tile's A loads ride along)
    __syncthreads();                                          // the ONLY barrier before the statistics hand-over at the very end
    // (the first tile's operands are consumed HERE as far as the compiler's wait bookkeeping goes: with `cur` pending on the loop's
    //  entry edge it would wait vmcnt(0) at the first MFMA of EVERY iteration -- right behind the prefetch it has just issued)
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) asm volatile("" : "+v"(cur[kg]));

    // ---- epilogue state
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)P.y, 0, (int)P.y_bytes, 0x00020000);
    const bool accum = P.accumulate != 0 || P.res != nullptr, relu = P.relu != 0;
    const __amdgpu_buffer_rsrc_t rold = P.res ? __builtin_amdgcn_make_buffer_rsrc((void*)P.res, 0, (int)P.y_bytes, 0x00020000) : ry;
    const int pstride = Cout * 4;
    const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc((void*)(bn_src ? bn_src : P.y), 0, (int)P.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbo = __builtin_amdgcn_make_buffer_rsrc((void*)(bn_out ? bn_out : P.y), 0, (int)P.y_bytes, 0x00020000);

    f32x16 acc[NT];
    const char* lds = (const char*)smem;

    // one (tile, chunk) item: MFMAs of every pass on `cur`, the epilogue after the last chunk; `have_next`: `nxt` holds loads in flight
    auto item = [&](int tile, int c, bool have_next, unsigned rn, int cn) {
        for (int pass = 0; pass < npass; ++pass) {
            f32x16 cacc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cacc[nt][r] = 0.f;
            int bptr = ((c * 16 + half) * NTC + pass * 64 + l31) * 16;
            const int bstride = 2 * NTC * 16;
            f32x4 fb[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[0][nt] = *(const f32x4*)(lds + bptr + nt * 512);
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                if (kg + 1 < 8) {
                    bptr += bstride;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) fb[(kg + 1) & 1][nt] = *(const f32x4*)(lds + bptr + nt * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) cacc[nt] = MFMA32(cur[kg][i], fb[kg & 1][nt][i], cacc[nt]);
                __builtin_amdgcn_sched_barrier(0);
                if (kg == 3 || kg == 7) {
                    // two-level summation in 32-channel sub-sums, the granularity of bpb_conv_s1's 32-channel chunks (a single fp32
                    // chain over 64 products measurably raised the gradient noise of the most chaotic golden fixture)
                    if (kg == 3 && (KC1 || c == 0)) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = cacc[nt];
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[nt][r] += cacc[nt][r];
                    }
                    if (kg == 3) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) cacc[nt][r] = 0.f;
                    }
                }
                if (kg == 0 && pass == 0 && have_next) {
                    // The next item's loads go out BEHIND the first MFMA group: hipcc (ROCm 7.2) waits vmcnt(0) at the first MFMA of
                    // a loop body whose back edge carries stores (measured in the ISA, whatever the operands), so a prefetch issued in
                    // front of it would be drained on the spot; from here it has the rest of the item to land.
                    PW_LOAD(nxt, rn, cn);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pass + 1 == npass) {     // the last MFMA of the item has been issued: the next item's operands take over
                if (have_next) {
#pragma unroll
                    for (int kg = 0; kg < 8; ++kg) {
                        cur[kg] = nxt[kg];
                        asm volatile("" : "+v"(cur[kg]));
                    }
                }
            }
            if (c + 1 < KC) continue;
            // ---- epilogue of (tile, pass).  C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * half
            const int cbase = ncol * NTC + pass * 64 + l31;
            unsigned offs[16];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int p0 = tile * 32 + 8 * rq + 4 * half;
                const unsigned qoff = M24(p0, pstride) + (unsigned)(cbase * 4);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) offs[rq * 4 + jj] = p0 + jj < P.P ? qoff + (unsigned)(jj * pstride) : OOB;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cl = pass * 64 + nt * 32 + l31;
                const float bias_v = cvals[cl];
                float old[16];
                if (accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rold, (int)(offs[r] + nt * 128), 0, 0));
                }
                double ssum = 0.0, ssq = 0.0;
                if (bn_bwd) {
                    // data gradient + BatchNorm-backward partials (sum G, sum G * xhat), G = v where O > 0 (bpb_conv_s1, BpbS1BnBwd)
                    const float mu = cvals[NTC + cl], is = cvals[2 * NTC + cl];
                    float bs[16], bo[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        bs[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbs, (int)(offs[r] + nt * 128), 0, 0));
                    if (bn_out) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            bo[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbo, (int)(offs[r] + nt * 128), 0, 0));
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) bo[r] = offs[r] < OOB ? 1.f : 0.f;
                    }
                    float fs = 0.f, fq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[nt][r];
                        if (accum) v += old[r];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)(offs[r] + nt * 128), 0, 0);
                        const float gg = bo[r] > 0.f ? v : 0.f;
                        fs += gg;
                        fq += gg * ((bs[r] - mu) * is);
                    }
                    ssum = (double)fs;
                    ssq = (double)fq;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[nt][r] + bias_v;
                        if (accum) v += old[r];
                        if (relu) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, (int)(offs[r] + nt * 128), 0, 0);
                        if (do_stats) {
                            const double dv = offs[r] < OOB ? (double)v : 0.0;
                            ssum += dv;
                            ssq += dv * dv;
                        }
                    }
                }
                if (do_stats) {      // this wave's running (fp64) sums of the channel, in its own scratch row: no other wave touches it
                    const double s = ssum + __shfl_xor(ssum, 32);
                    const double q = ssq + __shfl_xor(ssq, 32);
                    if (half == 0) {
                        double* rr = red + ((wave * NTC + pass * 64 + nt * 32 + l31) << 1);
                        rr[0] += s;
                        rr[1] += q;
                    }
                }
            }
        }
    };

    // ---- the item loop
    int c = 0;
    while (t < ntiles) {
        int tn = t, cn = c + 1;
        if (cn == KC) { cn = 0; tn = t + tstride; }
        const unsigned rn = cn == 0 ? rowoff(tn) : roff;
        const bool have_next = tn < ntiles;
        item(t, c, have_next, rn, cn);
        t = tn; c = cn; roff = rn;
    }
#undef PW_LOAD

    if (do_stats) {   // one partial row per workgroup: the four waves' sums in a fixed order (deterministic, no atomics)
        __syncthreads();
        if ((int)threadIdx.x < NTC) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += red[((w * NTC + (int)threadIdx.x) << 1) + 0];
                q += red[((w * NTC + (int)threadIdx.x) << 1) + 1];
            }
            const int co = ncol * NTC + (int)threadIdx.x;
            double BPB_GLOBAL* gs = (double BPB_GLOBAL*)P.stats;
            gs[((size_t)g * 2 + 0) * Cout + co] = s;
            gs[((size_t)g * 2 + 1) * Cout + co] = q;
        }
    }
}

// ------------------------------------ C ABI ------------------------------------------
static int conv_pw_lds_bytes(const BpbConvPwProb& p) { return p.Cin * p.NTC * 4 + 4 * p.NTC * 16 + 3 * p.NTC * 4; }

extern "C" {

int bpb_conv_pw_init(void)
{
    hipError_t e = hipFuncSetAttribute((const void*)bpb_conv_pw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bpb_conv_pw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e == hipSuccess ? 0 : bpb_set_error((int)e, "bpb_conv_pw_init: %s", hipGetErrorString(e));
}

// Grouped launch of pointwise problems of one variant (all Cin == 64, or all Cin in {128, 192, 256} with NTC == 64).
int bpb_conv_pw(const BpbConvPwProb* d_probs, const BpbConvPwProb* h_probs, int nprobs, hipStream_t stream)
{
    BPB_REQUIRE(nprobs >= 1 && nprobs <= 16, "bpb_conv_pw: nprobs=%d out of range", nprobs);
    const bool kc1 = h_probs[0].Cin == 64;
    int nblk = 0, lds = 0;
    for (int i = 0; i < nprobs; ++i) {
        const BpbConvPwProb& p = h_probs[i];
        BPB_REQUIRE((p.Cin == 64) == kc1, "bpb_conv_pw: mixed kernel variants in one group");
        BPB_REQUIRE(p.Cin % 64 == 0 && p.Cin >= 64 && p.Cin <= 256 && p.Cout % 64 == 0, "bpb_conv_pw: Cin=%d (64..256, multiple of 64), Cout=%d (multiple of 64)",
                    p.Cin, p.Cout);
        BPB_REQUIRE((p.NTC == 64 || p.NTC == 128 || p.NTC == 256) && (kc1 || p.NTC == 64) && p.Cout == p.NTC << p.l_ntiles && p.l_ntiles >= 0 &&
                        p.l_ntiles <= 4,
                    "bpb_conv_pw: column split NTC=%d x 2^%d does not cover Cout=%d (Cin > 64 needs NTC = 64)", p.NTC, p.l_ntiles, p.Cout);
        BPB_REQUIRE(p.P >= 1 && p.ntiles32 == bpb_cdiv(p.P, 32) && p.n_mtiles >= 1 && p.n_mtiles * 4 <= p.ntiles32 + 3,
                    "bpb_conv_pw: pixel tiling mismatch (P=%d, tiles=%d, groups=%d)", p.P, p.ntiles32, p.n_mtiles);
        BPB_REQUIRE(p.x_bytes > 0 && p.w_bytes > 0 && p.y_bytes > 0 && p.x_bytes < 0x80000000u && p.w_bytes < 0x80000000u && p.y_bytes < 0x80000000u &&
                        (double)p.P * p.Cin * 4 <= (double)p.x_bytes && (double)p.P * p.Cout * 4 <= (double)p.y_bytes &&
                        (double)p.Cin * p.Cout * 4 <= (double)p.w_bytes,
                    "bpb_conv_pw: tensors addressed through a buffer descriptor must be < 2 GiB and hold P x C elements");
        BPB_REQUIRE(p.P < 16777216 && p.Cin * 4 < 16777216 && p.Cout * 4 < 16777216, "bpb_conv_pw: 24-bit index arithmetic overflow");
        BPB_REQUIRE(((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0, "bpb_conv_pw: x/w must be 16-byte aligned");
        BPB_REQUIRE(p.res == nullptr || p.accumulate == 0, "bpb_conv_pw: a residual operand excludes the accumulate mode");
        BPB_REQUIRE(p.bnb == nullptr || (p.stats != nullptr && p.relu == 0 && p.res == nullptr && p.bias == nullptr),
                    "bpb_conv_pw: the BatchNorm-backward partials need `stats` and a plain (or accumulating) data-gradient epilogue");
        BPB_REQUIRE(p.blk_begin == nblk, "bpb_conv_pw: blk_begin mismatch");
        nblk += p.n_mtiles << p.l_ntiles;
        const int l = conv_pw_lds_bytes(p);
        lds = l > lds ? l : lds;
    }
    BPB_REQUIRE(lds <= 160 * 1024, "bpb_conv_pw: needs %d B of LDS", lds);
    const BpbBlkBegins bb = bpb_blk_begins(h_probs, nprobs);
    if (kc1) hipLaunchKernelGGL((bpb_conv_pw_kernel<true>), dim3(nblk), dim3(256), lds, stream, d_probs, bb);
    else hipLaunchKernelGGL((bpb_conv_pw_kernel<false>), dim3(nblk), dim3(256), lds, stream, d_probs, bb);
    BPB_LAUNCH_OK();
    return 0;
}

}   // extern "C"
