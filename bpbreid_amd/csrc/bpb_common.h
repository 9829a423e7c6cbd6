// Shared definitions for the bpbreid_amd HIP library (gfx950 / MI355X only).
//
// Conventions (SURVEY.md section 8b "C-ABI"):
//  * every entry point is extern "C", takes plain pointers / sizes / a hipStream_t, allocates
//    nothing persistent, keeps no global state, enqueues on the passed stream only;
//  * return 0 on success, negative = argument error (checked before launch), positive =
//    hipError_t from the launch; bpb_last_error() gives a thread-local message.
//  * activations are fp32 NHWC ("channels-last") in HBM: channel is the unit-stride dim so a
//    wave's 64 lanes read/write contiguous channel runs (coalesced 128-B+ segments) and the
//    MFMA A/B fragments are 16-byte vector loads.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/bpbreid_hip.h"
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>


int bpb_set_error(int code, const char* fmt, ...);

#define BPB_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) return bpb_set_error(-1, __VA_ARGS__); \
    } while (0)

#define BPB_LAUNCH_OK()                                                          \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) return bpb_set_error((int)e__, "%s: %s", __func__, hipGetErrorString(e__)); \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
// pointers that come out of descriptor structs are generic ("flat") to the compiler; these casts make the
// accesses global_load/global_store (vmcnt only) instead of flat_* (vmcnt + lgkmcnt)
#define BPB_GLOBAL __attribute__((address_space(1)))
typedef const float BPB_GLOBAL* bpb_gcf;
typedef float BPB_GLOBAL* bpb_gf;
#define BPB_GLD4(p) (*(const f32x4 BPB_GLOBAL*)(p))
#define BPB_GST4(p) (*(f32x4 BPB_GLOBAL*)(p))
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int bpb_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Grouped launches: block -> problem.  The first-block prefix of the (<= 16) problems travels BY VALUE in the kernel-argument
// segment: a workgroup finds its problem with scalar compares and then loads ONE descriptor, instead of a chain of dependent
// loads of descs[i].blk_begin from device memory in front of the descriptor load (two memory round trips at the start of every
// workgroup; the element-wise kernels live ~10 us).
struct BpbBlkBegins {
    int begin[16];
};
template <typename D>
static inline BpbBlkBegins bpb_blk_begins(const D* h_descs, int n)
{
    BpbBlkBegins b;
    for (int i = 0; i < 16; ++i) b.begin[i] = i < n ? h_descs[i].blk_begin : 0x7fffffff;
    return b;
}
#ifdef __HIPCC__
__device__ __forceinline__ int bpb_find_problem(const BpbBlkBegins& bb, int bid)
{
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i)
        if (bid >= bb.begin[i]) pi = i;
    return pi;
}
#endif

// ---------------------------------------------------------------------------------------
// Descriptor of one implicit-GEMM convolution problem.  Lives in DEVICE memory; a launch takes an
// array of them ("grouped launch": independent branches of an HRNet module share one launch so
// that 256 CUs stay filled).
//
//   y[n, a*osh+ooh, b*osw+oow, :] (+)= sum_t  x[n, a*sa + dh_t + ih0, b*sa + dw_t + iw0, :] . W_t
//
// covers forward convs (any R,S,stride,pad), stride-1 dgrad and the parity classes of strided
// dgrad with one kernel.  W is pre-packed as [tap][Cin/4][Cout][4] (k-quad innermost) so the
// MFMA B fragment of four consecutive k-steps is one 16-byte load.
// ---------------------------------------------------------------------------------------
