"""GPU: what the fp32 matrix pipe sustains on this board -- a register-only v_mfma_f32_32x32x2_f32 loop on every SIMD (no memory,
no LDS), short and long launches, with the shader clock read back through s_memtime against the 100 MHz wall clock.  The
roofline fractions in DESIGN.md are quoted against the nominal 157.3 TFLOP/s (2.4 GHz); this tool says how much of that the
power-managed clock leaves under a pure MFMA load.

    python tools/mfma_peak.py
"""
import os, subprocess, sys, tempfile, ctypes as C
import torch

SRC = r'''
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __global__ __launch_bounds__(256) void mfma_loop(int iters, unsigned long long* out, float* sink)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = (float)threadIdx.x * 1e-9f, y = 1.0f + x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
        out[w * 2] = t1 - t0;
        out[w * 2 + 1] = w1 - w0;
    }
}
extern "C" int launch(int blocks, int iters, void* out, void* sink, void* stream)
{
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)out, (float*)sink);
    return (int)hipGetLastError();
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, 'k.hip'), 'w').write(SRC)
so = os.path.join(d, 'k.so')
subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, os.path.join(d, 'k.hip')])
lib = C.CDLL(so)
lib.launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device('cuda', 0)
sink = torch.zeros(4, device=dev)
st = torch.cuda.current_stream().cuda_stream
print('%8s %8s %10s %10s %12s %14s' % ('blocks', 'iters', 'ms', 'TFLOP/s', 'shader MHz', 'cycles/MFMA'))
for blocks in (256, 512, 1024):
    for iters in (200, 2000, 20000, 100000):
        out = torch.zeros(blocks * 4 * 2, device=dev, dtype=torch.int64)
        for _ in range(2):
            lib.launch(blocks, iters, out.data_ptr(), sink.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            lib.launch(blocks, iters, out.data_ptr(), sink.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        o = out.view(-1, 2).double()
        cyc, wall = o[:, 0].median().item(), o[:, 1].median().item()
        flops = blocks * 4.0 * iters * 16 * 4096
        print('%8d %8d %10.3f %10.1f %12.0f %14.2f' % (blocks, iters, ms, flops / ms / 1e9, cyc / (wall / 100.0), cyc / (iters * 16.0)), flush=True)
