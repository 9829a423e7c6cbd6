"""GPU: the experimental mixed-tile 3x3 kernel (csrc/experimental/conv_s1_mixed.hip: the shallow wide branches of a grouped launch
take two pixel sub-tiles per wave, the deep ones one) against the product kernel.  Builds a library of its own
(bpbreid_amd/build/libbpbreid_hip_mixed.so = the product objects with conv_s1.o replaced) and runs, in child processes with
BPB_LIB_PATH / BPB_S1_MIXED set:

    python tools/s1_mixed.py test      kernel- and model-level parity tests on the experimental library
    python tools/s1_mixed.py trace     launch timelines x4 / x3 / x2, product vs experimental (tools/s1_trace.py)
    python tools/s1_mixed.py bench     forward-only and train-step A/B (tools/fwd_bench.py, bench.py)
    python tools/s1_mixed.py           all three
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from bpbreid_amd import build as B

objdir = os.path.join(os.path.dirname(B.LIB), 'build')
src = os.path.join(B.CSRC, 'experimental', 'conv_s1_mixed.hip')
obj = os.path.join(objdir, 'conv_s1_mixed.o')
lib = os.path.join(objdir, 'libbpbreid_hip_mixed.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def build():
    B.build()
    if not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(src):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', src, '-o', obj,
                               '-I', B.CSRC, '-I', os.path.join(ROOT, 'include'), '-Wno-unused-value'])
    objs = [os.path.join(objdir, s.rsplit('.', 1)[0] + '.o') for s in B.SOURCES if s != 'conv_s1.hip'] + [obj]
    if not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib


def run(cmd, mixed, **env):
    e = dict(os.environ, **env)
    if mixed:
        e.update(BPB_LIB_PATH=lib, BPB_S1_MIXED='1')
    print('+ [%s] %s' % ('experimental' if mixed else 'product', ' '.join(cmd)), flush=True)
    return subprocess.call(cmd, cwd=ROOT, env=e)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    build()
    rc = 0
    if what in ('test', 'all'):
        rc |= run([sys.executable, '-m', 'pytest', 'tests/test_gpu_kernels.py', '-x', '-q', '-m', 'gpu', '-k',
                   'conv or multires or basic_block or bottleneck or full_backbones'], True)
        rc |= run([sys.executable, '-m', 'pytest', 'tests/test_gpu_model.py', '-x', '-q', '-m', 'gpu', '-k', 'golden or full_size or odd_shapes'], True)
    if what in ('trace', 'all'):
        for launch in ('x4', 'x3', 'x2'):
            for mixed in (False, True):
                run([sys.executable, 'tools/s1_trace.py', launch], mixed, **({'S1_TRACE_SRC': src} if mixed else {}))
    if what in ('bench', 'all'):
        for mixed in (False, True, False, True):
            run([sys.executable, 'tools/fwd_bench.py', 'hrnet32'], mixed)
            run([sys.executable, 'bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-roofline'], mixed)
    sys.exit(rc)
