"""GPU, measurement only (results are WRONG): the train step with the BatchNorm + ReLU apply launches (`fuse_fwd`) of plain
convolution -> BatchNorm -> ReLU edges (one term, no residual) dropped from the training plan -- an upper bound on what applying the
BatchNorm inside the consumer's operand load could buy (round-4 review item 5).
    python tools/ablate_apply_on_load.py [drop|none] [bench.py arguments ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
what = sys.argv[1] if len(sys.argv) > 1 else 'drop'
from bpbreid_amd import graph, native as nv

orig = graph.Net._freeze
dropped = [0, 0.0]


def plain_edge(r):
    if r.kind != nv.OP_FUSE_FWD_MULTI or r.desc is None:
        return False
    d = r.desc
    return d.nterms == 1 and d.relu == 1 and bool(d.scale[0]) and d.up[0] == 0


def freeze(self, recs, name=None):
    if what == 'drop' and name == 'train':
        keep = []
        for r in recs:
            if plain_edge(r):
                dropped[0] += 1
                dropped[1] += r.bytes
            else:
                keep.append(r)
        recs = keep
        print('ablate_apply_on_load: %d fuse_fwd records (%.2f GB) dropped from the training plan' % (dropped[0], dropped[1] / 1e9), file=sys.stderr)
    return orig(self, recs, name)


graph.Net._freeze = freeze
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-roofline', '--no-forward-only', '--no-eval', '--no-extra',
            '--graph', '0'] + sys.argv[2:]
import bench
bench.main()
