"""GPU, measurement only (results are WRONG): the train step with the BatchNorm finalize launches dropped from the plans -- an upper
bound on what folding them into their consumers could buy.
    python tools/ablate_finalize.py [fwd|bwd|both|none] [bench.py arguments ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
what = sys.argv[1] if len(sys.argv) > 1 else 'both'
from bpbreid_amd import graph, native as nv

drop = {'fwd': (nv.OP_BN_FINALIZE_MULTI,), 'bwd': (nv.OP_BN_BWD_FINALIZE_MULTI,), 'both': (nv.OP_BN_FINALIZE_MULTI, nv.OP_BN_BWD_FINALIZE_MULTI),
        'none': ()}[what]
orig = graph.Net._freeze


def freeze(self, recs, name=None):
    return orig(self, [r for r in recs if r.kind not in drop], name)


graph.Net._freeze = freeze
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-roofline', '--no-forward-only', '--no-eval', '--graph', '0'] + sys.argv[2:]
import bench
bench.main()
