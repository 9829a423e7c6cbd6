"""GPU: eval-mode forward of the bench batch, eager launches vs one hipGraph replay (what would a captured feature-extraction step buy?).
    python tools/eval_graph.py [backbone]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import common as Cm
from bpbreid_amd.model import bpbreid

backbone = sys.argv[1] if len(sys.argv) > 1 else 'hrnet32'
dev = torch.device('cuda', 0)
cfg = Cm.make_cfg(backbone, 5, 512)
model = Cm.fill_state_dict_(bpbreid(751, config=cfg, pretrained=False)).to(dev)
imgs, masks, pids = Cm.synth_batch(64, 256, 128, 5, 751, seed=1234)
imgs, masks = imgs.to(dev), masks.to(dev)
model.materialize_spatial_features = False      # what ImagePartBasedEngine sets (feature extraction never reads the map)
model.eval()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(reps):
        fn()
    host = (time.perf_counter() - t0) / reps * 1e3
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, host


with torch.no_grad(), model.eval_weights_cached():
    eager = lambda: model(imgs, external_parts_masks=masks)
    ms, host = timed(eager)
    print('%s eval forward, eager: %.3f ms per batch (host enqueue %.3f ms)' % (backbone, ms, host))
    ref = eager()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eager()
    ms, host = timed(g.replay)
    print('%s eval forward, hipGraph replay: %.3f ms per batch (host %.3f ms)' % (backbone, ms, host))

    def flat(o):
        if isinstance(o, torch.Tensor):
            return [o]
        if isinstance(o, dict):
            return [t for v in o.values() for t in flat(v)]
        if isinstance(o, (list, tuple)):
            return [t for v in o for t in flat(v)]
        return []
    same = all(torch.equal(a, b) for a, b in zip(flat(ref), flat(out)))
    print('outputs bit-identical to the eager forward:', same)
