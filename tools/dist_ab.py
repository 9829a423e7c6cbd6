"""A/B of the eval part-distance kernel alone (BASELINE configs[4] size): ms with and without the per-part matrix.  Select a library
variant with BPB_LIB_PATH (measurement builds under bpbreid_amd/variants/, never the product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
from bpbreid_amd.metrics import compute_distance_matrix_using_bp_features
dev = torch.device('cuda', 0)
qf, gf, qv, gv, _ = bench.eval_inputs(dev)
Q, G, P, D = bench.EVAL_SHAPE
def timed(parts, reps=10):
    fn = lambda: compute_distance_matrix_using_bp_features(qf, gf, qv, gv, 'mean', 500, True, 'euclidean', return_device_tensors=True, want_parts=parts)
    for _ in range(3): r = fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): r = fn()
    e.record(); torch.cuda.synchronize()
    return r, s.elapsed_time(e) / reps
(dm, pm), ms_p = timed(True)
(dm2, _), ms_n = timed(False)
fl = 2.0 * P * Q * G * D
print('%s: with parts %.3f ms (%.3f of peak), without %.3f ms (%.3f); checksum %.6f %.6f equal=%s' % (
    os.path.basename(os.environ.get('BPB_LIB_PATH', 'product')), ms_p, fl / ms_p * 1e-9 / 157.3, ms_n, fl / ms_n * 1e-9 / 157.3,
    float(dm.double().sum()), float(pm.double().sum()), bool(torch.equal(dm, dm2))))
