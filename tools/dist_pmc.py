"""GPU probe: the eval-time part-based distance kernel alone at BASELINE config-5 size (Q=2048, G=20000, P=9, D=512), timed
with events; run under `rocprofv3 --pmc ...` for the SQ counters of exactly these launches.
    python tools/dist_pmc.py [Q G P D] [reps] [want_parts 0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import torch.nn.functional as F
from bpbreid_amd.metrics import part_distance_raw

Q, G, P, D = [int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (2048, 20000, 9, 512))]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
want = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(4321)
qf = F.normalize(torch.randn(Q, P, D, generator=g), dim=-1).to(dev)
gf = F.normalize(torch.randn(G, P, D, generator=g), dim=-1).to(dev)
qv = (torch.rand(Q, P, generator=g) < 0.8).to(dev)
gv = (torch.rand(G, P, generator=g) < 0.8).to(dev)
for _ in range(2):
    part_distance_raw(qf, gf, qv, gv, 'mean', 'euclidean', finalize=True, want_parts=want)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    part_distance_raw(qf, gf, qv, gv, 'mean', 'euclidean', finalize=True, want_parts=want)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print('part distance Q=%d G=%d P=%d D=%d parts=%d: %.3f ms  %.1f TF (%.3f of the fp32 MFMA peak)'
      % (Q, G, P, D, want, ms, 2.0 * P * Q * G * D / ms * 1e-9, 2.0 * P * Q * G * D / ms * 1e-9 / 157.3), flush=True)
