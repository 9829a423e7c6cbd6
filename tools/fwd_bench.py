"""GPU: forward-only (eval mode) throughput of the model, the figure the north star's '>= 60 % MFMA peak on the HRNet-W32
forward' refers to.  python tools/fwd_bench.py [backbone] [parts] [H] [W] [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch                                                  # noqa: E402
import common as Cm                                           # noqa: E402
from bpbreid_amd.model import bpbreid                        # noqa: E402

backbone = sys.argv[1] if len(sys.argv) > 1 else 'hrnet32'
k, h, w, n = [int(a) for a in (sys.argv[2:6] if len(sys.argv) > 5 else (5, 256, 128, 64))]
dev = torch.device('cuda', 0)
model = Cm.fill_state_dict_(bpbreid(751, config=Cm.make_cfg(backbone, k, 512), pretrained=False)).to(dev)
# the engine's setting (engine.py): `spatial_features` is not consumed, the head runs on the branch outputs.  BPB_FWD_SPATIAL=1
# measures the forward that also returns the concatenated 1 GB map.
model.materialize_spatial_features = os.environ.get('BPB_FWD_SPATIAL', '0') == '1'
imgs, masks, _ = Cm.synth_batch(n, h, w, k, 751)
imgs, masks = imgs.to(dev), masks.to(dev)
res = {}
import contextlib                                             # noqa: E402
for mode in os.environ.get('BPB_FWD_MODES', 'eval,train').split(','):
    model.train(mode == 'train')
    # eval: like the engine's feature extraction, the parameter-derived launches (BatchNorm affines, weight packing) run once
    cached = model.eval_weights_cached() if (mode == 'eval' and os.environ.get('BPB_FWD_CACHE', '1') != '0') else contextlib.nullcontext()
    with torch.no_grad(), cached:
        for _ in range(3):
            model(imgs, external_parts_masks=masks)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        reps = 10
        for _ in range(reps):
            model(imgs, external_parts_masks=masks)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    plan = next(iter(model._plans.values()))
    flops = sum(m['flops'] for m in (plan.net.plan_eval if mode == 'eval' else plan.net.plan_train)[2]) * n / plan.N
    res[mode] = {'ms_per_batch': ms, 'images_per_s': n / ms * 1e3, 'conv_tflops': flops / ms * 1e-9,
                 'frac_of_f32_mfma_peak': flops / ms * 1e-9 / 157.3}
print(json.dumps({'backbone': backbone, 'batch': n, 'spatial_features_materialised': model.materialize_spatial_features, 'forward_only': res}))
