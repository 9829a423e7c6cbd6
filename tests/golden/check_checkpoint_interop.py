"""Dev-container-only check (needs /root/reference, like gen_golden.py): checkpoints cross between the real reference and
bpbreid_amd.checkpoint in both directions.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/check_checkpoint_interop.py
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(os.path.dirname(HERE))]
import torch                                                   # noqa: E402
import common as Cm                                            # noqa: E402
from _ref_loader import load_reference                        # noqa: E402

load_reference()
import torchreid                                               # noqa: E402
from torchreid.utils import torchtools as RT                   # noqa: E402
from bpbreid_amd import checkpoint as ck                       # noqa: E402
from bpbreid_amd.model import bpbreid                          # noqa: E402

cfg = Cm.make_cfg('resnet50', 3, 64)
ref = Cm.fill_state_dict_(torchreid.models.build_model('bpbreid', num_classes=12, config=cfg, pretrained=False), 7)
mine = Cm.fill_state_dict_(bpbreid(12, config=cfg, pretrained=False), 8)
tmp = tempfile.mkdtemp()
# reference -> here (file written by the reference's save_checkpoint, optimizer state of torch.optim.Adam)
opt = torch.optim.Adam(ref.parameters(), lr=3.5e-4)
RT.save_checkpoint({'state_dict': ref.state_dict(), 'epoch': 5, 'rank1': 0.3, 'optimizer': opt.state_dict()}, tmp, job_id=1)
f = os.path.join(tmp, 'job-1_5_model.pth.tar')
matched, discarded = ck.load_pretrained_weights(mine, f)
assert not discarded, discarded[:5]
assert all(torch.equal(a, mine.state_dict()[k]) for k, a in ref.state_dict().items())
assert ck.parameter_names(ref.state_dict()) == [n for n, _ in ref.named_parameters()]
print('reference -> bpbreid_amd: %d entries, 0 discarded' % len(matched))
# here -> reference
mine2 = Cm.fill_state_dict_(bpbreid(12, config=cfg, pretrained=False), 9)
g = ck.save_checkpoint({'state_dict': mine2.state_dict(), 'epoch': 9}, tmp, job_id=2)
ref2 = torchreid.models.build_model('bpbreid', num_classes=12, config=cfg, pretrained=False)
RT.load_pretrained_weights(ref2, g)
assert all(torch.equal(a, ref2.state_dict()[k]) for k, a in mine2.state_dict().items())
assert RT.resume_from_checkpoint(g, ref2) == 9          # strict load_state_dict: the key sets are identical
print('bpbreid_amd -> reference: strict load OK (%d keys)' % len(mine2.state_dict()))
