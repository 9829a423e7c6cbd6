"""Checkpoint compatibility, CPU part: file format, tolerant unpickling of the embedded yacs config,
name/shape-matched weight loading, resume with a torch optimizer, scheduler state.  (FusedAdam <-> torch.optim.Adam state
interchange needs the GPU arenas: tests/test_gpu_model.py.)"""
import os
import sys
import types

import pytest
import torch

import common as Cm
from bpbreid_amd import checkpoint as ck
from bpbreid_amd.model import bpbreid
from bpbreid_amd.optim import WarmupMultiStepLR


def small_model(ncls=8, seed=0):
    return Cm.fill_state_dict_(bpbreid(ncls, config=Cm.make_cfg('hrnet_w8', 3, 32), pretrained=False), seed)


def test_foreign_config_object_does_not_break_loading(tmp_path):
    # the reference stores its yacs CfgNode in every checkpoint (engine.py:95); yacs is not installed here
    mod = types.ModuleType('yacs_like_missing_pkg.config')
    parent = types.ModuleType('yacs_like_missing_pkg')
    cls = type('CfgNode', (dict,), {'__module__': 'yacs_like_missing_pkg.config'})
    mod.CfgNode = cls
    sys.modules['yacs_like_missing_pkg'], sys.modules['yacs_like_missing_pkg.config'] = parent, mod
    try:
        cfg = cls(model=cls(name='bpbreid', bpbreid=cls(backbone='hrnet32')), train=cls(lr=3.5e-4))
        path = str(tmp_path / 'c.pth.tar')
        torch.save({'state_dict': {'w': torch.arange(4.)}, 'epoch': 7, 'rank1': 0.5, 'config': cfg}, path)
    finally:
        del sys.modules['yacs_like_missing_pkg'], sys.modules['yacs_like_missing_pkg.config']
    with pytest.raises(Exception):
        torch.load(path, weights_only=False)                    # what the reference's loader would do without yacs
    c = ck.load_checkpoint(path)
    assert c['epoch'] == 7 and torch.equal(c['state_dict']['w'], torch.arange(4.))
    assert c['config']['model']['bpbreid']['backbone'] == 'hrnet32' and c['config'].train.lr == 3.5e-4
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(str(tmp_path / 'missing'))
    with pytest.raises(ValueError):
        ck.load_checkpoint(None)


def test_save_checkpoint_layout_and_module_prefix(tmp_path):
    m = small_model()
    sd = {'module.' + k: v for k, v in m.state_dict().items()}
    f = ck.save_checkpoint({'state_dict': sd, 'epoch': 3, 'rank1': 0.1}, str(tmp_path / 'run'), job_id=42, is_best=True,
                           remove_module_from_keys=True)
    assert os.path.basename(f) == 'job-42_3_model.pth.tar'                 # torchtools.py:56
    assert os.path.exists(str(tmp_path / 'run' / 'model-best.pth.tar'))
    c = ck.load_checkpoint(f)
    assert list(c['state_dict']) == list(m.state_dict())


def test_load_pretrained_weights_matches_by_name_and_shape(tmp_path):
    src, dst = small_model(ncls=8, seed=1), small_model(ncls=5, seed=2)   # other dataset -> identity classifiers differ
    path = str(tmp_path / 'w.pth.tar')
    torch.save({'state_dict': {'module.' + k: v for k, v in src.state_dict().items()}, 'epoch': 1}, path)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    ptr = dst.pixel_classifier.classifier.weight.data_ptr()
    matched, discarded = ck.load_pretrained_weights(dst, path)
    assert dst.pixel_classifier.classifier.weight.data_ptr() == ptr        # in place (arena views stay valid)
    after, ref = dst.state_dict(), src.state_dict()
    cls_w = [k for k in ref if k.endswith('identity_classifier.classifier.weight') or 'parts_identity_classifier' in k and k.endswith('classifier.weight')]
    assert cls_w and all(k in discarded for k in cls_w)
    for k in matched:
        assert torch.equal(after[k], ref[k])
    for k in discarded:
        assert torch.equal(after[k], before[k])
    assert len(matched) + len(discarded) == len(ref) and len(matched) > 300
    # a bare state dict (no 'state_dict' wrapper) works too (torchtools.py:276-279)
    torch.save(src.state_dict(), path)
    assert len(ck.load_pretrained_weights(small_model(ncls=8, seed=3), path)[1]) == 0


def test_hrnet_imagenet_weights_into_trunk(tmp_path):
    m = small_model()
    trunk = m.backbone_appearance_feature_extractor
    donor = {k: torch.full_like(v, 0.25) for k, v in trunk.state_dict().items() if v.dtype == torch.float32}
    donor['classifier.weight'] = torch.zeros(1000, 2048)                    # ImageNet head: not in the trunk -> dropped
    path = str(tmp_path / 'hrnet.pth')
    torch.save(donor, path)
    taken = ck.load_hrnet_imagenet_weights(trunk, path)
    assert 'classifier.weight' not in taken and len(taken) == len(donor) - 1
    assert float(trunk.conv1.weight.mean()) == 0.25
    with pytest.raises(FileNotFoundError):
        ck.load_hrnet_imagenet_weights(trunk, str(tmp_path / 'nope'))


def test_resume_with_torch_optimizer_and_scheduler(tmp_path):
    m = small_model(seed=4)
    params = [p for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=3.5e-4, weight_decay=5e-4)
    sched = WarmupMultiStepLR(opt)
    for p in params[:5]:
        p.grad = torch.ones_like(p)
    opt.step()
    for _ in range(12):
        sched.step()
    path = ck.save_checkpoint({'state_dict': m.state_dict(), 'epoch': 12, 'optimizer': opt.state_dict(),
                               'scheduler': sched.state_dict()}, str(tmp_path), job_id=0)
    m2 = small_model(seed=5)
    opt2 = torch.optim.Adam(list(m2.parameters()), lr=1.0)
    sched2 = WarmupMultiStepLR(opt2)
    assert ck.resume_from_checkpoint(path, m2, opt2, sched2) == 12
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    assert sched2.last_epoch == 12 and opt2.param_groups[0]['lr'] == pytest.approx(3.5e-4)
    assert torch.equal(opt2.state[list(m2.parameters())[0]]['exp_avg'], opt.state[params[0]]['exp_avg'])
    names = ck.parameter_names(m.state_dict())
    assert names == [n for n, _ in m.named_parameters()]
