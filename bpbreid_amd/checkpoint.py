"""Checkpoint compatibility with the reference.

Same call surface as torchreid/utils/torchtools.py:24-62 (save_checkpoint), :65-98 (load_checkpoint), :101-137
(resume_from_checkpoint), :260-315 (load_pretrained_weights) and torchreid/models/hrnet.py:588-600 (HRNet ImageNet weights):
the authors' ``.pth.tar`` files load into the MI355X model unchanged (its state-dict keys are the reference's), and files
written here load back into the reference.

Two things the reference gets for free from its environment and that are handled explicitly here:
  * the authors' checkpoints embed their yacs ``CfgNode`` (engine.py:95); unpickling that needs `yacs`.  Classes whose
    module is not installed are materialised as plain attribute-dicts instead of failing the whole load.
  * their optimizer state is torch.optim.Adam's (indexed by parameter position).  `FusedAdam` keeps flat moment arenas;
    it exports / imports that format, matching positions through the parameter *names* stored in the checkpoint.
"""
import io
import os
import os.path as osp
import pickle
import shutil
import warnings
from collections import OrderedDict

import torch

BUFFER_LEAVES = ('running_mean', 'running_var', 'num_batches_tracked')


class ForeignObject(dict):
    """Stand-in for an instance of a class that cannot be imported here (e.g. yacs.config.CfgNode)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)
            self.__dict__.update({k: v for k, v in state.items() if isinstance(k, str) and k.startswith('_')})

    def __reduce_ex__(self, protocol):
        return (dict, (dict(self),))


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (ForeignObject,), {'__module__': module})


class _tolerant_pickle:
    """A `pickle_module` for torch.load."""
    __name__ = 'pickle'
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _TolerantUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler
    PickleError, UnpicklingError, PicklingError = pickle.PickleError, pickle.UnpicklingError, pickle.PicklingError
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


def strip_module_prefix(state_dict):
    """nn.DataParallel / DistributedDataParallel prefix their keys with 'module.' (torchtools.py:46-53, :287-288)."""
    return OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state_dict.items())


def save_checkpoint(state, save_dir, job_id=None, is_best=False, remove_module_from_keys=False):
    """Write ``job-<job_id>_<epoch>_model.pth.tar`` (and ``model-best.pth.tar`` when `is_best`) like the reference."""
    os.makedirs(save_dir, exist_ok=True)
    if remove_module_from_keys:
        state['state_dict'] = strip_module_prefix(state['state_dict'])
    fpath = osp.join(save_dir, 'job-{}_{}_model.pth.tar'.format(job_id, str(state['epoch'])))
    torch.save(state, fpath)
    if is_best:
        shutil.copy(fpath, osp.join(osp.dirname(fpath), 'model-best.pth.tar'))
    return fpath


def load_checkpoint(fpath, map_location='cpu'):
    """Read a checkpoint written by the reference or by `save_checkpoint` (python-2 era pickles included)."""
    if fpath is None:
        raise ValueError('File path is None')
    if not osp.exists(fpath):
        raise FileNotFoundError('File is not found at "{}"'.format(fpath))
    try:
        return torch.load(fpath, map_location=map_location, pickle_module=_tolerant_pickle, weights_only=False)
    except UnicodeDecodeError:
        class _Latin1(_tolerant_pickle):
            Unpickler = staticmethod(lambda f, **kw: _TolerantUnpickler(f, encoding='latin1'))
            load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, encoding='latin1').load())
        return torch.load(fpath, map_location=map_location, pickle_module=_Latin1, weights_only=False)


def parameter_names(state_dict):
    """Names of the parameters (not buffers) of a checkpoint's state dict, in registration order: position i of the
    reference's torch.optim state is the i-th of these (the optimizer is built from model.parameters(), optimizer.py:113)."""
    return [k for k in strip_module_prefix(state_dict) if k.rsplit('.', 1)[-1] not in BUFFER_LEAVES]


def load_pretrained_weights(model, weight_path, verbose=False):
    """Copy every entry whose name and shape match; ignore the rest (e.g. identity classifiers trained on another
    dataset).  Returns (matched, discarded) key lists."""
    checkpoint = load_checkpoint(weight_path)
    state_dict = checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint
    own = model.state_dict()
    matched, discarded = [], []
    with torch.no_grad():
        for k, v in strip_module_prefix(state_dict).items():
            if k in own and tuple(own[k].shape) == tuple(v.shape):
                own[k].copy_(v)                 # in place: parameters are views of the flat arena
                matched.append(k)
            else:
                discarded.append(k)
    if not matched:
        warnings.warn('The pretrained weights "{}" cannot be loaded, please check the key names manually '
                      '(** ignored and continue **)'.format(weight_path))
    elif verbose and discarded:
        print('** discarded (unmatched name or size): {}'.format(discarded))
    return matched, discarded


def load_hrnet_imagenet_weights(backbone, pretrained_path):
    """HRNet-W32-C ImageNet weights into the HRNet trunk: every key that exists in the trunk is taken, the classification
    head of the ImageNet model is dropped (hrnet.py:588-600)."""
    if not osp.exists(pretrained_path):
        raise FileNotFoundError('HRNet pretrained weights not found under "{}"'.format(pretrained_path))
    src = load_checkpoint(pretrained_path)
    own = backbone.state_dict()
    taken = []
    with torch.no_grad():
        for k, v in src.items():
            if k in own:
                own[k].copy_(v)
                taken.append(k)
    return taken


def resume_from_checkpoint(fpath, model, optimizer=None, scheduler=None):
    """Model weights (strict), optimizer and scheduler state; returns the epoch to start from."""
    checkpoint = load_checkpoint(fpath)
    sd = strip_module_prefix(checkpoint['state_dict'])
    model.load_state_dict(sd)
    if optimizer is not None and 'optimizer' in checkpoint:
        from .optim import FusedAdam
        if isinstance(optimizer, FusedAdam):
            optimizer.load_state_dict(checkpoint['optimizer'], param_names=parameter_names(sd))
        else:
            optimizer.load_state_dict(checkpoint['optimizer'])
    if scheduler is not None and 'scheduler' in checkpoint:
        scheduler.load_state_dict(checkpoint['scheduler'])
    return checkpoint['epoch']
