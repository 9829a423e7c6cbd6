"""Dev-container-only loader for the real reference (`/root/reference`).

TEST INFRASTRUCTURE, never shipped, never imported on the GPU box.  The reference eagerly
imports packages that are not installed here (yacs, albumentations, cv2, monai,
torchmetrics, wandb, deepdiff, torchvision, ...).  This loader registers stand-in modules for
those names *before* `import torchreid`, plus two functional shims that sit on the hot path:

* ``yacs.config.CfgNode``  -- attribute/item dict, used by hrnet.py:26-56 and default_config.py
* ``torchmetrics.Accuracy`` -- top-1 accuracy, reporting only (GiLt_loss.py:118)

Nothing from the reference is copied; the reference is imported with bytecode writing
disabled so the read-only tree is not touched (SURVEY.md handling note).
"""
import importlib.abc
import importlib.machinery
import sys
import types

sys.dont_write_bytecode = True
REFERENCE_ROOT = '/root/reference'

_STUB_TOPLEVEL = (
    'albumentations', 'cv2', 'monai', 'wandb', 'deepdiff', 'torchvision', 'skimage', 'h5py',
    'gdown', 'optuna', 'clearml', 'openpifpaf', 'detectron2', 'tensorboard', 'tb_nightly',
    'torchmetrics', 'yacs',
)


class _Anything:
    """Placeholder class/object: callable, subclassable, attribute-transparent."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = self.__name__ + '.' + name
        if full in sys.modules:
            return sys.modules[full]
        return type(name, (_Anything,), {})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _STUB_TOPLEVEL and fullname not in _REAL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_REAL = set()


class CfgNode(dict):
    """Minimal yacs.config.CfgNode: nested dict with attribute access."""

    def __init__(self, init_dict=None, new_allowed=False, **kw):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass


class Accuracy:
    """torchmetrics.Accuracy(top_k=1) stand-in: mean(argmax == target)."""

    def __init__(self, top_k=1, **kw):
        pass

    def cuda(self):
        return self

    def __call__(self, preds, target):
        return (preds.argmax(dim=1) == target).float().mean()


def load_reference():
    """Import and return the reference `torchreid` package (dev container only)."""
    if 'torchreid' in sys.modules:
        return sys.modules['torchreid']
    sys.meta_path.insert(0, _StubFinder())
    import importlib
    # torch.utils.tensorboard refuses to import without tensorboard: pre-register a stand-in
    tb = _StubModule('torch.utils.tensorboard')
    tb.__path__ = []
    sys.modules['torch.utils.tensorboard'] = tb
    yacs_cfg = importlib.import_module('yacs.config')
    yacs_cfg.CfgNode = CfgNode
    tm = importlib.import_module('torchmetrics')
    tm.Accuracy = Accuracy
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import torchreid  # noqa
    return torchreid


def default_cfg():
    """The reference's default config tree (scripts/default_config.py:11) with logging off."""
    load_reference()
    from torchreid.scripts.default_config import get_default_config
    cfg = get_default_config()
    cfg.project.logger.save_disk = False
    cfg.model.pretrained = False
    return cfg
