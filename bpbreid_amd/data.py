"""Input side of the hot path: part-mask preprocessing on the GPU and the P x K identity sampler.

`MaskPreprocessor` is the batched, on-device counterpart of the reference's per-sample CPU mask transforms
(torchreid/data/masks_transforms/mask_transform.py:20-85, chained in torchreid/data/transforms.py:133-158): raw
human-parsing confidence maps [N, C, H, W] -> soft part masks [N, K+1, H/scale, W/scale], the tensor the model takes as
`external_parts_masks`.  The grouping table is not baked in: build it from the reference's own transform object
(`from_reference_transform(CombinePifPafIntoFiveVerticalParts())`) or pass `parts_grouping` / `parts_map` dicts.

`RandomIdentitySampler` restates torchreid/data/sampler.py:11-76 (P identities x K instances per batch); with the same
`random` / `numpy.random` seeds it yields the same index sequence as the reference.
"""
import copy
import random
from collections import defaultdict

import numpy as np
import torch

from . import native as nv

BACKGROUND_STRATEGIES = {'sum': 0, 'threshold': 1, 'diff_from_max': 2}


class MaskPreprocessor:
    def __init__(self, height, width, mask_scale=4, parts_grouping=None, parts_map=None, combine_mode='max',
                 background_computation_strategy='threshold', softmax_weight=15, mask_filtering_threshold=0.5):
        """Defaults follow scripts/default_config.py:63-68 (threshold background, soft-max weight 15, threshold 0.5)."""
        if background_computation_strategy not in BACKGROUND_STRATEGIES:
            raise ValueError('Background mask combine strategy {} not supported'.format(background_computation_strategy))
        self.size = (int(height / mask_scale), int(width / mask_scale))            # mask_transform.py:48
        self.bg = BACKGROUND_STRATEGIES[background_computation_strategy]
        self.softmax_weight, self.threshold = float(softmax_weight), float(mask_filtering_threshold)
        self.combine_sum = 1 if combine_mode == 'sum' else 0
        self.groups = None
        if parts_grouping is not None:
            self.parts_names = list(parts_grouping.keys())
            self.groups = [[parts_map[name] for name in parts_grouping[part]] for part in self.parts_names]
        self._tables = {}

    @classmethod
    def from_reference_transform(cls, transform, height, width, mask_scale=4, **kw):
        """`transform`: an instance of one of the reference's MaskGroupingTransform subclasses (pifpaf_mask_transform.py)."""
        return cls(height, width, mask_scale, parts_grouping=transform.parts_grouping, parts_map=transform.parts_map,
                   combine_mode=getattr(transform, 'combine_mode', 'max'), **kw)

    @property
    def parts_num(self):
        return len(self.groups) if self.groups is not None else None

    def _group_tables(self, device):
        t = self._tables.get(device)
        if t is None and self.groups is not None:
            offs = np.cumsum([0] + [len(g) for g in self.groups]).astype(np.int32)
            chans = np.concatenate([np.asarray(g, dtype=np.int32) for g in self.groups])
            t = (torch.from_numpy(offs).to(device), torch.from_numpy(chans).to(device))
            self._tables[device] = t
        return t

    def __call__(self, raw_masks):
        """raw_masks: float32 CUDA tensor [N, C, H, W] (channels first, i.e. after PermuteMasksDim) -> [N, K+1, Ho, Wo]."""
        if raw_masks.device.type != 'cuda':
            raise nv.NativeError('MaskPreprocessor runs on an MI355X only (no CPU fallback)')
        nv.init_device()
        x = raw_masks.contiguous().float()
        n, c, h, w = x.shape
        k = len(self.groups) if self.groups is not None else c
        if self.groups is not None and max(max(g) for g in self.groups) >= c:
            raise ValueError('grouping refers to channel %d but the masks have %d channels' % (max(max(g) for g in self.groups), c))
        out = torch.empty(n, k + 1, self.size[0], self.size[1], device=x.device, dtype=torch.float32)
        t = self._group_tables(x.device)
        nv.call('bpb_mask_preprocess', x.data_ptr(), nv.ptr(t[0]) if t else None, nv.ptr(t[1]) if t else None, n, c, h, w, k,
                self.size[0], self.size[1], self.combine_sum, self.bg, self.softmax_weight, self.threshold, out.data_ptr(),
                nv.stream())
        return out


class RandomIdentitySampler(torch.utils.data.Sampler):
    """Batches of `batch_size // num_instances` identities x `num_instances` images (sampler.py:11-76).

    data_source: sequence of dicts with a 'pid' entry.  Identities with fewer than `num_instances` images are over-sampled
    with replacement; an epoch ends when fewer identities than a batch needs still have unused chunks."""

    def __init__(self, data_source, batch_size, num_instances):
        if batch_size < num_instances:
            raise ValueError('batch_size={} must be no less than num_instances={}'.format(batch_size, num_instances))
        self.data_source, self.batch_size, self.num_instances = data_source, batch_size, num_instances
        self.num_pids_per_batch = batch_size // num_instances
        self.index_dic = defaultdict(list)
        for index, sample in enumerate(data_source):
            self.index_dic[sample['pid']].append(index)
        self.pids = list(self.index_dic)
        self.length = sum(max(len(v), num_instances) - max(len(v), num_instances) % num_instances for v in self.index_dic.values())

    def __iter__(self):
        k = self.num_instances
        chunks = {}
        for pid in self.pids:                                   # RNG consumption order = the reference's (choice, shuffle)
            idxs = copy.deepcopy(self.index_dic[pid])
            if len(idxs) < k:
                idxs = np.random.choice(idxs, size=k, replace=True)
            random.shuffle(idxs)
            idxs = list(idxs)
            chunks[pid] = [idxs[i:i + k] for i in range(0, len(idxs) - len(idxs) % k, k)]
        avail = copy.deepcopy(self.pids)
        order = []
        while len(avail) >= self.num_pids_per_batch:
            for pid in random.sample(avail, self.num_pids_per_batch):
                order.extend(chunks[pid].pop(0))
                if not chunks[pid]:
                    avail.remove(pid)
        return iter(order)

    def __len__(self):
        return self.length
