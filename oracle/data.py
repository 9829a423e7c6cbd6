"""CPU restatement of the reference's mask preprocessing chain (TEST INFRASTRUCTURE).

torchreid/data/masks_transforms/mask_transform.py: MaskGroupingTransform :21-38, AddBackgroundMask :58-80, ResizeMasks
:45-52, applied per sample in the order of torchreid/data/transforms.py:133-158 (grouping -> background -> resize).
"""
import torch
import torch.nn.functional as F


def group_masks(masks, groups, combine_mode='max'):
    """masks [C,H,W]; groups: list of channel-index lists -> [K,H,W] (mask_transform.py:31-38)."""
    out = []
    for g in groups:
        sel = masks[g]
        out.append((sel.sum(0) if combine_mode == 'sum' else sel.max(0)[0]).clamp(0, 1))
    return torch.stack(out)


def add_background(masks, strategy='sum', softmax_weight=0, threshold=0.3):
    """mask_transform.py:65-80."""
    if strategy == 'sum':
        bg = (1 - masks.sum(0)).clamp(0, 1)
    elif strategy == 'threshold':
        bg = masks.max(0)[0] < threshold
    elif strategy == 'diff_from_max':
        bg = (1 - masks.max(0)[0]).clamp(0, 1)
    else:
        raise ValueError(strategy)
    masks = torch.cat([bg.unsqueeze(0), masks])
    return F.softmax(masks * softmax_weight, dim=0) if softmax_weight > 0 else masks / masks.sum(0)


def preprocess_masks(raw, height, width, mask_scale=4, groups=None, combine_mode='max', strategy='threshold',
                     softmax_weight=15, threshold=0.5):
    """raw [N,C,H,W] -> [N,K+1,H/scale,W/scale], sample by sample like the reference's dataset transform."""
    size = (int(height / mask_scale), int(width / mask_scale))
    outs = []
    for m in raw:
        if groups is not None:
            m = group_masks(m, groups, combine_mode)
        m = add_background(m, strategy, softmax_weight, threshold)
        outs.append(F.interpolate(m.unsqueeze(0), size, mode='nearest').squeeze(0))
    return torch.stack(outs)
