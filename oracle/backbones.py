"""Pure-torch CPU restatement of the two backbones on the path (TEST INFRASTRUCTURE).

* HRNet trunk, width-generic -- restates torchreid/models/hrnet.py:67-137 (blocks),
  :140-279 (multi-resolution module), :314-576 (network + forward).  W32 = widths
  (32,64,128,256); W48 = (48,96,192,384) (same class, other widths).
* ResNet-50 trunk with ``last_stride`` -- restates torchreid/models/resnet.py:105-154
  (bottleneck), :157-290 (network), :342-358 (featuremaps/forward, loss='part_based').

State-dict key names equal the reference's (that is the checkpoint compatibility surface,
SURVEY.md section 8b); the construction code is our own.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _cb(cin, cout, k, stride=1, relu=False, bias=False):
    """conv(k, pad k//2) + BN (+ReLU) as an nn.Sequential with child names '0','1'(,'2')."""
    mods = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=bias), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class Residual(nn.Module):
    """Basic (2x conv3x3, expansion 1) or bottleneck (1x1-3x3-1x1, expansion 4) residual unit.

    hrnet.py:67-137 / resnet.py:105-154.  `stride` sits on the 3x3 conv of the bottleneck
    (resnet.py:124-126) and on conv1 of the basic block.
    """

    def __init__(self, cin, planes, bottleneck, stride=1):
        super().__init__()
        self.bottleneck = bottleneck
        cout = planes * (4 if bottleneck else 1)
        if bottleneck:
            self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, cout, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(cout)
        else:
            self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))
        self.cout = cout

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        if self.bottleneck:
            y = F.relu(self.bn2(self.conv2(y)))
            y = self.bn3(self.conv3(y))
        else:
            y = self.bn2(self.conv2(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


def _chain(cin, planes, n, bottleneck, stride=1):
    units = [Residual(cin, planes, bottleneck, stride)]
    for _ in range(n - 1):
        units.append(Residual(units[0].cout, planes, bottleneck))
    return nn.Sequential(*units)


class MultiResModule(nn.Module):
    """hrnet.py:140-279 -- parallel branches of 4 basic blocks, then all-to-all fusion."""

    def __init__(self, widths, blocks_per_branch=4):
        super().__init__()
        nb = len(widths)
        self.branches = nn.ModuleList([_chain(w, w, blocks_per_branch, False) for w in widths])
        fuse = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:      # lower resolution -> 1x1 conv + BN, nearest upsample (hrnet.py:221-231)
                    row.append(nn.Sequential(nn.Conv2d(widths[j], widths[i], 1, bias=False),
                                             nn.BatchNorm2d(widths[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:          # higher resolution -> (i-j) strided 3x3 convs (hrnet.py:234-254)
                    steps = []
                    for k in range(i - j):
                        last = (k == i - j - 1)
                        steps.append(_cb(widths[j], widths[i] if last else widths[j], 3, 2, relu=not last))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        out = []
        for i, row in enumerate(self.fuse_layers):
            acc = None
            for j, x in enumerate(xs):
                t = x if row[j] is None else row[j](x)
                acc = t if acc is None else acc + t
            out.append(F.relu(acc))
        return out


class HRNet(nn.Module):
    """hrnet.py:314-576.  Output: concat of the four incre-module outputs, bilinearly
    (align_corners=True) upsampled to the highest resolution -> 1920 channels at 1/4 res."""

    def __init__(self, widths=(32, 64, 128, 256), modules=(1, 4, 3), enable_dim_reduction=False,
                 dim_reduction_channels=256):
        super().__init__()
        widths = list(widths)
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.layer1 = _chain(64, 64, 4, True)
        prev = [256]
        for s, nmod in enumerate(modules):
            cur = widths[:s + 2]
            trans = []
            for i, w in enumerate(cur):               # hrnet.py:449-483
                if i < len(prev):
                    trans.append(_cb(prev[i], w, 3, 1, relu=True) if prev[i] != w else None)
                else:
                    steps = []
                    for j in range(i + 1 - len(prev)):
                        steps.append(_cb(prev[-1], w if j == i - len(prev) else prev[-1], 3, 2, relu=True))
                    trans.append(nn.Sequential(*steps))
            setattr(self, 'transition%d' % (s + 1), nn.ModuleList(trans))
            setattr(self, 'stage%d' % (s + 2), nn.Sequential(*[MultiResModule(cur) for _ in range(nmod)]))
            prev = cur
        head = [32, 64, 128, 256]                     # hrnet.py:400-414 (fixed, width independent)
        self.incre_modules = nn.ModuleList([_chain(w, h, 1, True) for w, h in zip(widths, head)])
        self.layers_out_channels = sum(h * 4 for h in head)
        self.cls_head = _cb(self.layers_out_channels, dim_reduction_channels, 1, relu=True, bias=True)
        self.enable_dim_reduction = enable_dim_reduction
        self.feature_dim = dim_reduction_channels if enable_dim_reduction else self.layers_out_channels
        self.nstages = len(modules)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        ys = [self.layer1(x)]
        for s in range(self.nstages):
            trans = getattr(self, 'transition%d' % (s + 1))
            xs = []
            for i, t in enumerate(trans):             # hrnet.py:541-563: new branch is fed from ys[-1]
                if t is None:
                    xs.append(ys[i])
                else:
                    xs.append(t(ys[-1] if i >= len(ys) or s > 0 else ys[0]))
            ys = getattr(self, 'stage%d' % (s + 2))(xs)
        ys = [m(y) for m, y in zip(self.incre_modules, ys)]
        size = ys[0].shape[2:]
        ups = [ys[0]] + [F.interpolate(y, size=size, mode='bilinear', align_corners=True) for y in ys[1:]]
        x = torch.cat(ups, 1)
        if self.enable_dim_reduction:
            x = self.cls_head(x)
        return x


class ResNet50(nn.Module):
    """resnet.py:157-358 with Bottleneck [3,4,6,3]; returns the layer4 map (loss='part_based')."""

    def __init__(self, num_classes, last_stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _chain(64, 64, 3, True)
        self.layer2 = _chain(256, 128, 4, True, 2)
        self.layer3 = _chain(512, 256, 6, True, 2)
        self.layer4 = _chain(1024, 512, 3, True, last_stride)
        self.classifier = nn.Linear(2048, num_classes)   # unused on the part-based path (resnet.py:243)
        self.feature_dim = 2048

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def build_backbone(name, num_classes, last_stride=1, enable_dim_reduction=False, dim_reduction_channels=256):
    if name == 'hrnet32':
        return HRNet((32, 64, 128, 256), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name == 'hrnet48':
        return HRNet((48, 96, 192, 384), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name.startswith('hrnet_w'):          # test-only narrow variants, e.g. 'hrnet_w8'
        w = int(name[len('hrnet_w'):])
        return HRNet((w, 2 * w, 4 * w, 8 * w), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name == 'resnet50':
        return ResNet50(num_classes, last_stride)
    raise KeyError(name)
