"""HRNet (W32 / W48 / any width) and ResNet-50 trunks for the MI355X plan executor.

Each class is (a) a *parameter holder* whose state-dict keys equal the reference's
(torchreid/models/hrnet.py:314-576, torchreid/models/resnet.py:157-358) so the
authors' checkpoints load unchanged, and (b) an ``emit(net, x)`` method that records the network into a
``graph.Net`` launch plan instead of executing nn.Module calls.  The nn.Conv2d / nn.BatchNorm2d children
are never called: they only own the tensors (which live in one flat arena, see model.py).
"""
import os

import torch.nn as nn


def _bn_tuple(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def _emit_cb(net, x, conv, bn):
    """conv + BatchNorm batch statistics; the affine itself is applied by the consuming fuse op."""
    return net.conv(x, conv.weight, conv.stride[0], conv.padding[0], bias=conv.bias, bn=_bn_tuple(bn))


def _cb(cin, cout, k, stride=1, relu=False, bias=False):
    mods = [nn.Conv2d(cin, cout, k, stride, k // 2, bias=bias), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class Residual(nn.Module):
    """BasicBlock (hrnet.py:67-96) or Bottleneck (hrnet.py:99-137, resnet.py:105-154)."""

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
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.cout = cout

    def emit(self, net, x):
        c = _emit_cb(net, x, self.conv1, self.bn1)
        a = net.fuse([(c, 0)], relu=True)
        c = _emit_cb(net, a, self.conv2, self.bn2)
        if self.bottleneck:
            a = net.fuse([(c, 0)], relu=True)
            c = _emit_cb(net, a, self.conv3, self.bn3)
        skip = x if self.downsample is None else _emit_cb(net, x, self.downsample[0], self.downsample[1])
        return net.fuse([(c, 0), (skip, 0)], relu=True)      # bn(conv) + residual, then ReLU: one pass


def _chain(cin, planes, n, bottleneck, stride=1):
    units = [Residual(cin, planes, bottleneck, stride)]
    for _ in range(n - 1):
        units.append(Residual(units[0].cout, planes, bottleneck))
    return nn.Sequential(*units)


def _emit_chain(net, chain, x):
    for unit in chain:
        x = unit.emit(net, x)
    return x


def _emit_parallel_chains(net, chains, xs):
    """Independent chains (the branches of a module, the head's incre modules) are recorded on separate stream slots
    between a fork and a join so that low-resolution branches overlap with the high-resolution one."""
    n = len(chains)
    net.fork(n)
    outs = []
    for i, (chain, x) in enumerate(zip(chains, xs)):
        net.set_slot(i)
        outs.append(_emit_chain(net, chain, x))
    net.set_slot(0)
    net.join(n)
    return outs


EXCHANGE_PATHS = True      # one merge chain per exchange PATH (round 3: -1.2 ms per step against one chain per target); tools flip it


class MultiResModule(nn.Module):
    """HighResolutionModule (hrnet.py:140-279)."""

    def __init__(self, widths, blocks_per_branch=4):
        super().__init__()
        nb = len(widths)
        self.branches = nn.ModuleList([_chain(w, w, blocks_per_branch, False) for w in widths])
        rows = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(widths[j], widths[i], 1, bias=False), nn.BatchNorm2d(widths[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        last = (k == i - j - 1)
                        steps.append(_cb(widths[j], widths[i] if last else widths[j], 3, 2, relu=not last))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def emit(self, net, xs, first=True, last=True):
        """`first` / `last`: position of the module in its stage.  The branch chains of a module run on the stream slots its
        predecessor's exchange step left open (target i's fused output is produced on slot i and feeds branch i only), so a
        stage has ONE stream barrier per module -- between the branches and the exchange, where every target needs every
        branch -- instead of two."""
        nb = len(self.fuse_layers)
        if first:
            net.fork(nb)
        ys = []
        for i, (chain, x) in enumerate(zip(self.branches, xs)):
            net.set_slot(i)
            if not first:
                net.align()          # the exchange paths into branch i precede its blocks on this chain: re-synchronise the merge
            ys.append(_emit_chain(net, chain, x))
        net.set_slot(0)
        net.join(nb)
        xs = ys
        # Exchange step.  Down-paths of different targets share nothing, up-paths are 1x1 conv + BN whose nearest upsample is
        # folded into the fuse read (no upsampled tensor is ever written).  The paths into target i (up to three small convs
        # each, 16 convs + 24 BatchNorm/fuse launches in a 4-branch module) and its final fuse are recorded on stream slot i:
        # recorded on one stream they were ~10 ms of strictly serial small launches per train step.
        def path(i, j, x):
            row = self.fuse_layers[i]
            if j > i:
                return (_emit_cb(net, x, row[j][0], row[j][1]), j - i)
            t = x
            steps = list(row[j])
            for k, st in enumerate(steps):
                c = _emit_cb(net, t, st[0], st[1])
                if k < len(steps) - 1:
                    t = net.fuse([(c, 0)], relu=True)
            return (c, 0)

        if EXCHANGE_PATHS and nb * (nb - 1) <= 16:
            # Round 3: every PATH j -> i is its own chain of a first region (nb * (nb - 1) chains), the nb final sums form a
            # second one.  Recorded per target, the three down-paths into the deepest branch (3 + 2 + 1 strided convolutions)
            # were six convolution rounds of the lock-step merge; per path, step k of every path shares a round: three rounds, each
            # a grouped launch over up to six problems instead of single launches that cannot fill the chip (22 us at 54 TFLOP/s).
            net.fork(nb * (nb - 1))
            produced, slot = {}, 0
            for i in range(nb):
                for j, x in enumerate(xs):
                    if j != i:
                        net.set_slot(slot)
                        slot += 1
                        produced[(i, j)] = path(i, j, x)
            net.set_slot(0)
            net.join(nb * (nb - 1))
            net.fork(nb)
            outs = []
            for i in range(nb):
                net.set_slot(i)
                outs.append(net.fuse([(x, 0) if j == i else produced[(i, j)] for j, x in enumerate(xs)], relu=True))
        else:
            net.fork(nb)
            outs = []
            for i in range(nb):
                net.set_slot(i)
                outs.append(net.fuse([(x, 0) if j == i else path(i, j, x) for j, x in enumerate(xs)], relu=True))
        net.set_slot(0)
        if last:
            net.join(nb)
        return outs


class HRNet(nn.Module):
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
            for i, w in enumerate(cur):
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
        head = [32, 64, 128, 256]
        self.incre_modules = nn.ModuleList([_chain(w, h, 1, True) for w, h in zip(widths, head)])
        self.layers_out_channels = sum(h * 4 for h in head)
        self.cls_head = _cb(self.layers_out_channels, dim_reduction_channels, 1, relu=True, bias=True)
        self.enable_dim_reduction = enable_dim_reduction       # dim_reduce='before_pooling': hrnet.py:361-380, :574-575
        self.feature_dim = dim_reduction_channels if enable_dim_reduction else self.layers_out_channels
        self.nstages = len(modules)
        self.reduction = 4

    def emit(self, net, x):
        c = _emit_cb(net, x, self.conv1, self.bn1)
        x = net.fuse([(c, 0)], relu=True)
        c = _emit_cb(net, x, self.conv2, self.bn2)
        x = net.fuse([(c, 0)], relu=True)
        ys = [_emit_chain(net, self.layer1, x)]
        for s in range(self.nstages):
            trans = getattr(self, 'transition%d' % (s + 1))
            xs = []
            for i, t in enumerate(trans):
                if t is None:
                    xs.append(ys[i])
                    continue
                src = ys[0] if s == 0 else ys[-1]           # hrnet.py:541-563
                if isinstance(t[0], nn.Conv2d):              # same-resolution transition: conv, bn, relu
                    src = net.fuse([(_emit_cb(net, src, t[0], t[1]), 0)], relu=True)
                else:                                        # chain of strided conv-bn-relu
                    for st in t:
                        src = net.fuse([(_emit_cb(net, src, st[0], st[1]), 0)], relu=True)
                xs.append(src)
            stage = list(getattr(self, 'stage%d' % (s + 2)))
            for m, mod in enumerate(stage):
                xs = mod.emit(net, xs, first=m == 0, last=m == len(stage) - 1)
            ys = xs
        # head: the channel-increasing bottleneck of every branch (one chain per branch, merged lock-step), then the bilinear
        # up-sampling of all of them into the concatenated map (hrnet.py:565-573)
        n = len(self.incre_modules)
        net.fork(n)
        tops = []
        for i, (chain, y) in enumerate(zip(self.incre_modules, ys)):
            net.set_slot(i)
            tops.append(_emit_chain(net, chain, y))
        net.set_slot(0)
        net.join(n)
        out = net.concat_bilinear(tops)        # one launch writes whole pixel rows of the concatenated map (+ its channel stats)
        assert out.C == self.layers_out_channels
        if self.enable_dim_reduction:          # cls_head: 1x1 conv (with bias) + BN + ReLU on the concatenated map
            out = net.fuse([(_emit_cb(net, out, self.cls_head[0], self.cls_head[1]), 0)], relu=True)
        return out


class ResNet50(nn.Module):
    def __init__(self, num_classes, last_stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = _chain(64, 64, 3, True)
        self.layer2 = _chain(256, 128, 4, True, 2)
        self.layer3 = _chain(512, 256, 6, True, 2)
        self.layer4 = _chain(1024, 512, 3, True, last_stride)
        self.classifier = nn.Linear(2048, num_classes)   # present in the reference state dict, unused on this path
        self.feature_dim = 2048

    def emit(self, net, x):
        c = _emit_cb(net, x, self.conv1, self.bn1)
        x = net.fuse([(c, 0)], relu=True)
        x = net.maxpool(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = _emit_chain(net, layer, x)
        return x


def build_backbone(name, num_classes, last_stride=1, enable_dim_reduction=False, dim_reduction_channels=256, **_):
    if name == 'hrnet32':
        return HRNet((32, 64, 128, 256), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name == 'hrnet48':
        return HRNet((48, 96, 192, 384), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name.startswith('hrnet_w'):
        w = int(name[len('hrnet_w'):])
        return HRNet((w, 2 * w, 4 * w, 8 * w), enable_dim_reduction=enable_dim_reduction,
                     dim_reduction_channels=dim_reduction_channels)
    if name == 'resnet50':
        return ResNet50(num_classes, last_stride)
    raise KeyError('Unknown backbone: %s' % name)
