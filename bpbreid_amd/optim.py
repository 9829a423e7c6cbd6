"""Fused multi-tensor Adam over the model's flat arenas + the reference's warm-up multi-step LR schedule.

torch.optim.Adam semantics as configured by the reference (coupled L2 weight decay; lr 3.5e-4, wd 5e-4,
betas (0.9, 0.999); torchreid/optim/optimizer.py:113-119, scripts/default_config.py:125-127,153-155), executed
as ONE kernel launch over every parameter that received a gradient this step (csrc/optim.hip).  Parameters
whose ``.grad`` is None are skipped exactly like torch does.
"""
import torch

from . import native as nv

BLOCK = 1024


class FusedAdam:
    def __init__(self, model, lr=3.5e-4, weight_decay=5e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps
        self.step_index = 0
        self._tables = {}
        self.updated = set()                      # positions (model.parameters() order) of parameters that have been stepped
        self._last_key = None
        self.exp_avg = None
        self.param_groups = [{'lr': lr}]          # so that LR schedulers written for torch.optim can drive it

    def zero_grad(self, set_to_none=True):
        """Gradients are overwritten (not accumulated) by every backward: nothing to do."""

    def _state(self):
        a = self.model.arena()
        if self.exp_avg is None or self.exp_avg.data_ptr() == 0 or self.exp_avg.numel() != a['param'].numel():
            self.exp_avg = torch.zeros_like(a['param'])
            self.exp_avg_sq = torch.zeros_like(a['param'])
            self.step_dev = torch.full((1,), self.step_index, device=a['param'].device, dtype=torch.int32)
            self.lr_dev = torch.full((1,), float(self.param_groups[0]['lr']), device=a['param'].device, dtype=torch.float32)
            self._lr_uploaded = float(self.param_groups[0]['lr'])
        return a

    def sync_lr(self):
        """Upload the learning rate if a scheduler changed it.  The Adam kernel reads it from device memory, so a captured
        step (engine.capture_step) keeps following the schedule; call this OUTSIDE the captured region (replay does)."""
        self._state()
        lr = float(self.param_groups[0]['lr'])
        if lr != self._lr_uploaded:
            self.lr_dev.fill_(lr)
            self._lr_uploaded = lr

    def _table(self, arena, key):
        """Block table (offset, length) covering the arena slices of parameters that have a gradient."""
        tab = self._tables.get(key)
        if tab is None:
            offs, lens = [], []
            for has, (off, n) in zip(key, self.model._param_slices):
                if not has:
                    continue
                for b in range(0, n, BLOCK):
                    offs.append(off + b)
                    lens.append(min(BLOCK, n - b))
            dev = arena['param'].device
            tab = (torch.tensor(offs, dtype=torch.int64, device=dev), torch.tensor(lens, dtype=torch.int32, device=dev), len(offs))
            self._tables[key] = tab
        return tab

    def step(self, grad_scale=1.0):
        a = self._state()
        key = tuple(p.grad is not None for p in a['params'])
        offs, lens, nblocks = self._table(a, key)
        if nblocks == 0:
            return
        self.step_index += 1
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        lr = self.param_groups[0]['lr']
        if key != self._last_key:                  # remember which parameters have ever been stepped (state_dict)
            self.updated |= {k for k, has in enumerate(key) if has}
            self._last_key = key
        nv.call('bpb_adam_step', a['param'].data_ptr(), a['grad'].data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                offs.data_ptr(), lens.data_ptr(), nblocks, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                self.step_index, grad_scale, self.step_dev.data_ptr(), self.lr_dev.data_ptr(), nv.stream())
        if hasattr(self.model, 'bump_param_version'):
            self.model.bump_param_version()        # eval-plan weights derived from the parameters are stale now

    def state_dict(self):
        """torch.optim.Adam's format ({'state': {i: {'step','exp_avg','exp_avg_sq'}}, 'param_groups': [...]}, positions =
        model.parameters() order), so that checkpoints are interchangeable with the reference's (engine.py:96).  Only
        parameters that have been updated carry state, exactly like torch."""
        self._state()
        params = list(self.model.parameters())
        state = {}
        if self.step_index > 0:
            for i, (p, (off, n)) in enumerate(zip(params, self.model._param_slices)):
                if i not in self.updated:           # update history, not the current .grad (which a later backward may change)
                    continue
                state[i] = {'step': torch.tensor(float(self.step_index)),
                            'exp_avg': self.exp_avg[off:off + n].view_as(p).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + n].view_as(p).clone()}
        group = {'lr': self.param_groups[0]['lr'], 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'params': list(range(len(params)))}
        if 'initial_lr' in self.param_groups[0]:
            group['initial_lr'] = self.param_groups[0]['initial_lr']
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd, param_names=None):
        """Accepts torch.optim.Adam's state dict.  `param_names`: the parameter names in the order the saving optimizer saw
        them (checkpoint.parameter_names) -- positions are matched by NAME, so a reference checkpoint loads even though this
        model registers its sub-modules in a different order; default = this model's own order."""
        a = self._state()
        if 'state' not in sd:                       # flat format of earlier versions of this class
            self.step_index = int(sd['step'])
            self.exp_avg.copy_(sd['exp_avg'])
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            self.param_groups[0]['lr'] = sd['lr']
            self.step_dev.fill_(self.step_index)
            # the flat format carries no per-parameter history: every parameter whose second moment is non-zero has been
            # updated (state_dict() filters on `updated`; without this a resume + save would drop the Adam moments)
            self.updated = set()
            if self.step_index > 0:
                nz = (self.exp_avg_sq != 0).cpu()      # (checkpoint load, not the step: one device read)
                for i, (off, n) in enumerate(self.model._param_slices):
                    if n and bool(nz[off:off + n].any()):
                        self.updated.add(i)
            self._last_key = None
            return
        own = {name: k for k, (name, _) in enumerate(self.model.named_parameters())}
        names = param_names if param_names is not None else list(own)
        order = sd['param_groups'][0]['params'] if sd.get('param_groups') else list(range(len(names)))
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.updated = set()
        self._last_key = None
        steps = set()
        params = list(self.model.parameters())
        for pos, pid in enumerate(order):
            st = sd['state'].get(pid)
            if st is None:
                continue
            k = own[names[pos]]
            off, n = self.model._param_slices[k]
            assert st['exp_avg'].numel() == n, 'optimizer state of %s has the wrong size' % names[pos]
            self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
            self.updated.add(k)
        # one step counter for the whole arena (the reference's parameters all step together: unused ones never step)
        assert len(steps) <= 1, 'parameters with different step counts are not supported: %s' % sorted(steps)
        self.step_index = steps.pop() if steps else 0
        self.step_dev.fill_(self.step_index)
        if sd.get('param_groups'):
            g = sd['param_groups'][0]
            self.param_groups[0]['lr'] = g['lr']
            if 'initial_lr' in g:
                self.param_groups[0]['initial_lr'] = g['initial_lr']
            self.betas, self.eps, self.weight_decay = tuple(g.get('betas', self.betas)), g.get('eps', self.eps), g.get('weight_decay', self.weight_decay)


class WarmupMultiStepLR:
    """torchreid/optim/lr_scheduler.py:88-131: linear warm-up from warmup_factor over warmup_iters epochs, then
    gamma decay at the milestones (defaults: 10-epoch warm-up x0.01, milestones [40, 70], gamma 0.1)."""

    def __init__(self, optimizer, milestones=(40, 70), gamma=0.1, warmup_factor=0.01, warmup_iters=10, warmup_method='linear'):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.base_lr = optimizer.param_groups[0]['lr']
        self.last_epoch = 0
        self._apply()

    def get_lr(self):
        f = 1.0
        if self.last_epoch < self.warmup_iters:
            if self.warmup_method == 'constant':
                f = self.warmup_factor
            else:
                alpha = self.last_epoch / self.warmup_iters
                f = self.warmup_factor * (1 - alpha) + alpha
        passed = sum(1 for m in self.milestones if m <= self.last_epoch)
        return self.base_lr * f * self.gamma ** passed

    def _apply(self):
        self.optimizer.param_groups[0]['lr'] = self.get_lr()

    def step(self):
        self.last_epoch += 1
        self._apply()

    def state_dict(self):
        """Keys of the reference scheduler's torch state dict (lr_scheduler.py:88-131 subclasses torch's _LRScheduler)."""
        return {'milestones': list(self.milestones), 'gamma': self.gamma, 'warmup_factor': self.warmup_factor,
                'warmup_iters': self.warmup_iters, 'warmup_method': self.warmup_method, 'base_lrs': [self.base_lr],
                'last_epoch': self.last_epoch, '_step_count': self.last_epoch + 1, '_last_lr': [self.get_lr()]}

    def load_state_dict(self, sd):
        self.milestones = sorted(sd.get('milestones', self.milestones))
        for k in ('gamma', 'warmup_factor', 'warmup_iters', 'warmup_method'):
            setattr(self, k, sd.get(k, getattr(self, k)))
        self.base_lr = sd.get('base_lrs', [self.base_lr])[0]
        self.last_epoch = sd['last_epoch']
        self._apply()
