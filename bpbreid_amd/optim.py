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
        return a

    def _table(self, arena):
        """Block table (offset, length) covering the arena slices of parameters that have a gradient."""
        key = tuple(p.grad is not None for p in arena['params'])
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
        offs, lens, nblocks = self._table(a)
        if nblocks == 0:
            return
        self.step_index += 1
        lr = self.param_groups[0]['lr']
        nv.call('bpb_adam_step', a['param'].data_ptr(), a['grad'].data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                offs.data_ptr(), lens.data_ptr(), nblocks, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                self.step_index, grad_scale, self.step_dev.data_ptr(), nv.stream())

    def state_dict(self):
        return {'step': self.step_index, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'lr': self.param_groups[0]['lr']}

    def load_state_dict(self, sd):
        self._state()
        self.step_index = sd['step']
        self.step_dev.fill_(self.step_index)
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0]['lr'] = sd['lr']


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
