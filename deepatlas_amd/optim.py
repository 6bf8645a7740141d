"""Adam over one flat fp32 bucket (HIP kernel), API-compatible with torch.optim.Adam as used at
models/segmentation.py:91 (lr, betas=(0.9, 0.999), eps=1e-8, no weight decay / amsgrad).

All parameters are re-pointed into one contiguous buffer and their .grad into another, so that
  * the optimiser step is ONE kernel launch over every parameter,
  * data-parallel training all-reduces ONE flat gradient bucket over RCCL.
state_dict()/load_state_dict() keep torch.optim.Adam's format (per-parameter step / exp_avg / exp_avg_sq).
"""
import torch

from . import ops

from . import _native as nat
from ._native import call, ptr, stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        ps = [p for g in self.param_groups for p in g['params']]
        if not ps:
            raise ValueError('no parameters')
        nat.require_cuda(*ps)
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._steps = 0
        self._slices = []
        off = 0
        for p in ps:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            self.state[p] = {'step': torch.tensor(0.0),
                             'exp_avg': self.flat_m[off:off + k].view(p.shape),
                             'exp_avg_sq': self.flat_v[off:off + k].view(p.shape)}
            self._slices.append((p, off, k))
            off += k

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat bucket: zero it in one memset and keep the views attached."""
        ops.join_side_stream()                      # asynchronous weight gradients of the previous step have landed
        self.flat_g.zero_()
        for p, off, k in self._slices:
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + k].view(p.shape)

    def _gather_stray_grads(self):
        ops.join_side_stream()                      # weight gradients accumulated on the side stream (ops.ASYNC_WGRAD)
        # autograd may have replaced a .grad view (e.g. first backward after set_to_none); fold it back
        for p, off, k in self._slices:
            if p.grad is not None and p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                self.flat_g[off:off + k].copy_(p.grad.reshape(-1))
                p.grad = self.flat_g[off:off + k].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        self._gather_stray_grads()
        g = self.param_groups[0]
        self._steps += 1
        call('da_adam_step', ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.flat_p.numel(),
             float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self._steps, float(grad_scale), stream())
        for p, _, _ in self._slices:
            self.state[p]['step'] += 1
        return loss

    def load_state_dict(self, state_dict):
        sd = state_dict
        ids = [i for grp in sd['param_groups'] for i in grp['params']]
        for (p, off, k), i in zip(self._slices, ids):
            st = sd['state'].get(i)
            if st is None:
                continue
            self.flat_m[off:off + k].copy_(st['exp_avg'].reshape(-1))
            self.flat_v[off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
            self.state[p]['step'] = torch.tensor(float(st['step']))
            self._steps = int(float(st['step']))
        for grp, sg in zip(self.param_groups, sd['param_groups']):
            for key in ('lr', 'betas', 'eps'):
                if key in sg:
                    grp[key] = sg[key]
