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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('FlatAdam implements torch.optim.Adam as the reference uses it (models/segmentation.py:91): '
                                      'no weight decay, no amsgrad')
        params = [p for p in params]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError('FlatAdam updates ONE flat bucket with one (lr, betas, eps): pass a single parameter group')
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
        ops.register_flat_bucket(self.flat_g)
        ops.register_flat_params(self.flat_p)
        ops.bump_weights_epoch()                   # the parameters just moved into the bucket
        # device copy of this step's scalars for the graph-capturable update (da_adam_step_dev): refreshed before every replay
        self._dev_state = torch.zeros(6, dtype=torch.float32, device=dev)
        self.device_step = False          # True inside a captured step: step() launches da_adam_step_dev
        self._steps = 0
        self._slices = []
        off = 0
        # Tagged convolution weights (ops.tag_conv_layouts) are stored TAP-MAJOR in all four buckets -- the kernels' own layout -- and the
        # module sees a strided view of the reference's shape (ops.param_view): no per-step layout conversion in either direction.
        for p in ps:
            k = p.numel()
            ops.param_view(self.flat_p, off, p).copy_(p.detach())
            p.data = ops.param_view(self.flat_p, off, p)
            p.grad = ops.param_view(self.flat_g, off, p)
            self.state[p] = {'step': torch.tensor(0.0),
                             'exp_avg': ops.param_view(self.flat_m, off, p),
                             'exp_avg_sq': ops.param_view(self.flat_v, off, p)}
            self._slices.append((p, off, k))
            off += k

    def sync_device_state(self, grad_scale=1.0):
        """Host -> device copy of the scalars of the step about to run ({lr / bc1, betas, eps, 1 / sqrt(bc2), grad_scale}, from
        da_adam_host_state) ahead of a graph replay.  One 24-byte copy; lr schedulers keep working because lr is re-read every step."""
        import ctypes
        g = self.param_groups[0]
        host = (ctypes.c_float * 6)()
        nat.call('da_adam_host_state', float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self._steps + 1,
                 float(grad_scale), ctypes.cast(host, ctypes.c_void_p))
        self._dev_state.copy_(torch.tensor(list(host), dtype=torch.float32))

    def note_replayed_step(self):
        """Book-keeping of a step that ran inside a replayed graph (the host-side part of step())."""
        ops.bump_weights_epoch()
        self._steps += 1
        for p, _, _ in self._slices:
            if p.requires_grad:
                self.state[p]['step'] += 1

    def _set_step_count(self, n):
        self._steps = int(n)
        for p, _, _ in self._slices:
            self.state[p]['step'] = torch.tensor(float(n))

    def add_param_group(self, param_group):
        if getattr(self, 'flat_p', None) is not None:
            raise NotImplementedError('FlatAdam: parameters are fixed at construction (they live in one flat bucket)')
        super().add_param_group(param_group)

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat bucket: zero it in one memset and keep the views attached (`set_to_none` would detach them
        from the bucket the all-reduce and the Adam kernel work on, so it is ignored)."""
        ops.join_side_stream()                      # asynchronous weight gradients of the previous step have landed
        ops.drop_bwd_stats()                        # (producer-side BatchNorm-backward sums nobody consumed: release the gradient tensors they pin)
        self.flat_g.zero_()
        for p, off, k in self._slices:
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = ops.param_view(self.flat_g, off, p)
            p._da_gz = True                         # this parameter's gradient slice is all zeros: its first weight gradient may be WRITTEN there

    def _gather_stray_grads(self):
        ops.join_side_stream()                      # weight gradients accumulated on the side stream (ops.ASYNC_WGRAD)
        # autograd may have replaced a .grad view (e.g. first backward after set_to_none); fold it back
        for p, off, k in self._slices:
            if p.grad is not None and p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                ops.param_view(self.flat_g, off, p).copy_(p.grad)
                p.grad = ops.param_view(self.flat_g, off, p)
                p._da_gz = False                    # the slice is no longer all zeros

    def grads_written(self, params=None):
        """Tell the optimiser that gradient slices were written from OUTSIDE this package after zero_grad() (a hook that adds into
        `p.grad`, manual gradient accumulation across micro-batches through `p.grad.add_`): the next weight gradient of those
        parameters is then accumulated instead of written straight over the (assumed all-zero) slice.  Every writer inside the package
        (ops.WgradTarget, ops.grad_for_autograd, _gather_stray_grads) keeps the `_da_gz` marks itself."""
        for p, _, _ in self._slices:
            if params is None or any(p is q for q in params):
                p._da_gz = False

    def _frozen_mask(self):
        """1.0 for elements whose parameter takes part in the update, 0.0 for frozen ones (requires_grad=False or no gradient this
        step: torch.optim.Adam skips those entirely, moments included).  None when every parameter is live (the usual case)."""
        frozen = [(off, k) for p, off, k in self._slices if (not p.requires_grad) or p.grad is None]
        if not frozen:
            return None
        mask = torch.ones_like(self.flat_p)
        for off, k in frozen:
            mask[off:off + k] = 0
        return mask

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        ops.flush_batches_tracked()
        ops.drop_bwd_stats()
        self._gather_stray_grads()
        g = self.param_groups[0]
        for p, off, k in self._slices:
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * off:
                raise RuntimeError('FlatAdam: a parameter no longer lives in the flat bucket (model.to()/.half() after constructing the '
                                   'optimiser?); build the optimiser after moving the model')
        mask = self._frozen_mask()
        ops.bump_weights_epoch(self.flat_p)        # cached weight layouts (ops.weight_tio) and kept packs of THIS bucket are stale from here on
        if self.device_step:
            # being captured into a HIP graph (graphs.GraphedStep): step count and hyper-parameters come from device memory; the
            # host-side book-keeping happens per REPLAY in note_replayed_step()
            if mask is not None:
                raise NotImplementedError('FlatAdam: frozen parameters are not supported inside a captured step')
            call('da_adam_step_dev', ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.flat_p.numel(),
                 ptr(self._dev_state), stream())
            return loss
        if mask is not None:                        # frozen slices: keep p, m, v exactly as they are
            keep = (self.flat_p.clone(), self.flat_m.clone(), self.flat_v.clone())
        self._steps += 1
        call('da_adam_step', ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.flat_p.numel(),
             float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), self._steps, float(grad_scale), stream())
        if mask is not None:
            live = mask.bool()
            for cur, old in zip((self.flat_p, self.flat_m, self.flat_v), keep):
                cur.copy_(torch.where(live, cur, old))
        for p, _, _ in self._slices:
            if p.requires_grad and p.grad is not None:
                self.state[p]['step'] += 1
        ops.repack_after_step(self.flat_p)          # the kept packed operands of these weights, re-filled beside the next forward pass (side stream)
        return loss

    def load_state_dict(self, state_dict):
        sd = state_dict
        ids = [i for grp in sd['param_groups'] for i in grp['params']]
        for (p, off, k), i in zip(self._slices, ids):
            st = sd['state'].get(i)
            if st is None:
                continue
            ops.param_view(self.flat_m, off, p).copy_(st['exp_avg'])
            ops.param_view(self.flat_v, off, p).copy_(st['exp_avg_sq'])
            self.state[p]['step'] = torch.tensor(float(st['step']))
            self._steps = int(float(st['step']))
        for grp, sg in zip(self.param_groups, sd['param_groups']):
            for key in ('lr', 'betas', 'eps'):
                if key in sg:
                    grp[key] = sg[key]
