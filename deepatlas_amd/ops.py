"""torch.autograd.Function wrappers around the C-ABI launchers of libdeepatlas_hip.so.

Tensor convention at this layer: the *logical* shape is the reference's N x C x D x H x W, the *physical*
layout is dense channels-last ("NDHWC", torch.channels_last_3d strides).  `ndhwc(x)` returns the
N x D x H x W x C view (copying only when a caller hands in a differently-strided tensor); ops allocate
their outputs as N x D x H x W x C and return the permuted N x C x D x H x W view.

Every op is a HIP kernel launch on the current torch stream; there is no eager fallback.
"""
import os

import torch
from torch.autograd import Function

from . import _native as nat
from ._native import call, call_supported, ptr, stream, workspace


# ------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------
def ndhwc(x):
    """N x C x D x H x W (any strides) -> dense N x D x H x W x C tensor (a view when already channels-last)."""
    nat.require_cuda(x)
    if x.dtype != torch.float32 and x.dtype != torch.bfloat16:
        raise nat.NativeError('deepatlas_amd kernels take fp32 tensors (bf16 for the activations of the bf16 storage mode); got %s' % x.dtype)
    xp = x.permute(0, 2, 3, 4, 1)
    return xp if xp.is_contiguous() else xp.contiguous()


def ncdhw(t):
    """dense N x D x H x W x C -> logical N x C x D x H x W view."""
    return t.permute(0, 4, 1, 2, 3)


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ------------------------------------------------------------------------------------------------
# bf16 activation storage (BASELINE configs[4]; include/deepatlas_hip.h, last section)
# ------------------------------------------------------------------------------------------------
# 'fp32': every tensor fp32 (the reference's arithmetic).  'bf16': the tensors BETWEEN the layers of a network -- convolution / transposed
# convolution outputs, BatchNorm + activation outputs, pooled / up-sampled tensors and all their gradients -- are stored as bf16; network
# inputs, logits, displacement fields, everything the losses see, parameters, their gradients, statistics and every accumulation stay fp32.
# Goes with set_matrix_precision('bf16').  A kernel without a bf16 twin for some shape is bridged by conversion passes (call_act).
ACT_STORAGE_MODES = ('fp32', 'bf16')
ACT_STORAGE = 'fp32'


def set_activation_storage(mode):
    """Process-wide storage type of the network-internal activations; returns the previous mode."""
    global ACT_STORAGE
    if mode not in ACT_STORAGE_MODES:
        raise ValueError("activation storage must be one of %r, got %r" % (ACT_STORAGE_MODES, mode))
    prev, ACT_STORAGE = ACT_STORAGE, mode
    if _LAZY_UP_ENV is None:
        global LAZY_BN_UPSAMPLER
        LAZY_BN_UPSAMPLER = mode == 'bf16'
    return prev


def _act_dtype(channels=None):
    """dtype of a network-internal activation with `channels` channels (thin outputs -- the 3-channel displacement field -- stay fp32)."""
    if ACT_STORAGE == 'bf16' and (channels is None or (channels >= 8 and channels % 4 == 0)):
        return torch.bfloat16
    return torch.float32


class A(object):
    """Marks an activation / gradient tensor among the arguments of call_act (out=True: the kernel writes it)."""
    __slots__ = ('t', 'out')

    def __init__(self, t, out=False):
        self.t, self.out = t, out


def O(t):
    return A(t, True)


# bf16 twins whose activation arguments are not all bf16: expected pattern over the non-None activation arguments, in signature order
_TWIN_PATTERN = {'da_conv1x1_fwd': (1, 0), 'da_conv1x1_fwd_pro': (1, 0), 'da_conv1x1_dgrad': (0, 1), 'da_conv1x1_wgrad': (1, 0),
                 'da_conv1x1_wgrad_pro': (1, 0)}
BF16_FORCE_BRIDGE = False       # tests: take the conversion route even where a bf16 twin exists (A/B of every twin at network level)
bridged_calls = {}       # name -> number of calls that went through conversion passes (diagnostics: tests / bench report it)


def call_act(name, *args, may_decline=False):
    """call() for entry points with activation arguments (wrapped in A / O).  All fp32: the plain entry.  Some bf16: the `_bf16` twin when it
    exists and takes this combination natively; otherwise the bf16 tensors are converted to fp32 scratch tensors around the plain entry
    (inputs before, outputs after -- the stored values are the same, only passes are added).  may_decline: like call_supported."""
    acts = [a for a in args if isinstance(a, A) and a.t is not None]
    plain = lambda: [(ptr(a.t) if isinstance(a, A) else a) for a in args]
    if not any(a.t.dtype == torch.bfloat16 for a in acts):
        if may_decline:
            return call_supported(name, *plain())
        call(name, *plain())
        return True
    twin = name + '_bf16'
    pattern = tuple(1 if a.t.dtype == torch.bfloat16 else 0 for a in acts)
    if twin in nat.SIGNATURES and not BF16_FORCE_BRIDGE:
        if name in nat.BF16_MASKED_TWINS:
            mask, k = 0, 0
            for a in args:
                if isinstance(a, A):
                    if a.t is not None and a.t.dtype == torch.bfloat16:
                        mask |= 1 << k
                    k += 1
            if call_supported(twin, *(plain() + [mask])):
                return True
        elif pattern == _TWIN_PATTERN.get(name, (1,) * len(acts)):
            if call_supported(twin, *plain()):
                return True
    # bridge
    st = stream()
    conv, outs, keep = [], [], []          # keep: the fp32 scratch tensors stay referenced until the entry has been queued
    for a in args:
        if isinstance(a, A) and a.t is not None and a.t.dtype == torch.bfloat16:
            t32 = torch.empty(a.t.shape, dtype=torch.float32, device=a.t.device)
            keep.append(t32)
            if a.out:
                outs.append((a.t, t32))
            else:
                call('da_cast_bf16_to_f32', ptr(a.t), ptr(t32), a.t.numel(), st)
            conv.append(ptr(t32))
        elif isinstance(a, A):
            conv.append(ptr(a.t))
        else:
            conv.append(a)
    bridged_calls[name] = bridged_calls.get(name, 0) + 1
    if may_decline:
        if not call_supported(name, *conv):
            return False
    else:
        call(name, *conv)
    for t, t32 in outs:
        call('da_cast_f32_to_bf16', ptr(t32), ptr(t), t.numel(), st)
    del keep
    return True


def _ws(nbytes, like):
    return workspace.get(nbytes, like.device)


def _labels(t):
    """Index target -> (tensor kept alive, label_bytes).  uint8 and int64 are consumed in place."""
    nat.require_cuda(t)
    if t.dtype == torch.uint8:
        return t.contiguous(), 1
    if t.dtype != torch.int64:
        t = t.long()
    return t.contiguous(), 8


# ------------------------------------------------------------------------------------------------
# weight gradients on a side stream
# ------------------------------------------------------------------------------------------------
# The 3x3x3 weight-gradient kernels are MFMA-bound and off the critical path of the backward pass: nothing downstream needs dW
# before the optimiser step.  With ASYNC_WGRAD they run on a second HIP stream and ACCUMULATE straight into `weight.grad` (FlatAdam's
# flat bucket), so the HBM-bound BatchNorm / activation / pooling backward kernels of the following layers overlap with them
# instead of queueing behind them.  The consumer of the gradients (FlatAdam.zero_grad / step, parallel.allreduce_gradients) joins
# the side stream first.  Off by default: plain autograd semantics (p.grad valid on the current stream right after backward()).
ASYNC_WGRAD = False
_side_stream = None
_side_keep = []          # tensors the side stream still reads: referenced until the join instead of Tensor.record_stream(), whose
                         # event-polled frees made the caching allocator fall back to hipMalloc on random steps (40 -> 64 ms)


MATRIX_MODES = ('fp32', 'bf16', 'fp32_split')
# What the training entry points (SegmentationExperiment / train_seg.py, JointExperiment, bench.py) switch to unless told otherwise.  The C
# library itself starts in 'fp32' (mode 0): the kernel-variant bit-identity tests are written against the fp32 matrix instructions.
DEFAULT_MATRIX_PRECISION = 'fp32_split'


def set_matrix_precision(mode):
    """Arithmetic of the 3x3x3 convolutions on the matrix cores (process-wide; returns the previous mode):
    'fp32'        fp32 operands on the fp32 matrix instructions (one fmaf per product, like the reference's CPU convolution);
    'fp32_split'  fp32 operands scaled by a per-tile power of two and split into two fp16 terms (22 significand bits), three partial products
                  per multiply on the fp16 matrix pipe with fp32 accumulation.  Per product NARROWER than fp32 (bound 2^-21 + 2^-22 = 7e-7 against
                  fp32's 6e-8; an element more than 2^15..2^18 below its staged tile's maximum keeps an absolute 2^-40 of that maximum);
                  over the K >= 216 sums of these layers the error against double is not larger than 'fp32's (tests/test_gpu_split.py,
                  incl. the structured-outlier cases; csrc/split_f16.h states the bounds).  3/16 of the matrix time;
    'bf16'        operands ROUNDED to bf16 (BASELINE config 5); everything else stays fp32."""
    if mode not in MATRIX_MODES:
        raise ValueError("matrix precision must be one of %r, got %r" % (MATRIX_MODES, mode))
    from ._native import lib
    global _matrix_mode
    prev = lib().da_set_matrix_mode(MATRIX_MODES.index(mode))
    _matrix_mode = mode
    return MATRIX_MODES[prev]


# what the library was last told (it starts in 'fp32' unless DA_MATRIX_MODE says otherwise); only the kept-pack logic below reads it
_matrix_mode = {'1': 'bf16', '2': 'fp32_split'}.get(os.environ.get('DA_MATRIX_MODE', '0'), 'fp32')


# Every kernel of the package reduces in a fixed order (per-block partials + a second launch) except ONE: the scatter of the trilinear
# warp's gradient with respect to its SOURCE (only the joint step's 32-channel probability warp needs it), which uses float atomics.
# DETERMINISTIC switches that one to an order-independent fixed-point accumulation, making whole training steps run-to-run bit-identical.
DETERMINISTIC = os.environ.get('DA_DETERMINISTIC') == '1'
CHECK_LABELS = os.environ.get('DA_CHECK_LABELS') == '1'      # validate index targets of the cross-entropy family like torch does (host sync per call)


_nbt_pending = {}          # id(tensor) -> [tensor, count]: an alias bumped twice is ONE entry (a multi-tensor add over duplicates could race)
_NBT_FLUSH_AT = 256        # blocks used on their own, with another optimiser, never reach a flush hook: bound the list


def bump_batches_tracked(bn):
    """`bn.num_batches_tracked += 1` (nn.BatchNorm3d in training mode), deferred: one 4-us launch per BatchNorm layer in the dependent chain
    of a forward pass (17 per UNet_light step) becomes ONE multi-tensor launch when the network's forward returns (a forward hook the
    network classes register), at the next FlatAdam.step(), before any state_dict() of a module that owns a bumped counter (a state_dict
    pre-hook registered here), or when 256 distinct counters are pending."""
    t = bn.num_batches_tracked
    e = _nbt_pending.get(id(t))
    if e is None:
        _nbt_pending[id(t)] = [t, 1]
        if not getattr(bn, '_da_nbt_hooked', False):
            bn._da_nbt_hooked = True
            bn.register_state_dict_pre_hook(_nbt_state_dict_pre_hook)
        if len(_nbt_pending) >= _NBT_FLUSH_AT:
            flush_batches_tracked()
    else:
        e[1] += 1


def _nbt_state_dict_pre_hook(module, prefix, keep_vars):      # (a module-level function: a lambda here made every trained BatchNorm module unpicklable)
    flush_batches_tracked()


def flush_batches_tracked(*_):
    if _nbt_pending:
        ones = [e[0] for e in _nbt_pending.values() if e[1] == 1]
        if ones:
            torch._foreach_add_(ones, 1)
        for t, n in _nbt_pending.values():
            if n != 1:
                t.add_(n)
        _nbt_pending.clear()


def init_into(param, init_fn):
    """Fill a parameter with `init_fn` (an nn.init.*_ function) in the element order of a CONTIGUOUS tensor of its shape.  Tagged convolution
    weights are strided views of FlatAdam's tap-major buckets once an optimiser exists; a random fill of the view itself would walk
    memory order and hand the same random stream to different elements than the reference's contiguous parameter gets
    (lib/network_factory/unets.py:61-67): same seed, different weights.  Drawing into a contiguous temporary and copying keeps
    seed-for-seed reproducibility against the reference and against DA_NO_NATIVE_TIO=1."""
    with torch.no_grad():
        if param.data.is_contiguous():
            init_fn(param.data)
        else:
            tmp = torch.empty(param.shape, dtype=param.dtype, device=param.device)
            init_fn(tmp)
            param.data.copy_(tmp)


def set_deterministic(flag=True):
    global DETERMINISTIC
    prev, DETERMINISTIC = DETERMINISTIC, bool(flag)
    return prev


def enable_async_wgrad(flag=True):
    global ASYNC_WGRAD
    ASYNC_WGRAD = bool(flag)


LAST_WGRAD_ON_MAIN = os.environ.get('DA_LAST_WGRAD_ON_MAIN', '1') == '1'      # see Conv3dFn.backward
LAST_WGRAD_ON_MAIN_BN = os.environ.get('DA_LAST_WGRAD_ON_MAIN_BN', '0') == '1'
_N_SIDE = max(1, int(os.environ.get('DA_SIDE_STREAMS', '1')))      # > 1: weight gradients alternate between that many side streams (experiment)
_SIDE_PRIO = int(os.environ.get('DA_SIDE_PRIO', '0'))      # HIP stream priority of the side stream (lower number = higher priority; out-of-range values clamp)
_side_streams = []
_side_rr = 0


def side_stream():
    """The second HIP stream (round-robin over DA_SIDE_STREAMS of them); every caller is about to queue work on it."""
    global _side_stream, _side_dirty, _side_rr
    if _side_stream is None:
        _side_stream = torch.cuda.Stream(priority=_SIDE_PRIO)
        _side_streams.append(_side_stream)
        for _ in range(_N_SIDE - 1):
            _side_streams.append(torch.cuda.Stream(priority=_SIDE_PRIO))
    _side_dirty = True
    _side_rr = (_side_rr + 1) % len(_side_streams)
    return _side_streams[_side_rr]


_side_dirty = False      # work has been queued on the side stream since the last join


def join_side_stream():
    """Make the current stream wait for every weight gradient issued on the side stream.  Nothing outstanding -> no stream
    dependency at all (inside a HIP-graph capture a wait on work from before the capture would be illegal)."""
    global _side_dirty
    if _side_stream is not None and (_side_dirty or _side_keep):
        for s_ in _side_streams:
            torch.cuda.current_stream().wait_stream(s_)
    _side_keep.clear()
    _side_dirty = False


_last_side_flops = 0.0          # FLOPs of the weight gradient most recently queued on the side stream


def _serialize_matrix_kernels(flops, voxels):
    """Called right before a matrix-bound data-gradient kernel is queued on the main stream: for a LARGE full-resolution data
    gradient (>= 3e11 FLOPs on >= 4e6 voxels) behind a much shorter queued weight gradient (<= 0.4 of its FLOPs) the main stream first
    waits for the side stream.  The conv kernels are persistent with a static tile partition; such a data gradient that starts
    while the short weight gradient drains its last workgroups gets its own workgroups started unevenly and finishes late (the
    48 <- 16 data gradient of UNet_light: 3.3 ms alone, 4.9 ms when it starts 0.7 ms before the 16 -> 16 weight gradient ends).
    Whether that happens depends on microseconds of launch timing -- removing sixteen 2-us fill kernels per step moved the step from
    38.7 to 40.9 ms -- so the schedule is pinned instead of left to chance: 39.05 ms in either case.  The rule is deliberately
    narrow: on coarse, few-tile layers (all of the full UNet below full resolution) overlapping MFMA kernels fill each other's
    tails and waiting costs 10 % (268 vs 300 ms per step), and equally long kernels are left to overlap too."""
    if (ASYNC_WGRAD and _side_stream is not None and _side_dirty and flops >= _SERIALIZE_MIN_FLOPS and voxels >= _SERIALIZE_MIN_VOXELS
            and _last_side_flops <= _SERIALIZE_MAX_RATIO * flops):
        for s_ in _side_streams:
            torch.cuda.current_stream().wait_stream(s_)


_SERIALIZE_MIN_FLOPS = float(os.environ.get('DA_MFMA_SERIALIZE_MIN_FLOPS', '3e11'))      # 'inf' disables the rule
_SERIALIZE_MAX_RATIO = float(os.environ.get('DA_MFMA_SERIALIZE_MAX_RATIO', '0.4'))
_SERIALIZE_MIN_VOXELS = float(os.environ.get('DA_MFMA_SERIALIZE_MIN_VOXELS', '4e6'))


def _run_on_side(fn, keep_alive):
    """Run `fn()` on the side stream after everything queued so far on the current stream; `keep_alive` tensors stay referenced
    until the join."""
    side = side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    _side_keep.extend(t for t in keep_alive if t is not None)


_flat_buckets = []       # weak references to every FlatAdam gradient bucket alive in this process


def register_flat_bucket(flat_g):
    """FlatAdam announces its gradient bucket: only gradients that live inside such a bucket are accumulated asynchronously (their
    consumer -- FlatAdam.zero_grad / step, parallel.allreduce_gradients -- joins the side stream; any other optimiser would not)."""
    import weakref
    _flat_buckets[:] = [r for r in _flat_buckets if r() is not None]
    _flat_buckets.append(weakref.ref(flat_g))


def _in_flat_bucket(g):
    a = g.data_ptr()
    for r in _flat_buckets:
        b = r()
        if b is not None and b.device == g.device and b.data_ptr() <= a and a + 4 * g.numel() <= b.data_ptr() + 4 * b.numel():
            return True
    return False


def _async_target(param):
    """The tensor to accumulate an asynchronous gradient into, or None for the synchronous autograd path (frozen parameters,
    gradients that are not views of a registered FlatAdam bucket, ASYNC_WGRAD off)."""
    if not ASYNC_WGRAD or param is None or not isinstance(param, torch.nn.Parameter) or not param.requires_grad:
        return None
    g = param.grad
    if g is None or not g.is_cuda or g.dtype != torch.float32:
        return None
    if not g.is_contiguous() and _tio_native(g, getattr(param, '_da_kind', None)) is None:      # (a tap-major native gradient view is dense too)
        return None
    if not _in_flat_bucket(g):
        return None
    return g


# ------------------------------------------------------------------------------------------------
# weight layouts, converted once per optimiser step
# ------------------------------------------------------------------------------------------------
# The kernels take weights tap-major ([k^3][Cin][Cout], "TIO"); the parameters keep the reference's state_dict layouts.  A weight only
# changes when an optimiser steps (or a state_dict is loaded), so its TIO copy is cached ON the parameter object and rebuilt when
#   * FlatAdam stepped (its HIP kernel writes through raw pointers, so torch's version counters do not move: bump_weights_epoch()),
#   * the parameter or a registered flat parameter bucket was written in place (torch's _version), or its storage moved.
# Inside a HIP-graph capture the cache is bypassed (the conversion kernel must be part of the graph).
_weights_epoch = 0
_flat_param_buckets = []          # weak references to every FlatAdam parameter bucket


_other_bumps = 0                  # bumps that are not one optimiser's step (parameters moved into a bucket, broadcast, graph replay): everything is stale
_bucket_epoch = {}                # storage pointer of a FlatAdam parameter bucket -> number of its optimiser's steps


def bump_weights_epoch(flat_p=None):
    """Weights changed behind torch's version counters.  flat_p: only the parameters of that FlatAdam bucket (its step); None: anything."""
    global _weights_epoch, _other_bumps
    _weights_epoch += 1
    if flat_p is None:
        _other_bumps += 1
    else:
        k = flat_p.untyped_storage().data_ptr()
        _bucket_epoch[k] = _bucket_epoch.get(k, 0) + 1


def register_flat_params(flat_p):
    import weakref
    _flat_param_buckets[:] = [r for r in _flat_param_buckets if r() is not None]
    _flat_param_buckets.append(weakref.ref(flat_p))


# ------------------------------------------------------------------------------------------------
# packed convolution operands, kept across calls (include/deepatlas_hip.h: da_conv3d_k3_prepack / _use_prepacked)
# ------------------------------------------------------------------------------------------------
# In the split matrix mode every stride-1 3x3x3 forward / data-gradient call used to launch a 10-us pack kernel (weights into fragment order + tile
# table) in front of its matrix kernel: 27 launches in the dependent chain of a seg step.  The packs only change when the weights do, so they are
# kept per (weights, direction, shape): filled in-chain the first time, and from then on re-filled for ALL layers of an optimiser right after its
# step -- on the side stream, beside the start of the next forward pass -- with one event per entry that the consuming call waits for.
# An entry is valid while its stamp (weights epoch, version counters: the one of weight_tio) matches and the weights' base tensor is alive; it holds
# a weak reference to that base only.  Not inside a HIP-graph capture.  DA_NO_PACK_CACHE=1 switches it off.
PACK_CACHE = os.environ.get('DA_NO_PACK_CACHE') != '1'
_pack_entries = {}


class _Pack(object):
    __slots__ = ('bufs', 'stamp', 'event', 'base', 'view', 'args', 'disabled', 'storage', 'pversion', 'used', 'any', 'tag')


def _pack_stamp(e):
    # pversion: the version counter of the PARAMETER the weights belong to (a Parameter whose .data is a view of the flat bucket keeps its own
    # counter), as seen at the last use; FlatAdam's kernel moves neither counter, hence the epochs -- the one of the entry's OWN bucket, so that the
    # other network's optimiser step (joint training) leaves it valid
    b = e.base()
    return (_other_bumps, _bucket_epoch.get(e.storage, 0), e.pversion, b._version if b is not None else -1)


def _pack_any(w_tio, C1, C2, Cout, dgrad, stride, up2, N, D, H, W, buf, st):
    """da_conv3d_k3_prepack_any: (bytes such a pack needs, tag, filled).  buf None: size query."""
    import ctypes
    need, tag, filled = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_int(0)
    wsb = nat.lib().da_upconv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout) if up2 else nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, C1 + C2, Cout, stride)
    wp, wn = _ws(wsb, w_tio)
    call('da_conv3d_k3_prepack_any', ptr(w_tio), C1, C2, Cout, dgrad, stride, 1 if up2 else 0, N, D, H, W, ptr(buf), buf.numel() if buf is not None else 0,
         ctypes.byref(need), ctypes.byref(tag), ctypes.byref(filled), wp, wn, st)
    return need.value, tag.value, filled.value


def _pack_fill(e, w_tio, st):
    import ctypes
    if e.any is not None:                                # the families outside the split matrix kernels: one region, filled by the family's own pack kernel
        C1, C2, Cout, dgrad, N, D, H, W = e.args
        stride, up2 = e.any
        return _pack_any(w_tio, C1, C2, Cout, dgrad, stride, up2, N, D, H, W, e.bufs[0], st)[2]
    C1, C2, Cout, dgrad, N, D, H, W = e.args
    used = ctypes.c_int(0)
    b1 = e.bufs[1]
    call('da_conv3d_k3_prepack', ptr(w_tio), C1, C2, Cout, dgrad, N, D, H, W, ptr(e.bufs[0]), e.bufs[0].numel(),
         ptr(b1), b1.numel() if b1 is not None else 0, ctypes.byref(used), st)
    return used.value


# Kept packs ALSO for the folded up-sampling / native stride-2 / flow / thin kernels (da_conv3d_k3_prepack_any).  OFF by default: measured on the registration step
# (22 such packs of 5 - 20 us) it is 0.07 - 0.09 ms SLOWER than packing per call (4.25 vs 4.33 ms; backward packs only: 4.32) -- a kept pack costs the host a dictionary
# lookup, a stamp, an event wait and a hand-over call per layer plus a re-fill call per layer behind the optimiser step, more than the 3-us launch it removes, and the
# reg step's forward runs 10 - 20 us kernels that leave the host no slack.  DA_PACK_CACHE_ANY=1 switches it on (tests/test_gpu_nets.py exercises it).
_PACK_ANY_BWD_ONLY = os.environ.get('DA_PACK_CACHE_ANY_BWD_ONLY') == '1'
PACK_CACHE_ANY = os.environ.get('DA_PACK_CACHE_ANY') == '1'


def use_pack(w_tio, dgrad, C1, C2, Cout, N, D, H, W, stride=1, up2=False):
    """Right before a da_conv3d_k3_{fwd,fwd_bnstats,fwd_pro,dgrad} or da_upconv3d_k3_{fwd,dgrad} call (up2: D, H, W = the coarse extents): hand it the kept
    packed operand of these weights (filling it now if it is new or stale).  No-op outside the split matrix mode, for weights that are not views of a live base
    tensor, and inside graph capture."""
    if not PACK_CACHE or _matrix_mode != 'fp32_split' or w_tio.dtype != torch.float32 or torch.cuda.is_current_stream_capturing():
        return
    key = (w_tio.data_ptr(), dgrad, C1, C2, Cout, N, D, H, W, stride, bool(up2))
    e = _pack_entries.get(key)
    sp = w_tio.untyped_storage().data_ptr()
    if e is not None and (e.base() is None or e.storage != sp):
        e = None                                        # (the address was reused by other weights)
    if e is None:
        # only weights that live in a registered FlatAdam parameter bucket (tap-major views of it, weight_tio): the bucket is the live base object
        base = next((r for r in _flat_param_buckets if r() is not None and r().untyped_storage().data_ptr() == sp), None)
        if base is None:
            return
        nbytes = nat.lib().da_conv3d_k3_pack_bytes(N, D, H, W, C1 + C2, Cout) if (stride == 1 and not up2) else 0
        e = _Pack()
        e.any, e.tag = None, 0
        if nbytes == 0 and PACK_CACHE_ANY and (dgrad or not _PACK_ANY_BWD_ONLY):                # not a split matrix-kernel layer: one of the other families may keep its pack
            nbytes, e.tag, _ = _pack_any(w_tio, C1, C2, Cout, dgrad, stride, up2, N, D, H, W, None, stream())
            if nbytes:
                e.any = (stride, bool(up2))
        e.args, e.event, e.stamp, e.disabled, e.used = (C1, C2, Cout, dgrad, N, D, H, W), None, None, nbytes == 0, True
        e.base = base
        e.storage = sp
        e.view = (tuple(w_tio.shape), tuple(w_tio.stride()), w_tio.storage_offset())
        e.bufs = [None, None]
        if not e.disabled:
            e.bufs[0] = torch.empty((nbytes,), dtype=torch.uint8, device=w_tio.device)
            if dgrad and C2 > 0 and e.any is None:
                e.bufs[1] = torch.empty((nbytes,), dtype=torch.uint8, device=w_tio.device)
        _pack_entries[key] = e
    if e.disabled:
        return
    e.pversion = w_tio._version
    e.used = True
    stamp = _pack_stamp(e)
    if e.stamp != stamp:
        if e.event is not None:                         # a side-stream re-fill of these buffers may still be in flight (weights changed again right after the step)
            torch.cuda.current_stream().wait_event(e.event)
        if _pack_fill(e, w_tio, stream()) == 0:
            e.disabled = True
            e.bufs = [None, None]
            return
        e.stamp, e.event = stamp, None
    elif e.event is not None:
        torch.cuda.current_stream().wait_event(e.event)
    if e.any is not None:
        nat.lib().da_conv3d_k3_use_prepacked_any(ptr(w_tio), ptr(e.bufs[0]), e.bufs[0].numel(), e.tag)
        return
    b1 = e.bufs[1]
    nat.lib().da_conv3d_k3_use_prepacked(ptr(w_tio), ptr(e.bufs[0]), e.bufs[0].numel(), ptr(b1), b1.numel() if b1 is not None else 0)


def repack_after_step(flat_p):
    """FlatAdam.step() calls this behind its update kernel: every kept pack of weights that live in `flat_p` is re-filled on the side stream."""
    if not PACK_CACHE or not _pack_entries or _matrix_mode != 'fp32_split' or torch.cuda.is_current_stream_capturing():
        return
    sp = flat_p.untyped_storage().data_ptr()
    # only the packs a call asked for since the previous step of this bucket are re-filled (a validation size, random crops or sliding windows would
    # otherwise grow the per-step work with every shape ever seen); the others go stale (re-filled in their next call) and, past _PACK_KEEP shapes
    # idle for a step, are dropped
    all_mine = [(k, e) for k, e in _pack_entries.items() if not e.disabled and e.storage == sp and e.base() is not None]
    mine, idle = [], []
    for k, e in all_mine:
        if e.used:
            mine.append(e)
            e.used = False
        else:
            idle.append(k)
            e.stamp = None                              # stale: its next call re-fills it in the call's own chain
    dead = [k for k, e in _pack_entries.items() if e.base() is None] + (idle[:len(idle) - _PACK_KEEP] if len(idle) > _PACK_KEEP else [])
    for k in dead:
        _pack_entries.pop(k, None)
    if not mine:
        return
    import ctypes
    others = [e for e in mine if e.any is not None]
    mine = [e for e in mine if e.any is None]
    if others:                                          # the other families: one small pack launch each, on the side stream like the batched ones
        side = side_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sst = stream()
            for e in others:
                v = torch.as_strided(e.base(), e.view[0], e.view[1], e.view[2])
                if _pack_fill(e, v, sst) == 0:
                    e.disabled = True
            ev = torch.cuda.Event()
            ev.record(side)
        for e in others:
            if not e.disabled:
                e.stamp, e.event = _pack_stamp(e), ev
    if not mine:
        return
    n = len(mine)
    views = [torch.as_strided(e.base(), e.view[0], e.view[1], e.view[2]) for e in mine]
    PA, IA, SA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_size_t * n
    cols = list(zip(*[e.args for e in mine]))                   # C1, C2, Cout, dgrad, N, D, H, W
    used = IA()
    side = side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call('da_conv3d_k3_prepack_many', n, PA(*[v.data_ptr() for v in views]), IA(*cols[0]), IA(*cols[1]), IA(*cols[2]), IA(*cols[3]),
             IA(*cols[4]), IA(*cols[5]), IA(*cols[6]), IA(*cols[7]),
             PA(*[e.bufs[0].data_ptr() for e in mine]), SA(*[e.bufs[0].numel() for e in mine]),
             PA(*[(e.bufs[1].data_ptr() if e.bufs[1] is not None else None) for e in mine]), SA(*[(e.bufs[1].numel() if e.bufs[1] is not None else 0) for e in mine]),
             used, stream())
        ev = torch.cuda.Event()
        ev.record(side)
    for e, v, u in zip(mine, views, used):
        if u == 0:
            e.disabled = True
            continue
        e.stamp, e.event = _pack_stamp(e), ev


_PACK_KEEP = 64          # kept packs idle for a whole optimiser step beyond this many are dropped (oldest first: dicts keep insertion order)


def clear_pack_cache():
    _pack_entries.clear()


_TIO_ENTRY = {'oik': 'da_w_oik_to_tio', 'iok': 'da_w_iok_to_tio', 'iok_flip': 'da_w_iok_flip_to_tio'}
_TIO_BACK = {'oik': 'da_w_tio_to_oik', 'iok': 'da_w_tio_to_iok', 'iok_flip': 'da_w_tio_to_iok_flip'}

# TAP-MAJOR NATIVE STORAGE.  FlatAdam lays the convolution weights (and their gradients / moments) out in its flat buckets tap-major --
# the kernels' [k^3][Cin][Cout] -- and hands the modules STRIDED VIEWS of the reference's shapes ([Cout][Cin][k,k,k] / [Cin][Cout][k,k,k]):
# state_dict keys, shapes and values are the reference's, Adam is elementwise (layout-agnostic), and the per-step layout conversions
# (one launch per weight each way) do not exist.  A parameter's kind comes from tag_conv_layouts(); a parameter without an optimiser
# (or with another one) stays contiguous in the reference layout and takes the conversion kernels + cache below.
NATIVE_TIO = os.environ.get('DA_NO_NATIVE_TIO') != '1'


def tag_conv_layouts(module):
    """Mark the conv weights of `module` with the layout kind the kernels read them in ('oik': nn.Conv3d, 'iok': nn.ConvTranspose3d with
    kernel 2; the k3 transposed convs of the full UNet are read tap-FLIPPED, which no view expresses: untagged).  Called by the networks'
    constructors; FlatAdam reads the tag."""
    for m in module.modules():
        w = getattr(m, 'weight', None)
        if isinstance(m, torch.nn.ConvTranspose3d):
            if w is not None and tuple(w.shape[2:]) == (2, 2, 2):
                w._da_kind = 'iok'
        elif isinstance(m, torch.nn.Conv3d) and w is not None:
            w._da_kind = 'oik'


def param_view(buf, off, p):
    """View of buf[off : off + p.numel()] with p's logical shape: tap-major storage for tagged conv weights, plain otherwise."""
    k = p.numel()
    flat = buf[off:off + k]
    kind = getattr(p, '_da_kind', None) if NATIVE_TIO else None
    if kind is not None and p.dim() == 5:
        s0, s1, a, b, c = (int(v) for v in p.shape)
        if kind == 'oik':
            return flat.view(a, b, c, s1, s0).permute(4, 3, 0, 1, 2)
        if kind == 'iok':
            return flat.view(a, b, c, s0, s1).permute(3, 4, 0, 1, 2)
    return flat.view(p.shape)


def _tio_native(t, kind):
    """[K3][Cin][Cout] VIEW of a parameter-layout tensor whose storage already is tap-major, else None."""
    if t is None or t.dim() != 5 or kind not in ('oik', 'iok'):
        return None
    v = t.permute(2, 3, 4, 1, 0) if kind == 'oik' else t.permute(2, 3, 4, 0, 1)
    if not v.is_contiguous():
        return None
    return v.reshape(v.shape[0] * v.shape[1] * v.shape[2], v.shape[3], v.shape[4])


def _from_tio_view(dw_tio, kind, shape):
    """Parameter-layout strided VIEW of a [K3][Cin][Cout] tensor (no kernel)."""
    s0, s1, a, b, c = (int(v) for v in shape)
    if kind == 'oik':
        return dw_tio.view(a, b, c, s1, s0).permute(4, 3, 0, 1, 2)
    return dw_tio.view(a, b, c, s0, s1).permute(3, 4, 0, 1, 2)


class WgradTarget(object):
    """Where an ASYNCHRONOUS weight-gradient kernel of a parameter writes.  `gw` is the parameter's gradient tensor in the flat bucket (None:
    take the synchronous autograd path).  out() -- called INSIDE the stream context the kernel runs in, so that a scratch tensor belongs to
    that stream's allocator pool -- returns the [K3][Cin][Cout] tensor to hand to the kernel; finish() folds it into the bucket:
      * tap-major native bucket, not written since FlatAdam.zero_grad(): the bucket slice itself (writing over zeros == accumulating),
      * native, already written this step: scratch, then one add into the slice,
      * reference-layout gradient: scratch, then the converting accumulate kernel."""
    __slots__ = ('gw', 'kind', 'like', 'direct', 'view', 'tmp')

    def __init__(self, param, kind, w_tio):
        self.gw = _async_target(param)
        self.kind, self.like, self.direct, self.view, self.tmp = kind, w_tio, False, None, None
        if self.gw is not None:
            self.view = _tio_native(self.gw, kind)
            if self.view is not None and getattr(param, '_da_gz', False) and os.environ.get('DA_NO_DIRECT_WGRAD') != '1':
                param._da_gz = False
                self.direct = True

    def out(self):
        if self.direct:
            return self.view
        self.tmp = torch.empty_like(self.like)
        return self.tmp

    def finish(self):
        if self.direct:
            return
        if self.view is not None:
            self.view.add_(self.tmp)
        else:
            grad_from_tio(self.tmp, self.kind, self.gw.shape, acc=self.gw)
        self.tmp = None


_NO_WGRAD_TARGET = type('NoTarget', (), {'gw': None})()


def grad_for_autograd(dw_tio, kind, param):
    """The gradient tensor handed to autograd for `param` from its [K3][Cin][Cout] gradient: a strided view when the parameter lives
    tap-major (no kernel), else the layout conversion."""
    if isinstance(param, torch.nn.Parameter):
        param._da_gz = False                        # autograd is about to accumulate into .grad
    if kind in ('oik', 'iok') and _tio_native(param.detach(), kind) is not None:
        return _from_tio_view(dw_tio, kind, param.shape)
    return grad_from_tio(dw_tio, kind, param.shape)


def weight_tio(weight, kind):
    """[K3][Cin][Cout] copy of a conv weight.  kind 'oik': nn.Conv3d [Cout][Cin][k^3]; 'iok': nn.ConvTranspose3d [Cin][Cout][k^3];
    'iok_flip': a k3/s1/p1 transposed conv used as a convolution (taps flipped)."""
    a_, b_ = int(weight.shape[0]), int(weight.shape[1])
    K3 = int(weight.shape[2] * weight.shape[3] * weight.shape[4])
    Cin, Cout = (b_, a_) if kind == 'oik' else (a_, b_)
    nv = _tio_native(weight.detach(), kind)
    if nv is not None:                         # tap-major native storage (FlatAdam): the kernels read the parameter's own memory
        return nv
    capturing = torch.cuda.is_current_stream_capturing()
    stamp = None
    if not capturing and os.environ.get('DA_NO_WEIGHT_CACHE') != '1':
        stamp = (_weights_epoch, weight._version, weight.data_ptr(), kind,
                 tuple(r()._version for r in _flat_param_buckets if r() is not None))
        ent = getattr(weight, '_da_tio', None)
        if ent is not None and ent[0] == stamp:
            return ent[1]
    w_tio = torch.empty((K3, Cin, Cout), dtype=torch.float32, device=weight.device)
    call(_TIO_ENTRY[kind], ptr(weight.detach().contiguous()), ptr(w_tio), a_, b_, K3, stream())
    if stamp is not None:
        try:
            weight._da_tio = (stamp, w_tio)
        except AttributeError:
            pass
    return w_tio


def grad_from_tio(dw_tio, kind, shape, acc=None):
    """The gradient in the parameter's own layout from a [K3][Cin][Cout] gradient: a new tensor, or (acc given) accumulated into
    `acc` by the same kernel."""
    a_, b_ = int(shape[0]), int(shape[1])
    K3 = int(dw_tio.shape[0])
    if acc is not None:
        call(_TIO_BACK[kind] + '_acc', ptr(dw_tio), ptr(acc), a_, b_, K3, stream())
        return None
    dw = torch.empty(tuple(shape), dtype=torch.float32, device=dw_tio.device)
    call(_TIO_BACK[kind], ptr(dw_tio), ptr(dw), a_, b_, K3, stream())
    return dw


# ------------------------------------------------------------------------------------------------
# deferred BatchNorm + activation
# ------------------------------------------------------------------------------------------------
# In Conv -> BatchNorm -> LeakyReLU -> Conv chains (unets.py:24-39, 259-278) the activated tensor only exists to be read by the next
# 3x3x3 convolution.  With LAZY_BN the producer returns its RAW output together with the BatchNorm scale / shift, and the consumer's
# kernels apply `act(x * scale + shift)` while staging their input tile (da_conv3d_k3_fwd_pro / da_conv3d_k3_wgrad_pro): the apply
# pass and the activated tensor disappear, the arithmetic (and therefore every result) stays bit-identical
# (tests/test_gpu_nets.py::test_lazy_batchnorm_matches_materialised_activations).
# Measured at 160x192x160, batch 2.  In the HBM-bound head kernels the prologue is free; in the MFMA-bound 3x3x3
# kernels its 4 VALU operations per staged element are not: 16 -> 16 forward 1.17 -> 1.20 ms, weight gradient 1.22 -> 1.29 ms; 48 -> 16
# forward 3.26 -> 3.40 ms, weight gradient 3.69 -> 3.93 ms.  Conv -> conv and conv -> head links (on by default, DA_LAZY_BN=0 turns them
# off): 39.15 -> 38.6 ms/step, and the activated tensors of those links are never allocated.  The up-sampler -> concat-conv link
# (DA_LAZY_BN_UPSAMPLER=1; off by default) saves another 0.2 ms/step but moves the bench's roofline call onto the prologue variant of
# the 48 -> 16 forward (0.757 instead of 0.778 of the fp32 matrix peak for the same algorithmic FLOPs).
LAZY_BN = os.environ.get('DA_LAZY_BN', '1') != '0'
_LAZY_UP_ENV = os.environ.get('DA_LAZY_BN_UPSAMPLER')          # '1' / '0' force the up-sampler link on / off; unset: on with bf16 activation storage
LAZY_BN_UPSAMPLER = _LAZY_UP_ENV == '1'                         # (bf16 storage: the prologue is cheaper there -- seg step 13.04 -> 12.61 ms; fp32: 27.70 -> 27.70 - 27.78)


class LazyAct(object):
    """A tensor whose per-channel affine + activation has not been applied yet.  `raw` carries the autograd history; the gradient
    that flows back into it is the gradient with respect to the ACTIVATED tensor (the producer's backward expects exactly that)."""
    __slots__ = ('raw', 'scale', 'shift', 'slope')

    def __init__(self, raw, scale, shift, slope):
        self.raw, self.scale, self.shift, self.slope = raw, scale, shift, float(slope)

    @property
    def shape(self):
        return self.raw.shape

    def materialize(self):
        return ApplyAffineActFn.apply(self.raw, self.scale, self.shift, self.slope)


def materialize(x):
    return x.materialize() if isinstance(x, LazyAct) else x


class ApplyAffineActFn(Function):
    """act(raw * scale + shift) as its own pass (da_bn_act_fwd) for consumers without an input prologue.  The incoming gradient is
    already the gradient with respect to the activated tensor, which is what `raw`'s producer expects: passed through unchanged."""

    @staticmethod
    def forward(ctx, raw, scale, shift, slope):
        a = ndhwc(raw)
        C = a.shape[-1]
        out = torch.empty_like(a)
        call_act('da_bn_act_fwd', A(a), ptr(scale), ptr(shift), float(slope), O(out), a.numel() // C, C, stream())
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        return gout, None, None, None


def _pro_args(pro):
    """(scale_ptr, shift_ptr, slope) of an optional prologue for the C entry points."""
    if pro is None:
        return None, None, -1.0
    return ptr(pro[0]), ptr(pro[1]), float(pro[2])


def _apply_pro(a, pro, st):
    """Materialise act(a * scale + shift) (fallback when a shape is not taken by the prologue kernels)."""
    if pro is None:
        return a
    C = a.shape[-1]
    out = torch.empty_like(a)
    call_act('da_bn_act_fwd', A(a), ptr(pro[0]), ptr(pro[1]), float(pro[2]), O(out), a.numel() // C, C, st)
    return out


# ------------------------------------------------------------------------------------------------
# convolutions
# ------------------------------------------------------------------------------------------------
class Conv3dK3Fn(Function):
    """nn.Conv3d(k=3, p=1, stride 1|2) on concat(x1, x2) (+ optional fused ReLU/LeakyReLU).
    Reference call sites: unets.py:30,36; modules.py:48,56-58; voxel_morph.py:57,82."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, stride, act_slope, *extra):
        # extra[0] (optional): `weight` is a ConvTranspose3d(k=3, s=1, p=1) weight [Cin][Cout][3,3,3] (unets.py:88-96): the same
        # operation as this convolution with flipped taps, so only the weight re-layout kernels differ
        # extra[1] (optional): fork -- the output is returned TWICE (two aliases) for a tensor with two consumers (skip connection):
        # the two gradients then arrive separately and are summed inside the activation-backward pass instead of by autograd
        transposed = bool(extra[0]) if extra else False
        fork = bool(extra[1]) if len(extra) > 1 else False
        # extra[2] (optional): up2 -- x1 / x2 are the COARSE tensors of `conv(F.interpolate(x, scale 2, nearest))` (voxel_morph.py:72-80): the
        # up-sampling is folded into the convolution (conv3d_up2.hip), the up-sampled tensor and its gradient never exist
        up2 = bool(extra[2]) if len(extra) > 2 else False
        ctx.n_extra, ctx.transposed, ctx.fork, ctx.up2 = len(extra), transposed, fork, up2
        a1 = ndhwc(x1)
        a2 = ndhwc(x2) if x2 is not None else None
        N, D, H, W, C1 = a1.shape
        C2 = a2.shape[-1] if a2 is not None else 0
        Cout, Cin = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        if Cin != C1 + C2 or tuple(weight.shape[2:]) != (3, 3, 3):
            raise ValueError('weight %s does not match input channels %d+%d' % (tuple(weight.shape), C1, C2))
        if transposed and stride != 1:
            raise NotImplementedError('transposed 3x3x3 conv: stride 1 only')
        if up2 and (transposed or stride != 1 or not upconv_supported(C1, C2, Cout)):
            raise NotImplementedError('folded up-sampling: plain stride-1 convolution, channel counts of ops.upconv_supported, split matrix mode')
        st = stream()
        w_tio = weight_tio(weight, 'iok_flip' if transposed else 'oik')
        b = bias.detach().contiguous() if bias is not None else None
        if up2:
            out = _empty((N, 2 * D, 2 * H, 2 * W, Cout), a1, _act_dtype(Cout))
            wsb = nat.lib().da_upconv3d_k3_ws_bytes(N, D, H, W, Cin, Cout)
            wp, wn = _ws(wsb, a1)
            use_pack(w_tio, 0, C1, C2, Cout, N, D, H, W, 1, True)
            call_act('da_upconv3d_k3_fwd', A(a1), C1, A(a2), C2, ptr(w_tio), ptr(b), O(out), N, D, H, W, Cout, float(act_slope), wp, wn, st)
        else:
            Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
            out = _empty((N, Do, Ho, Wo, Cout), a1, _act_dtype(Cout))
            wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, stride)
            wp, wn = _ws(wsb, a1)
            use_pack(w_tio, 0, C1, C2, Cout, N, D, H, W, stride)
            call_act('da_conv3d_k3_fwd', A(a1), C1, A(a2), C2, ptr(w_tio), ptr(b), O(out),
                     N, D, H, W, Cout, stride, float(act_slope), wp, wn, st)
        ctx.dims = (N, D, H, W, C1, C2, Cout, stride, float(act_slope), wsb)
        ctx.has_bias = bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        ctx.save_for_backward(a1, a2, w_tio, out if act_slope >= 0 else None)
        if fork:
            return ncdhw(out), ncdhw(out.view(out.shape))
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout, *gmore):
        a1, a2, w_tio, out = ctx.saved_tensors
        N, D, H, W, C1, C2, Cout, stride, slope, wsb = ctx.dims
        up2 = ctx.up2
        st = stream()
        flops = 54.0 * (C1 + C2) * Cout * N * D * H * W * (8.0 if up2 else 1.0 / (stride ** 3))       # algorithmic (2 * 27 * Cin * Cout per output voxel)

        def k_dgrad(g_, dx1_, dx2_, wp_, wn_, st_):
            if up2:
                use_pack(w_tio, 1, C1, C2, Cout, N, D, H, W, 1, True)
                call_act('da_upconv3d_k3_dgrad', A(g_), ptr(w_tio), O(dx1_), C1, O(dx2_), C2, N, D, H, W, Cout, wp_, wn_, st_)
            else:
                use_pack(w_tio, 1, C1, C2, Cout, N, D, H, W, stride)
                call_act('da_conv3d_k3_dgrad', A(g_), ptr(w_tio), O(dx1_), C1, O(dx2_), C2, N, D, H, W, Cout, stride, wp_, wn_, st_)

        def k_wgrad(g_, dw_tio_, db_, wp_, wn_, st_):
            if up2:
                call_act('da_upconv3d_k3_wgrad', A(a1), C1, A(a2), C2, A(g_), ptr(dw_tio_), N, D, H, W, Cout, wp_, wn_, st_)
                if db_ is not None:
                    call_act('da_colsum', A(g_), g_.numel() // Cout, Cout, ptr(db_), wp_, wn_, st_)
            elif db_ is not None and (g_.dtype == torch.bfloat16 or a1.dtype == torch.bfloat16):          # (the bf16 twins leave the bias gradient to its own pass)
                call_act('da_conv3d_k3_wgrad', A(a1), C1, A(a2), C2, A(g_), ptr(dw_tio_), None, N, D, H, W, Cout, stride, wp_, wn_, st_)
                call_act('da_colsum', A(g_), g_.numel() // Cout, Cout, ptr(db_), wp_, wn_, st_)
            else:
                call_act('da_conv3d_k3_wgrad', A(a1), C1, A(a2), C2, A(g_), ptr(dw_tio_), ptr(db_), N, D, H, W, Cout, stride, wp_, wn_, st_)
        g_b = gmore[0] if (ctx.fork and gmore) else None
        if gout is None:
            gout, g_b = g_b, None
        g = ndhwc(gout)
        gb2 = ndhwc(g_b) if g_b is not None else None
        want_w = ctx.needs_input_grad[2]
        want_b = ctx.has_bias and ctx.needs_input_grad[3]
        db = None
        db_partial = None            # (partial buffer, number of partial sets): bias gradient still to be finished, on the side stream
        if out is not None or gb2 is not None:
            # dy = (g [+ g']) * act'(y) and the bias gradient (column sums of dy) in one pass
            g2 = torch.empty_like(g)
            M = g.numel() // Cout
            gbt0 = _async_target(ctx.bparam) if want_b else None
            if gbt0 is not None:
                # the per-channel finish of the column sums (a Cout-workgroup kernel) goes to the side stream and accumulates straight
                # into the flat gradient bucket: behind persistent matrix kernels such a kernel waits 20 - 100 us for a CU slot, and
                # nothing on the main stream needs its result
                import ctypes
                pbytes = nat.lib().da_bn_ws_bytes(M, Cout)
                pbuf = torch.empty((pbytes,), dtype=torch.uint8, device=g.device)
                npar = ctypes.c_int(0)
                call_act('da_act_bwd_add_partial', A(g), A(gb2), A(out), slope if out is not None else -1.0, O(g2), M, Cout,
                         ptr(pbuf), pbytes, ctypes.byref(npar), st)
                db_partial = (pbuf, npar.value)
            else:
                db = _empty((Cout,), a1) if want_b else None
                bwp, bwn = _ws(max(wsb, nat.lib().da_bn_ws_bytes(M, Cout)), a1)
                call_act('da_act_bwd_add_dbias', A(g), A(gb2), A(out), slope if out is not None else -1.0, O(g2), ptr(db), M, Cout, bwp, bwn, st)
            g = g2
        wp, wn = _ws(wsb, a1)
        dx1 = dx2 = None
        if ctx.needs_input_grad[0] or (a2 is not None and ctx.needs_input_grad[1]):
            dx1 = torch.empty_like(a1)
            dx2 = torch.empty_like(a2) if a2 is not None else None
            _serialize_matrix_kernels(flops, N * D * H * W)
            k_dgrad(g, dx1, dx2, wp, wn, st)
        dw = None
        need_db_in_wgrad = want_b and db is None and db_partial is None
        kind_w = 'iok_flip' if ctx.transposed else 'oik'
        wt = WgradTarget(ctx.wparam, kind_w, w_tio) if want_w else _NO_WGRAD_TARGET
        gw = wt.gw
        gbt = _async_target(ctx.bparam) if want_b else None
        if db_partial is not None and not (want_w and gw is not None):
            def finish():                            # (frozen or non-bucket weight: the bias finish still goes to the side stream)
                call('da_colsum_finish', ptr(db_partial[0]), db_partial[1], Cout, ptr(gbt), 1, stream())
            _run_on_side(finish, (db_partial[0],))
            db_partial = None
        if want_w and gw is not None and (not want_b or gbt is not None):
            global _last_side_flops
            _last_side_flops = flops
            if db is not None:
                gbt.add_(db)                         # the fused pass above already produced the bias gradient (main stream)
                db = None
            # a layer whose inputs want no gradient is the net's first: nothing follows it on the main chain, while the side stream still has the weight
            # gradients of the layers before it queued up -- its own weight gradient runs on the main stream (reg step: the side stream's backlog was the step's tail)
            on_main = LAST_WGRAD_ON_MAIN and dx1 is None and dx2 is None
            side = torch.cuda.current_stream() if on_main else side_stream()
            if not on_main:
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sst = stream()
                if db_partial is not None:
                    call('da_colsum_finish', ptr(db_partial[0]), db_partial[1], Cout, ptr(gbt), 1, sst)
                    _side_keep.append(db_partial[0])
                dbs = _empty((Cout,), a1) if need_db_in_wgrad else None
                swp, swn = _ws(max(wsb, nat.lib().da_bn_ws_bytes(g.numel() // Cout, Cout)) if up2 else wsb, a1)
                k_wgrad(g, wt.out(), dbs, swp, swn, sst)
                wt.finish()
                if dbs is not None:
                    gbt.add_(dbs)
            _side_keep.extend(t for t in (a1, a2, g) if t is not None)
        elif want_w or need_db_in_wgrad:
            dw_tio = torch.empty_like(w_tio)
            dbw = _empty((Cout,), a1) if need_db_in_wgrad else None
            if up2 and dbw is not None:
                wp, wn = _ws(max(wsb, nat.lib().da_bn_ws_bytes(g.numel() // Cout, Cout)), a1)
            k_wgrad(g, dw_tio, dbw, wp, wn, st)
            if dbw is not None:
                db = dbw
            if want_w:
                dw = grad_for_autograd(dw_tio, kind_w, ctx.wparam)
        return (ncdhw(dx1) if dx1 is not None else None, ncdhw(dx2) if dx2 is not None else None, dw, db, None, None) + (None,) * ctx.n_extra


def upconv_supported(C1, C2, Cout):
    """Can `conv3x3x3(F.interpolate(cat(x1, x2), scale 2, nearest))` run with the up-sampling folded in (conv3d_up2.hip)?  Channel counts
    as da_upconv3d_k3_supported documents; only in split matrix mode (the kernels' arithmetic)."""
    return bool(nat.lib().da_upconv3d_k3_supported(int(C1), int(C2), int(Cout))) and os.environ.get('DA_NO_UPCONV') != '1'


class Conv1x1Fn(Function):
    """nn.Conv3d(Cin, Cout, 1): the segmentation head (unets.py:249-250)."""

    @staticmethod
    def forward(ctx, x, weight, bias, *extra):
        pro = extra[0] if extra else None          # (scale, shift, slope) still to be applied to x (LazyAct input)
        ctx.n_extra = len(extra)
        a = ndhwc(x)
        N, D, H, W, Cin = a.shape
        Cout = weight.shape[0]
        st = stream()
        w_io = weight_tio(weight, 'oik').view(Cin, Cout)
        out = _empty((N, D, H, W, Cout), a)
        M = N * D * H * W
        b = bias.detach().contiguous() if bias is not None else None
        wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(1, Cin, Cout), a)
        if pro is not None and not call_act('da_conv1x1_fwd_pro', A(a), ptr(pro[0]), ptr(pro[1]), float(pro[2]), ptr(w_io), ptr(b), O(out),
                                            M, Cin, Cout, wp, wn, st, may_decline=True):
            a, pro = _apply_pro(a, pro, st), None
        if pro is None:
            call_act('da_conv1x1_fwd', A(a), ptr(w_io), ptr(b), O(out), M, Cin, Cout, wp, wn, st)
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        ctx.pro_slope = pro[2] if pro is not None else None
        ctx.save_for_backward(a, w_io, pro[0] if pro is not None else None, pro[1] if pro is not None else None)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        a, w_io, ps, pt = ctx.saved_tensors
        Cin, Cout = w_io.shape
        M = a.numel() // Cin
        st = stream()
        g = ndhwc(gout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(a)
            wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(1, Cin, Cout), a)
            call_act('da_conv1x1_dgrad', A(g), ptr(w_io), O(dx), M, Cin, Cout, wp, wn, st)
        # (the head's weight gradient stays on the main stream: it is the first backward kernel, HBM-bound like its neighbours --
        # on the side stream it only competed with them: 39.2 -> 41.5 ms)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw_io = torch.empty_like(w_io)
            db = _empty((Cout,), a) if ctx.has_bias else None
            wp, wn = _ws(nat.lib().da_conv1x1_wgrad_ws_bytes(M, Cin, Cout), a)
            if ps is not None:
                if not call_act('da_conv1x1_wgrad_pro', A(a), ptr(ps), ptr(pt), float(ctx.pro_slope), A(g), ptr(dw_io), ptr(db),
                                M, Cin, Cout, wp, wn, st, may_decline=True):
                    call_act('da_conv1x1_wgrad', A(_apply_pro(a, (ps, pt, ctx.pro_slope), st)), A(g), ptr(dw_io), ptr(db), M, Cin, Cout, wp, wn, st)
            else:
                call_act('da_conv1x1_wgrad', A(a), A(g), ptr(dw_io), ptr(db), M, Cin, Cout, wp, wn, st)
            dw = grad_for_autograd(dw_io.view(1, Cin, Cout), 'oik', ctx.wparam)
        return ((ncdhw(dx) if dx is not None else None), dw, db) + (None,) * ctx.n_extra


class DeconvK2S2Fn(Function):
    """nn.ConvTranspose3d(k=2, s=2) (unets.py:49,55)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        a = ndhwc(x)
        N, D, H, W, Cin = a.shape
        Cout = weight.shape[1]
        if weight.shape[0] != Cin or tuple(weight.shape[2:]) != (2, 2, 2):
            raise ValueError('ConvTranspose3d weight %s does not match input channels %d' % (tuple(weight.shape), Cin))
        st = stream()
        w_tio = weight_tio(weight, 'iok')
        out = _empty((N, 2 * D, 2 * H, 2 * W, Cout), a, _act_dtype(Cout))
        b = bias.detach().contiguous() if bias is not None else None
        wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), a)
        call_act('da_deconv_k2s2_fwd', A(a), ptr(w_tio), ptr(b), O(out), N, D, H, W, Cin, Cout, wp, wn, st)
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        ctx.save_for_backward(a, w_tio)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        a, w_tio = ctx.saved_tensors
        N, D, H, W, Cin = a.shape
        Cout = w_tio.shape[2]
        st = stream()
        g = ndhwc(gout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(a)
            wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), a)
            call_act('da_deconv_k2s2_dgrad', A(g), ptr(w_tio), O(dx), N, D, H, W, Cin, Cout, wp, wn, st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw_tio = torch.empty_like(w_tio)
            db = _empty((Cout,), a) if ctx.has_bias else None
            wp, wn = _ws(nat.lib().da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout), a)
            call_act('da_deconv_k2s2_wgrad', A(a), A(g), ptr(dw_tio), ptr(db), N, D, H, W, Cin, Cout, wp, wn, st)
            dw = grad_for_autograd(dw_tio, 'iok', ctx.wparam)
        return (ncdhw(dx) if dx is not None else None), dw, db


class ConvK2S2Fn(Function):
    """nn.Conv3d(kernel 2, stride 2, padding 0): UNet_generator(maxpool=False) down-sampler (unets.py:231-233).  The adjoint of
    the k2/s2 transposed conv, on the same pointwise MFMA kernels.  Even spatial sizes, channels in multiples of 16."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        a = ndhwc(x)
        N, D2, H2, W2, Cin = a.shape
        Cout = weight.shape[0]
        if weight.shape[1] != Cin or tuple(weight.shape[2:]) != (2, 2, 2):
            raise ValueError('Conv3d(k2,s2) weight %s does not match input channels %d' % (tuple(weight.shape), Cin))
        if (D2 | H2 | W2) & 1 or Cin % 16 or Cout % 16:
            raise NotImplementedError('HIP strided-conv down-sampler: even spatial sizes and channels in multiples of 16')
        D, H, W = D2 // 2, H2 // 2, W2 // 2
        st = stream()
        w_tio = weight_tio(weight, 'oik')
        out = _empty((N, D, H, W, Cout), a, _act_dtype(Cout))
        b = bias.detach().contiguous() if bias is not None else None
        wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), a)
        call_act('da_conv_k2s2_fwd', A(a), ptr(w_tio), ptr(b), O(out), N, D, H, W, Cin, Cout, wp, wn, st)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(a, w_tio)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        a, w_tio = ctx.saved_tensors
        N, D2, H2, W2, Cin = a.shape
        D, H, W = D2 // 2, H2 // 2, W2 // 2
        Cout = w_tio.shape[2]
        st = stream()
        g = ndhwc(gout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(a)
            wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cout, Cin), a)
            call_act('da_conv_k2s2_dgrad', A(g), ptr(w_tio), O(dx), N, D, H, W, Cin, Cout, wp, wn, st)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw_toi = _empty((8, Cout, Cin), a)
            db = _empty((Cout,), a) if ctx.has_bias else None
            wp, wn = _ws(nat.lib().da_conv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout), a)
            call_act('da_conv_k2s2_wgrad', A(a), A(g), ptr(dw_toi), ptr(db), N, D, H, W, Cin, Cout, wp, wn, st)
            dw = _empty((Cout, Cin, 2, 2, 2), a)
            call('da_w_tio_to_iok', ptr(dw_toi), ptr(dw), Cout, Cin, 8, st)          # [8][Cout][Cin] -> [Cout][Cin][8]
        return (ncdhw(dx) if dx is not None else None), dw, db


class UpsampleTrilinear2Fn(Function):
    """nn.Upsample(scale_factor=2, mode='trilinear') (align_corners=False): UNet_generator(upsample=True), unets.py:236."""

    @staticmethod
    def forward(ctx, x):
        a = ndhwc(x)
        N, D, H, W, C = a.shape
        out = _empty((N, 2 * D, 2 * H, 2 * W, C), a, a.dtype)
        call_act('da_upsample_trilinear2_fwd', A(a), O(out), N, D, H, W, C, stream())
        ctx.dims = (N, D, H, W, C)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        N, D, H, W, C = ctx.dims
        g = ndhwc(gout)
        dx = _empty((N, D, H, W, C), g, g.dtype)
        call_act('da_upsample_trilinear2_bwd', A(g), O(dx), N, D, H, W, C, stream())
        return ncdhw(dx)


# ------------------------------------------------------------------------------------------------
# BatchNorm3d + activation
# ------------------------------------------------------------------------------------------------
class BNActFn(Function):
    """nn.BatchNorm3d followed by LeakyReLU/ReLU (unets.py:31-32,51-52), one fused pass each way.
    running_mean / running_var are updated in place in training mode (momentum, unbiased variance)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, slope):
        a = ndhwc(y)
        C = a.shape[-1]
        M = a.numel() // C
        st = stream()
        stats = _empty((4, C), a)                     # mean, rstd, scale, shift
        g = gamma.detach().contiguous() if gamma is not None else None
        b = beta.detach().contiguous() if beta is not None else None
        wsb = nat.lib().da_bn_ws_bytes(M, C)
        if training or running_mean is None:
            wp, wn = _ws(wsb, a)
            call_act('da_bn_train_stats', A(a), M, C, ptr(g), ptr(b), float(eps), float(momentum),
                     ptr(running_mean), ptr(running_var), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), wp, wn, st)
        else:
            call('da_bn_eval_affine', ptr(g), ptr(b), ptr(running_mean), ptr(running_var), float(eps), C,
                 ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), st)
        out = torch.empty_like(a)
        call_act('da_bn_act_fwd', A(a), ptr(stats[2]), ptr(stats[3]), float(slope), O(out), M, C, st)
        ctx.cfg = (M, C, float(slope), bool(training or running_mean is None), wsb)
        ctx.save_for_backward(a, stats, g)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        a, stats, g = ctx.saved_tensors
        M, C, slope, train, wsb = ctx.cfg
        st = stream()
        go = ndhwc(gout)
        dx = torch.empty_like(a)
        dgb = _empty((2, C), a)
        wp, wn = _ws(wsb, a)
        call_act('da_bn_act_bwd', A(go), A(a), ptr(stats[0]), ptr(stats[1]), ptr(g), ptr(stats[2]), ptr(stats[3]),
                 slope, 1 if train else 0, O(dx), ptr(dgb[0]), ptr(dgb[1]), M, C, wp, wn, st)
        return ncdhw(dx), dgb[0], dgb[1], None, None, None, None, None, None


def _bn_forward(a, gamma, beta, running_mean, running_var, training, momentum, eps, slope, st, partials=None, apply=True):
    """stats (train: batch statistics + running-stat update, eval: running stats) and fused apply+activation.
    partials = (double tensor [nparts][2][C], nparts) when the producing conv already accumulated the sums in its epilogue."""
    C = a.shape[-1]
    M = a.numel() // C
    stats = _empty((4, C), a)                     # mean, rstd, scale, shift
    g = gamma.detach().contiguous() if gamma is not None else None
    b = beta.detach().contiguous() if beta is not None else None
    wsb = nat.lib().da_bn_ws_bytes(M, C)
    train = bool(training or running_mean is None)
    if train and partials is not None and partials[1] > 0:
        call('da_bn_train_stats_from_partials', ptr(partials[0]), partials[1], M, C, ptr(g), ptr(b), float(eps), float(momentum),
             ptr(running_mean), ptr(running_var), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), st)
    elif train:
        wp, wn = _ws(wsb, a)
        call_act('da_bn_train_stats', A(a), M, C, ptr(g), ptr(b), float(eps), float(momentum),
                 ptr(running_mean), ptr(running_var), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), wp, wn, st)
    else:
        call('da_bn_eval_affine', ptr(g), ptr(b), ptr(running_mean), ptr(running_var), float(eps), C,
             ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), st)
    out = None
    if apply:
        out = torch.empty_like(a)
        call_act('da_bn_act_fwd', A(a), ptr(stats[2]), ptr(stats[3]), float(slope), O(out), M, C, st)
    return out, stats, g, (M, C, float(slope), train, wsb)


# BatchNorm-backward sums that the PRODUCER of a gradient tensor accumulated in its own epilogue (da_head_dice_bwd_bst): gradient data pointer ->
# (the gradient tensor itself -- kept alive so that the pointer cannot be reused while the entry exists --, partials, count, M, C).  The consumer
# (_bn_backward of the layer whose output that gradient belongs to) pops its entry; FlatAdam.zero_grad / step drop whatever was never consumed.
_bwd_stats = {}
# one-input layers whose data gradient carries the producer's sums: <= 16 input channels.  The entry also takes 32 (two N-tiles, one per workgroup), measured
# slower in the step: 20.46 - 20.56 -> 20.59 - 20.67 ms (DA_DGRAD_BST_MAXC=32)
_DGRAD_BST_MAXC = int(os.environ.get('DA_DGRAD_BST_MAXC', '16'))
FUSE_BN_BWD_STATS = os.environ.get('DA_NO_BN_BWD_FUSE') != '1'
_DGRAD_BST_CONCAT = os.environ.get('DA_NO_DGRAD_BST_CONCAT') != '1'      # the 32 + 16 concat layer's data gradient with the producer's sums (only reached with DA_LAZY_BN_UPSAMPLER=1)


def drop_bwd_stats(*_):
    _bwd_stats.clear()


DIRECT_SMALL_GRADS = os.environ.get('DA_NO_DIRECT_SMALL_GRADS') != '1'


def _direct_small_target(small_params, C, want_dbias):
    """The (3, C) slice of a FlatAdam gradient bucket that holds (bias, gamma, beta) of one block, if the BatchNorm-backward kernels may WRITE their three
    results straight into it: all three gradients are views of a registered bucket, consecutive there (conv.bias, BN.weight, BN.bias are consecutive
    parameters), and still all zeros since zero_grad (`_da_gz`, the mark the direct weight gradients use).  Saves the (3, C) scratch tensor and one
    torch add per block (19 launches on the dependent chain of a seg step).  None: take the scratch tensor + _accumulate_small_grads."""
    if not DIRECT_SMALL_GRADS or small_params is None or not want_dbias or False:
        return None
    bias, gamma, beta = small_params
    if bias is None or gamma is None or beta is None:
        return None
    if not (getattr(bias, '_da_gz', False) and getattr(gamma, '_da_gz', False) and getattr(beta, '_da_gz', False)):
        return None
    gb, gg, gbt = _async_target(bias), _async_target(gamma), _async_target(beta)
    if gb is None or gg is None or gbt is None:
        return None
    if not (gb.numel() == gg.numel() == gbt.numel() == C and gb.is_contiguous() and gg.data_ptr() == gb.data_ptr() + 4 * C and gbt.data_ptr() == gg.data_ptr() + 4 * C):
        return None
    bias._da_gz = gamma._da_gz = beta._da_gz = False
    return torch.as_strided(gb, (3, C), (C, 1))


def _bn_backward(go, y, stats, cfg, want_dbias, st, small_params=None):
    """BN+activation backward; returns (dy, dgamma, dbeta, dbias_of_producer or None) -- the producer's bias gradient is the
    column sum of dy and is accumulated inside the apply pass.  With `small_params` = the block's (bias, gamma, beta) parameters the three
    gradients may be written straight into the optimiser's bucket (_direct_small_target); the returned gradients are then all None."""
    M, C, slope, train, wsb = cfg
    dy = torch.empty_like(y)
    direct = _direct_small_target(small_params, C, want_dbias) if train else None
    dgb = direct if direct is not None else _empty((3, C), y)                   # rows: producer bias, gamma, beta -- the order the parameters have in a block
    wp, wn = _ws(wsb, y)
    pre = _bwd_stats.pop(go.data_ptr(), None) if _bwd_stats else None
    if pre is not None and train and pre[3] == M and pre[4] == C and go.dtype == torch.float32 and y.dtype == torch.float32 and go.is_contiguous():
        call('da_bn_act_bwd_dbias_pre', ptr(go), ptr(y), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             slope, 1, ptr(dy), ptr(dgb[1]), ptr(dgb[2]), ptr(dgb[0]) if want_dbias else None, M, C, ptr(pre[1]), pre[2], wp, wn, st)
        return (dy, None, None, None) if direct is not None else (dy, dgb[1], dgb[2], (dgb[0] if want_dbias else None))
    call_act('da_bn_act_bwd_dbias', A(go), A(y), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             slope, 1 if train else 0, O(dy), ptr(dgb[1]), ptr(dgb[2]), ptr(dgb[0]) if want_dbias else None, M, C, wp, wn, st)
    return (dy, None, None, None) if direct is not None else (dy, dgb[1], dgb[2], (dgb[0] if want_dbias else None))


def _accumulate_small_grads(bias, gamma, beta, db, dgamma, dbeta):
    """With ASYNC_WGRAD (gradients may be accumulated straight into FlatAdam's bucket): if bias / gamma / beta of a block sit next to
    each other in the bucket (they do: conv.bias, BN.weight, BN.bias are consecutive parameters) their three gradients -- one (3, C)
    tensor, see _bn_backward -- are added with ONE kernel instead of three autograd accumulations.  Returns the gradients still to
    be handed to autograd (None where already accumulated)."""
    if db is None or dgamma is None or dbeta is None or os.environ.get('DA_NO_SMALL_GRAD_FUSE') == '1':
        return db, dgamma, dbeta
    gb, gg, gbt = _async_target(bias), _async_target(gamma), _async_target(beta)
    if gb is None or gg is None or gbt is None:
        return db, dgamma, dbeta
    C = db.numel()
    if not (gb.numel() == gg.numel() == gbt.numel() == C and gg.data_ptr() == gb.data_ptr() + 4 * C and gbt.data_ptr() == gg.data_ptr() + 4 * C
            and dgamma.data_ptr() == db.data_ptr() + 4 * C and dbeta.data_ptr() == dgamma.data_ptr() + 4 * C):
        return db, dgamma, dbeta
    dst = torch.as_strided(gb, (3 * C,), (1,))
    src = torch.as_strided(db, (3 * C,), (1,))
    dst.add_(src)
    return None, None, None


def _wgrad_with_pro(a1, C1, pro1, a2, C2, pro2, dy, dw_tio, N, D, H, W, Cout, wp, wn, st):
    """3x3x3 weight gradient whose inputs may still carry a deferred BatchNorm + activation."""
    if pro1 is not None or pro2 is not None:
        s1, t1, sl1 = _pro_args(pro1)
        s2, t2, sl2 = _pro_args(pro2)
        if call_act('da_conv3d_k3_wgrad_pro', A(a1), C1, s1, t1, sl1, A(a2), C2, s2, t2, sl2, A(dy), ptr(dw_tio),
                    N, D, H, W, Cout, wp, wn, st, may_decline=True):
            return
        a1 = _apply_pro(a1, pro1, st)
        a2 = _apply_pro(a2, pro2, st) if a2 is not None else None
    call_act('da_conv3d_k3_wgrad', A(a1), C1, A(a2), C2, A(dy), ptr(dw_tio), None, N, D, H, W, Cout, 1, wp, wn, st)


class ConvBNActFn(Function):
    """unets.convBlock with batchnorm=True as ONE autograd node: Conv3d(k3,p1) on concat(x1, x2) -> BatchNorm3d -> LeakyReLU
    (unets.py:24-33).  Saves the raw conv output; the backward runs BN/act backward (which also yields the conv bias gradient),
    then the conv data / weight gradients."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope, *extra):
        transposed = bool(extra[0]) if extra else False        # ConvTranspose3d(k3,s1,p1) weight, see Conv3dK3Fn
        # extra[1] / extra[2]: (scale, shift, slope) still to be applied to x1 / x2 (LazyAct inputs); extra[3]: return the raw output
        # + (scale, shift) instead of the activated tensor (LazyAct output)
        pro1 = extra[1] if len(extra) > 1 else None
        pro2 = extra[2] if len(extra) > 2 else None
        lazy_out = bool(extra[3]) if len(extra) > 3 else False
        ctx.n_extra, ctx.transposed, ctx.lazy_out = len(extra), transposed, lazy_out
        a1 = ndhwc(x1)
        a2 = ndhwc(x2) if x2 is not None else None
        N, D, H, W, C1 = a1.shape
        C2 = a2.shape[-1] if a2 is not None else 0
        Cout, Cin = (weight.shape[1], weight.shape[0]) if transposed else (weight.shape[0], weight.shape[1])
        if Cin != C1 + C2 or tuple(weight.shape[2:]) != (3, 3, 3):
            raise ValueError('weight %s does not match input channels %d+%d' % (tuple(weight.shape), C1, C2))
        st = stream()
        w_tio = weight_tio(weight, 'iok_flip' if transposed else 'oik')
        y = _empty((N, D, H, W, Cout), a1, _act_dtype(Cout))
        wsb = nat.lib().da_conv3d_k3_ws_bytes(N, D, H, W, Cin, Cout, 1)
        wp, wn = _ws(wsb, a1)
        b = bias.detach().contiguous() if bias is not None else None
        partials = None
        import ctypes
        train_stats = bool(training or running_mean is None)
        pbuf = torch.empty((512, 2, Cout), dtype=torch.float64, device=a1.device) if train_stats else None
        npar = ctypes.c_int(0)
        cap = 512 if (train_stats and os.environ.get('DA_NO_FUSED_STATS') != '1') else 0
        done = False
        if pro1 is not None or pro2 is not None:
            s1, t1, sl1 = _pro_args(pro1)
            s2, t2, sl2 = _pro_args(pro2)
            use_pack(w_tio, 0, C1, C2, Cout, N, D, H, W)
            done = call_act('da_conv3d_k3_fwd_pro', A(a1), C1, s1, t1, sl1, A(a2), C2, s2, t2, sl2, ptr(w_tio), ptr(b), O(y),
                            N, D, H, W, Cout, -1.0, ptr(pbuf), cap, ctypes.byref(npar), wp, wn, st, may_decline=True)
            if not done:           # shape not taken by the prologue kernels: apply the deferred activation as its own pass
                a1, a2, pro1, pro2 = _apply_pro(a1, pro1, st), (_apply_pro(a2, pro2, st) if a2 is not None else None), None, None
        if not done:
            use_pack(w_tio, 0, C1, C2, Cout, N, D, H, W)
            if train_stats:
                # the MFMA epilogue accumulates the BatchNorm partial sums, so the statistics need no pass over y
                call_act('da_conv3d_k3_fwd_bnstats', A(a1), C1, A(a2), C2, ptr(w_tio), ptr(b), O(y), N, D, H, W, Cout, 1,
                         ptr(pbuf), cap, ctypes.byref(npar), wp, wn, st)
            else:
                call_act('da_conv3d_k3_fwd', A(a1), C1, A(a2), C2, ptr(w_tio), ptr(b), O(y), N, D, H, W, Cout, 1, -1.0, wp, wn, st)
        if train_stats:
            partials = (pbuf, npar.value)
        out, stats, g, cfg = _bn_forward(y, gamma, beta, running_mean, running_var, training, momentum, eps, slope, st, partials,
                                         apply=not lazy_out)
        ctx.dims = (N, D, H, W, C1, C2, Cout, wsb)
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        ctx.small_params = (bias, gamma, beta)
        ctx.pro_slopes = (pro1[2] if pro1 is not None else None, pro2[2] if pro2 is not None else None)
        ctx.save_for_backward(a1, a2, w_tio, y, stats, pro1[0] if pro1 is not None else None, pro1[1] if pro1 is not None else None,
                              pro2[0] if pro2 is not None else None, pro2[1] if pro2 is not None else None)
        if lazy_out:
            scale, shift = stats[2], stats[3]
            ctx.mark_non_differentiable(scale, shift)
            ctx.set_materialize_grads(False)          # no zero-filled gradients for (scale, shift): they are statistics, not graph nodes
            return ncdhw(y), scale, shift
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout, *unused):
        a1, a2, w_tio, y, stats, p1s, p1t, p2s, p2t = ctx.saved_tensors
        pro1 = (p1s, p1t, ctx.pro_slopes[0]) if p1s is not None else None
        pro2 = (p2s, p2t, ctx.pro_slopes[1]) if p2s is not None else None
        N, D, H, W, C1, C2, Cout, wsb = ctx.dims
        st = stream()
        dy, dgamma, dbeta, db = _bn_backward(ndhwc(gout), y, stats, ctx.cfg, ctx.has_bias, st, ctx.small_params)
        wp, wn = _ws(wsb, a1)
        dx1 = dx2 = None
        if ctx.needs_input_grad[0] or (a2 is not None and ctx.needs_input_grad[1]):
            dx1 = torch.empty_like(a1)
            dx2 = torch.empty_like(a2) if a2 is not None else None
            _serialize_matrix_kernels(54.0 * (C1 + C2) * Cout * N * D * H * W, N * D * H * W)
            use_pack(w_tio, 1, C1, C2, Cout, N, D, H, W)
            done = False
            if (FUSE_BN_BWD_STATS and pro1 is not None and ((a2 is None and C1 <= _DGRAD_BST_MAXC) or (a2 is not None and C1 == 32 and C2 == 16 and a2.dtype == torch.float32 and _DGRAD_BST_CONCAT))
                    and _matrix_mode == 'fp32_split' and dy.dtype == torch.float32
                    and a1.dtype == torch.float32 and p1s.data_ptr() - 8 * C1 == p1t.data_ptr() - 12 * C1 and p1s.untyped_storage().data_ptr() <= p1s.data_ptr() - 8 * C1):
                # (also while a HIP graph is being captured: whether the route exists is decided on the host, and a graphed step must run the same kernels --
                # the same summation order -- as an eager one: tests/test_gpu_dp.py compares them bit for bit)
                # the input was the raw output of a conv + BatchNorm block (a1) with its statistics rows (mean, rstd, scale, shift): that block's
                # BatchNorm-backward sums are accumulated in this data gradient's epilogue instead of a pass over (dx1, a1)
                import ctypes
                bst = torch.empty((512, 2, C1), dtype=torch.float64, device=a1.device)
                nb = ctypes.c_int(0)
                done = nat.call_supported('da_conv3d_k3_dgrad_bst', ptr(dy), ptr(w_tio), ptr(dx1), C1, ptr(dx2), C2, N, D, H, W, Cout, ptr(a1), ctypes.c_void_p(p1s.data_ptr() - 8 * C1),
                                          float(ctx.pro_slopes[0]), ptr(bst), 512, ctypes.byref(nb), wp, wn, st) and nb.value > 0
                if done:
                    _bwd_stats[dx1.data_ptr()] = (dx1, bst, nb.value, N * D * H * W, C1)
                else:
                    use_pack(w_tio, 1, C1, C2, Cout, N, D, H, W)          # (the hand-over was dropped with the declined call)
            if not done:
                call_act('da_conv3d_k3_dgrad', A(dy), ptr(w_tio), O(dx1), C1, O(dx2), C2, N, D, H, W, Cout, 1, wp, wn, st)
        kind_w = 'iok_flip' if ctx.transposed else 'oik'
        wt = WgradTarget(ctx.wparam, kind_w, w_tio) if ctx.needs_input_grad[2] else _NO_WGRAD_TARGET
        gw = wt.gw
        if not ctx.needs_input_grad[2]:
            dw = None                                  # frozen weight: no weight-gradient kernel at all
        elif gw is not None:
            global _last_side_flops
            _last_side_flops = 54.0 * (C1 + C2) * Cout * N * D * H * W
            on_main = LAST_WGRAD_ON_MAIN_BN and dx1 is None and dx2 is None      # the net's first layer: see Conv3dFn.backward (A/B switch; off: no gain measured on the seg step)
            side = torch.cuda.current_stream() if on_main else side_stream()
            if not on_main:
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                sst = stream()
                swp, swn = _ws(wsb, a1)
                _wgrad_with_pro(a1, C1, pro1, a2, C2, pro2, dy, wt.out(), N, D, H, W, Cout, swp, swn, sst)
                wt.finish()
            _side_keep.extend(t for t in (a1, a2, dy, p1s, p1t, p2s, p2t) if t is not None)
            dw = None
        else:
            dw_tio = torch.empty_like(w_tio)
            _wgrad_with_pro(a1, C1, pro1, a2, C2, pro2, dy, dw_tio, N, D, H, W, Cout, wp, wn, st)
            dw = grad_for_autograd(dw_tio, kind_w, ctx.wparam)
        db, dgamma, dbeta = _accumulate_small_grads(*ctx.small_params, db, dgamma, dbeta)
        return (ncdhw(dx1) if dx1 is not None else None, ncdhw(dx2) if dx2 is not None else None, dw, db, dgamma, dbeta,
                None, None, None, None, None, None) + (None,) * ctx.n_extra


FUSE_DECONV_BN_BWD = os.environ.get('DA_NO_DECONV_BN_BWD_FUSE') != '1'


def _deconv_bn_bwd_fused(ctx, go, a, w_tio, y, stats, N, D, H, W, Cin, Cout, st):
    """The up-sampler block's whole backward as ONE pass over (gout, y) (da_deconv_k2s2_bn_bwd: BatchNorm-backward apply on the fly + transposed-conv data
    and weight gradient; the tensor dy is never written).  Taken in training mode, fp32 tensors, Cin = Cout = 32 (the full-resolution link of UNet_light),
    when input and weight both want their gradients and the weight gradient goes straight into the optimiser's bucket or to a fresh tensor.  Runs on the
    MAIN stream (it replaces the apply pass and the data gradient as well; the side stream keeps the 3x3x3 weight gradients).  None: not taken."""
    M, C, slope, train, wsb = ctx.cfg
    if not (FUSE_DECONV_BN_BWD and train and Cin == 32 and Cout == 32 and go.dtype == torch.float32 and y.dtype == torch.float32 and a.dtype == torch.float32
            and go.is_contiguous() and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.has_bias):
        return None
    import ctypes
    pre = _bwd_stats.pop(go.data_ptr(), None) if _bwd_stats else None
    if pre is not None and not (pre[3] == M and pre[4] == C):
        pre = None
    dx = torch.empty_like(a)
    direct = _direct_small_target(ctx.small_params, C, True)
    dgb = direct if direct is not None else _empty((3, C), y)         # rows: transposed-conv bias, gamma, beta
    wt = WgradTarget(ctx.wparam, 'iok', w_tio)
    wp, wn = _ws(nat.lib().da_deconv_k2s2_bn_bwd_ws_bytes(N, D, H, W, Cin, Cout), a)
    # (the weight-gradient target: the optimiser's bucket slice / a scratch tensor added into it by wt.finish(), or a fresh tensor handed to autograd)
    if wt.gw is not None:
        dw_out = wt.out()
    else:
        dw_out = torch.empty_like(w_tio)
    ok = nat.call_supported('da_deconv_k2s2_bn_bwd', ptr(go), ptr(y), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), float(slope),
                            ptr(a), ptr(w_tio), ptr(dx), ptr(dw_out), ptr(dgb[0]), ptr(dgb[1]), ptr(dgb[2]),
                            N, D, H, W, Cin, Cout, ptr(pre[1]) if pre is not None else None, pre[2] if pre is not None else 0, wp, wn, st)
    if not ok:
        raise nat.NativeError('da_deconv_k2s2_bn_bwd declined a shape da_deconv_bn_bwd_supported accepts')
    if wt.gw is not None:
        wt.finish()
        dw = None
    else:
        dw = grad_for_autograd(dw_out, 'iok', ctx.wparam)
    if direct is not None:
        db = dgamma = dbeta = None
    else:
        db, dgamma, dbeta = _accumulate_small_grads(*ctx.small_params, dgb[0], dgb[1], dgb[2])
    return (ncdhw(dx), dw, db, dgamma, dbeta, None, None, None, None, None, None) + (None,) * ctx.n_extra


class DeconvBNActFn(Function):
    """unets.deconvBlock with batchnorm=True as one autograd node: ConvTranspose3d(k2,s2) -> BatchNorm3d -> LeakyReLU (unets.py:42-52)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope, *extra):
        lazy_out = bool(extra[0]) if extra else False            # return the raw output + (scale, shift): see LazyAct
        ctx.n_extra = len(extra)
        a = ndhwc(x)
        N, D, H, W, Cin = a.shape
        Cout = weight.shape[1]
        if weight.shape[0] != Cin or tuple(weight.shape[2:]) != (2, 2, 2):
            raise ValueError('ConvTranspose3d weight %s does not match input channels %d' % (tuple(weight.shape), Cin))
        st = stream()
        w_tio = weight_tio(weight, 'iok')
        y = _empty((N, 2 * D, 2 * H, 2 * W, Cout), a, _act_dtype(Cout))
        b = bias.detach().contiguous() if bias is not None else None
        wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), a)
        partials = None
        if (training or running_mean is None) and os.environ.get('DA_NO_FUSED_STATS') != '1' and os.environ.get('DA_NO_FUSED_DECONV_STATS') != '1':
            # the matrix-core epilogue accumulates the BatchNorm partial sums (one set per 256 coarse voxels): no statistics pass over y
            import ctypes
            nblk = (N * D * H * W + 255) // 256
            pbuf = torch.empty((nblk, 2, Cout), dtype=torch.float64, device=a.device)
            npar = ctypes.c_int(0)
            call_act('da_deconv_k2s2_fwd_bnstats', A(a), ptr(w_tio), ptr(b), O(y), N, D, H, W, Cin, Cout, ptr(pbuf), nblk, ctypes.byref(npar), wp, wn, st)
            partials = (pbuf, npar.value)
        else:
            call_act('da_deconv_k2s2_fwd', A(a), ptr(w_tio), ptr(b), O(y), N, D, H, W, Cin, Cout, wp, wn, st)
        out, stats, g, cfg = _bn_forward(y, gamma, beta, running_mean, running_var, training, momentum, eps, slope, st, partials, apply=not lazy_out)
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.wparam = weight
        ctx.small_params = (bias, gamma, beta)
        ctx.save_for_backward(a, w_tio, y, stats)
        if lazy_out:
            scale, shift = stats[2], stats[3]
            ctx.mark_non_differentiable(scale, shift)
            ctx.set_materialize_grads(False)          # no zero-filled gradients for (scale, shift): they are statistics, not graph nodes
            return ncdhw(y), scale, shift
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout, *unused):
        a, w_tio, y, stats = ctx.saved_tensors
        N, D, H, W, Cin = a.shape
        Cout = w_tio.shape[2]
        st = stream()
        fused = _deconv_bn_bwd_fused(ctx, ndhwc(gout), a, w_tio, y, stats, N, D, H, W, Cin, Cout, st)
        if fused is not None:
            return fused
        dy, dgamma, dbeta, db = _bn_backward(ndhwc(gout), y, stats, ctx.cfg, ctx.has_bias, st, ctx.small_params)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(a)
            wp, wn = _ws(nat.lib().da_pointwise_ws_bytes(8, Cin, Cout), a)
            call_act('da_deconv_k2s2_dgrad', A(dy), ptr(w_tio), O(dx), N, D, H, W, Cin, Cout, wp, wn, st)
        wt = WgradTarget(ctx.wparam, 'iok', w_tio) if ctx.needs_input_grad[1] else _NO_WGRAD_TARGET
        gw = wt.gw
        if not ctx.needs_input_grad[1]:
            dw = None                                  # frozen weight
        elif gw is not None:
            def side_work():                        # HBM-bound: overlaps the MFMA-bound conv data gradients on the main stream
                swp, swn = _ws(nat.lib().da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout), a)
                call_act('da_deconv_k2s2_wgrad', A(a), A(dy), ptr(wt.out()), None, N, D, H, W, Cin, Cout, swp, swn, stream())
                wt.finish()
            _run_on_side(side_work, (a, dy))
            dw = None
        else:
            dw_tio = torch.empty_like(w_tio)
            wp, wn = _ws(nat.lib().da_deconv_k2s2_wgrad_ws_bytes(N, D, H, W, Cin, Cout), a)
            call_act('da_deconv_k2s2_wgrad', A(a), A(dy), ptr(dw_tio), None, N, D, H, W, Cin, Cout, wp, wn, st)
            dw = grad_for_autograd(dw_tio, 'iok', ctx.wparam)
        db, dgamma, dbeta = _accumulate_small_grads(*ctx.small_params, db, dgamma, dbeta)
        return ((ncdhw(dx) if dx is not None else None), dw, db, dgamma, dbeta, None, None, None, None, None, None) + (None,) * ctx.n_extra


class ActFn(Function):
    """Stand-alone ReLU / LeakyReLU (used when a block has no BatchNorm and the conv did not fuse it)."""

    @staticmethod
    def forward(ctx, x, slope):
        a = ndhwc(x)
        C = a.shape[-1]
        ones = torch.ones(C, dtype=torch.float32, device=a.device)
        zeros = torch.zeros(C, dtype=torch.float32, device=a.device)
        out = torch.empty_like(a)
        call_act('da_bn_act_fwd', A(a), ptr(ones), ptr(zeros), float(slope), O(out), a.numel() // C, C, stream())
        ctx.slope = float(slope)
        ctx.save_for_backward(out)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        out, = ctx.saved_tensors
        g = ndhwc(gout)
        dx = torch.empty_like(out)
        call_act('da_act_bwd', A(g), A(out), ctx.slope, O(dx), g.numel(), stream())
        return ncdhw(dx), None


# ------------------------------------------------------------------------------------------------
# pooling / resampling
# ------------------------------------------------------------------------------------------------
class MaxPool2Fn(Function):
    """nn.MaxPool3d(2) (unets.py:230,267)."""

    @staticmethod
    def forward(ctx, x):
        a = ndhwc(x)
        N, D, H, W, C = a.shape
        out = _empty((N, D // 2, H // 2, W // 2, C), a, a.dtype)
        call_act('da_maxpool2_fwd', A(a), O(out), N, D, H, W, C, stream())
        ctx.save_for_backward(a)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        a, = ctx.saved_tensors
        N, D, H, W, C = a.shape
        g = ndhwc(gout)
        dx = torch.empty_like(a)
        call_act('da_maxpool2_bwd', A(g), A(a), O(dx), N, D, H, W, C, stream())
        return ncdhw(dx)


class MaxPool2SkipFn(Function):
    """x -> (x as the skip tensor, MaxPool3d(2)(x)) as ONE autograd node (unets.py:266-267,275): the two gradients reaching x are
    summed inside the max-pool backward kernel instead of by a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, *extra):
        # extra = (scale, shift, slope): x is a RAW producer output (LazyAct); its BatchNorm + activation is applied in the same pass that
        # pools it, and the activated tensor written by that pass is the skip tensor
        ctx.n_extra = len(extra)
        a = ndhwc(x)
        N, D, H, W, C = a.shape
        out = _empty((N, D // 2, H // 2, W // 2, C), a, a.dtype)
        st = stream()
        raw = None
        ctx.pro_slope = None
        if extra:
            act = torch.empty_like(a)
            if not call_act('da_maxpool2_fwd_pro', A(a), ptr(extra[0]), ptr(extra[1]), float(extra[2]), O(act), O(out), N, D, H, W, C, st, may_decline=True):
                act = _apply_pro(a, extra, st)
                call_act('da_maxpool2_fwd', A(act), O(out), N, D, H, W, C, st)
            ps, pt = extra[0], extra[1]
            if (FUSE_BN_BWD_STATS and os.environ.get('DA_NO_POOL_BST') != '1' and a.dtype == torch.float32 and ps.data_ptr() - 8 * C == pt.data_ptr() - 12 * C
                    and ps.untyped_storage().data_ptr() <= ps.data_ptr() - 8 * C):
                # (scale, shift) are rows of the producer's [mean | rstd | scale | shift] buffer: its BatchNorm-backward sums can ride on this node's backward
                raw, ctx.pro_slope = a, float(extra[2])
            a = act
        else:
            call_act('da_maxpool2_fwd', A(a), O(out), N, D, H, W, C, st)
        if raw is not None:
            ctx.save_for_backward(a, raw, extra[0])
        else:
            ctx.save_for_backward(a)
        return ncdhw(a), ncdhw(out)

    @staticmethod
    def backward(ctx, gskip, gpool):
        a = ctx.saved_tensors[0]
        N, D, H, W, C = a.shape
        if gpool is None:
            return (gskip,) + (None,) * ctx.n_extra
        g = ndhwc(gpool)
        dx = torch.empty_like(a)
        if len(ctx.saved_tensors) == 3 and FUSE_BN_BWD_STATS and g.dtype == torch.float32:
            # the pooled tensor was the raw output of a conv + BatchNorm block: dx and that block's BatchNorm-backward sums from one kernel
            import ctypes
            raw, ps = ctx.saved_tensors[1], ctx.saved_tensors[2]
            bst = torch.empty((1024, 2, C), dtype=torch.float64, device=a.device)
            nb = ctypes.c_int(0)
            gs = ndhwc(gskip) if gskip is not None else None
            if nat.call_supported('da_maxpool2_bwd_bst', ptr(g), ptr(raw), ptr(gs), ptr(dx), N, D, H, W, C, ctypes.c_void_p(ps.data_ptr() - 8 * C),
                                  ctx.pro_slope, ptr(bst), 1024, ctypes.byref(nb), stream()) and nb.value > 0:
                _bwd_stats[dx.data_ptr()] = (dx, bst, nb.value, N * D * H * W, C)
                return (ncdhw(dx),) + (None,) * ctx.n_extra
        if gskip is None:
            call_act('da_maxpool2_bwd', A(g), A(a), O(dx), N, D, H, W, C, stream())
        else:
            call_act('da_maxpool2_bwd_add', A(g), A(a), A(ndhwc(gskip)), O(dx), N, D, H, W, C, stream())
        return (ncdhw(dx),) + (None,) * ctx.n_extra


class UpsampleNearestFn(Function):
    """F.interpolate(x, size=...) with the default 'nearest' mode (voxel_morph.py:72,74,76,80)."""

    @staticmethod
    def forward(ctx, x, size):
        a = ndhwc(x)
        N, D, H, W, C = a.shape
        Do, Ho, Wo = int(size[0]), int(size[1]), int(size[2])
        out = _empty((N, Do, Ho, Wo, C), a, a.dtype)
        call_act('da_upsample_nearest_fwd', A(a), O(out), N, D, H, W, C, Do, Ho, Wo, stream())
        ctx.dims = (N, D, H, W, C, Do, Ho, Wo)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        N, D, H, W, C, Do, Ho, Wo = ctx.dims
        g = ndhwc(gout)
        dx = _empty((N, D, H, W, C), g, g.dtype)
        call_act('da_upsample_nearest_bwd', A(g), O(dx), N, D, H, W, C, Do, Ho, Wo, stream())
        return ncdhw(dx), None


class WarpFn(Function):
    """deform = disp + identity; warped = grid_sample(src, deform, bilinear, zeros, align_corners=True)
    (voxel_morph.py:85-91, lib/utils.py:89-102).  Returns (warped, deform)."""

    @staticmethod
    def forward(ctx, src, disp):
        s = ndhwc(src)
        u = ndhwc(disp)
        N, D, H, W, C = s.shape
        if tuple(u.shape) != (N, D, H, W, 3):
            raise ValueError('displacement field must be N x 3 x D x H x W matching the source volume')
        out = torch.empty_like(s)
        deform = torch.empty_like(u)
        call('da_warp_fwd', ptr(s), ptr(u), ptr(deform), ptr(out), N, D, H, W, C, stream())
        ctx.save_for_backward(s, u)
        return ncdhw(out), ncdhw(deform)

    @staticmethod
    def backward(ctx, g_out, g_deform):
        s, u = ctx.saved_tensors
        N, D, H, W, C = s.shape
        st = stream()
        d_disp = d_src = None
        go = ndhwc(g_out) if g_out is not None else torch.zeros_like(s)
        if ctx.needs_input_grad[1]:
            d_disp = torch.empty_like(u)
        if ctx.needs_input_grad[0] and DETERMINISTIC:
            # parity runs: order-independent fixed-point accumulation instead of float atomics (da_warp_bwd_dsrc_det)
            d_src = torch.empty_like(s)
            wp, wn = _ws(nat.lib().da_warp_bwd_dsrc_det_ws_bytes(N, D, H, W, C), s)
            call('da_warp_bwd_dsrc_det', ptr(go), ptr(u), ptr(d_src), N, D, H, W, C, wp, wn, st)
            if d_disp is not None:
                call('da_warp_bwd', ptr(go), ptr(s), ptr(u), ptr(d_disp), None, N, D, H, W, C, st)
        else:
            if ctx.needs_input_grad[0]:
                d_src = torch.zeros_like(s)
            call('da_warp_bwd', ptr(go), ptr(s), ptr(u), ptr(d_disp), ptr(d_src), N, D, H, W, C, st)
        if d_disp is not None and g_deform is not None:
            d_disp = d_disp + ndhwc(g_deform)
        return (ncdhw(d_src) if d_src is not None else None), (ncdhw(d_disp) if d_disp is not None else None)


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
_WEIGHT_TYPES = {'Uniform': 0, 'Simple': 1, 'Volume': 2}


class WarpLabelsFn(Function):
    """warp(one_hot(labels), identity + disp) without the one-hot tensor (joint step, registration phase): N x C x D x H x W."""

    @staticmethod
    def forward(ctx, labels, disp, n_classes):
        u = ndhwc(disp)
        N, D, H, W, _ = u.shape
        lab, nbytes = _labels(labels.reshape(N, -1))
        if lab.shape[1] != D * H * W:
            raise ValueError('label map and displacement field must cover the same volume')
        out = _empty((N, D, H, W, int(n_classes)), u)
        call('da_warp_labels_fwd', ptr(lab), nbytes, ptr(u), ptr(out), N, D, H, W, int(n_classes), stream())
        ctx.cfg = (N, D, H, W, int(n_classes), nbytes)
        ctx.save_for_backward(lab, u)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        lab, u = ctx.saved_tensors
        N, D, H, W, C, nbytes = ctx.cfg
        d_disp = torch.empty_like(u)
        call('da_warp_labels_bwd', ptr(ndhwc(gout)), ptr(lab), nbytes, ptr(u), ptr(d_disp), N, D, H, W, C, stream())
        return None, ncdhw(d_disp), None


# ------------------------------------------------------------------------------------------------
# fused segmentation head + softmax + Dice
# ------------------------------------------------------------------------------------------------
# models/segmentation.py:141-157 calls `output = self.model(images); loss = self.criterion(output, truths)`.  With `model.lazy_head`
# set (SegmentationExperiment and bench.py do, for the softmax-Dice criterion) the network's 1x1x1 output convolution is not run on
# its own: forward() returns a LazyLogits, and DiceLossMultiClass evaluates head + softmax + Dice in one kernel pair (da_head_dice_*)
# that never writes the 629 MB-per-volume logits.  Anything else that touches the object gets real logits via .materialize().
FUSE_HEAD_DICE = os.environ.get('DA_FUSE_HEAD_DICE', '1') != '0'


def head_dice_supported(cin, n_classes):
    return cin in (16, 64) and n_classes in (16, 32)


class LazyLogits(object):
    """The segmentation head's output before anyone has asked for it: (input tensor or LazyAct, weight, bias)."""
    __slots__ = ('x', 'weight', 'bias', '_logits')

    def __init__(self, x, weight, bias):
        self.x, self.weight, self.bias, self._logits = x, weight, bias, None

    @property
    def shape(self):
        xs = self.x.shape
        return torch.Size((xs[0], self.weight.shape[0]) + tuple(xs[2:]))

    def materialize(self):
        if self._logits is None:
            if isinstance(self.x, LazyAct):
                self._logits = Conv1x1Fn.apply(self.x.raw, self.weight, self.bias, (self.x.scale, self.x.shift, self.x.slope))
            else:
                self._logits = Conv1x1Fn.apply(self.x, self.weight, self.bias)
        return self._logits

    def detach(self):
        return self.materialize().detach()


def materialize_logits(x):
    return x.materialize() if isinstance(x, LazyLogits) else x


class HeadDiceFn(Function):
    """Dice(softmax(conv1x1(x)), labels) as one node (da_head_dice_fwd / _bwd): the logits are never written; the backward pass
    recomputes them from x and produces dx, dW, db directly."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, weight_type, no_bg, eps, *extra):
        pro = extra[0] if extra else None            # (scale, shift, slope) still to be applied to x (LazyAct input)
        ctx.n_extra = len(extra)
        a = ndhwc(x)
        N, D, H, W, Cin = a.shape
        C = weight.shape[0]
        V = D * H * W
        st = stream()
        lab, lb = _labels(labels.reshape(N, -1))
        if lab.shape[1] != V:
            raise ValueError('label map and input must cover the same volume')
        w_io = weight_tio(weight, 'oik').view(Cin, C)
        b = bias.detach().contiguous() if bias is not None else None
        loss = _empty((1,), a)
        coef = _empty((2, N, C), a)
        wsb = nat.lib().da_head_dice_ws_bytes(N, V, Cin, C)
        wp, wn = _ws(wsb, a)
        ps, pt, sl = _pro_args(pro)
        call_act('da_head_dice_fwd', A(a), ps, pt, sl, ptr(w_io), ptr(b), ptr(lab), lb, N, V, Cin, C, _WEIGHT_TYPES[weight_type], 1 if no_bg else 0,
                 float(eps), ptr(loss), ptr(coef), wp, wn, st)
        ctx.wparam = weight
        ctx.cfg = (N, V, Cin, C, lb, wsb, pro[2] if pro is not None else -1.0, bias is not None)
        ctx.save_for_backward(a, w_io, b, lab, coef, pro[0] if pro is not None else None, pro[1] if pro is not None else None)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        a, w_io, b, lab, coef, ps, pt = ctx.saved_tensors
        N, V, Cin, C, lb, wsb, sl, has_bias = ctx.cfg
        st = stream()
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(a)
        dw_io = torch.empty_like(w_io)
        db = _empty((C,), a) if has_bias else None
        wp, wn = _ws(wsb, a)
        if (FUSE_BN_BWD_STATS and ps is not None and Cin == 16 and a.dtype == torch.float32 and ps.data_ptr() - 8 * Cin == pt.data_ptr() - 12 * Cin
                and ps.untyped_storage().data_ptr() <= ps.data_ptr() - 8 * Cin):
            # x is the raw output of a conv + BatchNorm block whose statistics are the rows (mean, rstd, scale, shift) of one [4][C] buffer
            # (_bn_forward): that block's BatchNorm-backward sums are accumulated here, where x and its gradient are in registers anyway
            import ctypes
            bst = torch.empty((1024, 2, Cin), dtype=torch.float64, device=a.device)
            nb = ctypes.c_int(0)
            call('da_head_dice_bwd_bst', ptr(a), ptr(ps), ptr(pt), float(sl), ctypes.c_void_p(ps.data_ptr() - 8 * Cin), ptr(w_io), ptr(b), ptr(lab), lb,
                 ptr(coef), ptr(gl), ptr(dx), ptr(dw_io), ptr(db), N, V, Cin, C, ptr(bst), 1024, ctypes.byref(nb), wp, wn, st)
            if nb.value > 0:
                _bwd_stats[dx.data_ptr()] = (dx, bst, nb.value, N * V, Cin)
        else:
            call_act('da_head_dice_bwd', A(a), ptr(ps), ptr(pt), float(sl), ptr(w_io), ptr(b), ptr(lab), lb, ptr(coef), ptr(gl),
                     O(dx), ptr(dw_io), ptr(db), N, V, Cin, C, wp, wn, st)
        dw = grad_for_autograd(dw_io.view(1, Cin, C), 'oik', ctx.wparam)
        return (ncdhw(dx), dw, db, None, None, None, None) + (None,) * ctx.n_extra


def fused_anatomy_supported(n_classes):
    """The fused anatomy-loss kernels take class counts whose 4-channel lane groups are a power of two (4, 8, 16, 32, 64)."""
    q = n_classes // 4
    return n_classes % 4 == 0 and 1 <= q <= 16 and (q & (q - 1)) == 0


class LabelWarpDiceFn(Function):
    """Dice(warp(one_hot(labels_m), identity + disp), one_hot(labels_t)) -- the registration phase's anatomy loss (models/joint.py) --
    straight from the two label maps: neither the warped one-hot tensor nor Dice's gradient tensor is materialised
    (da_label_warp_dice_fwd / _bwd).  Same value and d loss / d disp as WarpLabelsFn followed by DiceFn(softmax=False)."""

    @staticmethod
    def forward(ctx, labels_m, labels_t, disp, n_classes, weight_type, no_bg, eps):
        u = ndhwc(disp)
        N, D, H, W, _ = u.shape
        lm, bm = _labels(labels_m.reshape(N, -1))
        lt, bt = _labels(labels_t.reshape(N, -1))
        if lm.shape[1] != D * H * W or lt.shape[1] != D * H * W:
            raise ValueError('label maps and displacement field must cover the same volume')
        C = int(n_classes)
        loss = _empty((1,), u)
        coef = _empty((2, N, C), u)
        wp, wn = _ws(nat.lib().da_label_warp_dice_ws_bytes(N, C), u)
        call('da_label_warp_dice_fwd', ptr(lm), bm, ptr(lt), bt, ptr(u), N, D, H, W, C, _WEIGHT_TYPES[weight_type], 1 if no_bg else 0,
             float(eps), ptr(loss), ptr(coef), wp, wn, stream())
        ctx.cfg = (N, D, H, W, C, bm, bt)
        ctx.save_for_backward(lm, lt, u, coef)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        lm, lt, u, coef = ctx.saved_tensors
        N, D, H, W, C, bm, bt = ctx.cfg
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        d_disp = torch.empty_like(u)
        call('da_label_warp_dice_bwd', ptr(lm), bm, ptr(lt), bt, ptr(u), ptr(coef), ptr(gl), ptr(d_disp), N, D, H, W, C, stream())
        return None, None, ncdhw(d_disp), None, None, None, None


class SegPhaseLossFn(Function):
    """The segmentation phase's two Dice terms as ONE node (models/joint.py):
        l_sup  = Dice(softmax(logits), labels_m)                                   (None labels_m: 0)
        l_anat = Dice(warp(softmax(logits), identity + disp), one_hot(labels_t))    with disp a constant.
    Forward runs the existing kernels (Dice, softmax, warp, Dice).  Backward: Dice's gradient is rank-structured, so the adjoint
    warp collapses to two scatters of 8 atomics per voxel (da_warp_adjoint_labels) and ONE dense pass forms the gradient with respect
    to the logits through the softmax Jacobian for both terms (da_seg_anat_dlogits) -- instead of Dice backward x 2, a 256-atomics-per-
    voxel scatter, softmax backward and the sum of the two logit gradients."""

    @staticmethod
    def forward(ctx, logits, labels_m, disp, labels_t, weight_type, no_bg, eps):
        a = ndhwc(logits)
        u = ndhwc(disp.detach())
        N, D, H, W, C = a.shape
        V = D * H * W
        st = stream()
        wt, nb = _WEIGHT_TYPES[weight_type], 1 if no_bg else 0
        lt, bt = _labels(labels_t.reshape(N, -1))
        lm = coef_s = None
        bm = 0
        loss_s = torch.zeros((1,), dtype=torch.float32, device=a.device)
        # ONE workspace for both Dice entries (a second, larger _ws request on the same stream would replace -- and free -- the first)
        wp, wn = _ws(max(nat.lib().da_dice_ws_bytes(N, V, C), nat.lib().da_warp_dice_ws_bytes(N, C)), a)
        prob = torch.empty_like(a)
        fused_fwd = os.environ.get('DA_NO_FUSED_SEGPHASE_FWD') != '1'
        if labels_m is not None:
            lm, bm = _labels(labels_m.reshape(N, -1))
            coef_s = _empty((2, N, C), a)
            # supervised Dice sums and softmax(logits) from ONE pass over the logits (da_softmax_dice_fwd) ...
            if not (fused_fwd and call_supported('da_softmax_dice_fwd', ptr(a), ptr(lm), bm, ptr(prob), N, V, C, wt, nb, float(eps),
                                                 ptr(loss_s), ptr(coef_s), wp, wn, st)):
                call('da_dice_fwd', ptr(a), ptr(lm), bm, None, N, V, C, 1, wt, nb, float(eps), ptr(loss_s), ptr(coef_s), wp, wn, st)
                call('da_softmax_fwd', ptr(a), ptr(prob), N * V, C, st)
        else:
            call('da_softmax_fwd', ptr(a), ptr(prob), N * V, C, st)
        loss_a = _empty((1,), a)
        coef_a = _empty((2, N, C), a)
        # ... and the anatomy Dice of the WARPED probabilities without writing the warped tensor (da_warp_dice_fwd)
        wp2, wn2 = wp, wn
        warped = None
        if not (fused_fwd and call_supported('da_warp_dice_fwd', ptr(prob), ptr(u), ptr(lt), bt, N, D, H, W, C, wt, nb, float(eps),
                                             ptr(loss_a), ptr(coef_a), wp2, wn2, st)):
            warped = torch.empty_like(a)
            call('da_warp_fwd', ptr(prob), ptr(u), None, ptr(warped), N, D, H, W, C, st)
            call('da_dice_fwd', ptr(warped), ptr(lt), bt, None, N, V, C, 0, wt, nb, float(eps), ptr(loss_a), ptr(coef_a), wp, wn, st)
        ctx.cfg = (N, D, H, W, C, bm, bt)
        ctx.scratch = warped                       # (separate route: dead after the Dice sums, reused as B in backward)
        ctx.save_for_backward(prob, u, lm, lt, coef_s, coef_a)
        return loss_s.reshape(()), loss_a.reshape(())

    @staticmethod
    def backward(ctx, g_s, g_a):
        prob, u, lm, lt, coef_s, coef_a = ctx.saved_tensors
        N, D, H, W, C, bm, bt = ctx.cfg
        st = stream()
        zero = lambda: torch.zeros((1,), dtype=torch.float32, device=prob.device)
        gs = g_s.detach().reshape(1).to(torch.float32).contiguous() if g_s is not None else zero()
        ga = g_a.detach().reshape(1).to(torch.float32).contiguous() if g_a is not None else zero()
        B = ctx.scratch if ctx.scratch is not None else torch.empty_like(prob)
        ctx.scratch = None
        # labels outside [0, C) (the reference's one-hot scatter would raise on them) put their weights into a separate array: uint8 labels
        # with C = 256 cannot have any, everything else gets the array
        A = None if (bt == 1 and C >= 256) else _empty((N, D * H * W), prob)
        call('da_warp_adjoint_labels', ptr(lt), bt, ptr(u), ptr(A), ptr(B), N, D, H, W, C, st)
        dlogits = torch.empty_like(prob)          # B is class-major ([N][C][V]): the gradient cannot be formed in place
        call('da_seg_anat_dlogits', ptr(prob), ptr(lm), bm, ptr(A), ptr(B), ptr(dlogits), ptr(coef_s), ptr(coef_a),
             ptr(gs) if coef_s is not None else None, ptr(ga), N, D * H * W, C, st)
        return ncdhw(dlogits), None, None, None, None, None, None


class DiceFn(Function):
    """DiceLossMultiClass.forward (lib/loss.py:410-476) with the softmax and the one-hot fused in."""

    @staticmethod
    def forward(ctx, source, target, soft_target, weight_type, no_bg, softmax, eps):
        s = ndhwc(source)
        N, C = s.shape[0], s.shape[-1]
        V = s.numel() // (N * C)
        st = stream()
        lab = soft = None
        lab_bytes = 0
        if soft_target is not None:
            soft = ndhwc(soft_target)
        else:
            lab, lab_bytes = _labels(target)
        loss = _empty((1,), s)
        coef = _empty((2, N, C), s)
        wp, wn = _ws(nat.lib().da_dice_ws_bytes(N, V, C), s)
        call('da_dice_fwd', ptr(s), ptr(lab), lab_bytes, ptr(soft), N, V, C, 1 if softmax else 0,
             _WEIGHT_TYPES[weight_type], 1 if no_bg else 0, float(eps), ptr(loss), ptr(coef), wp, wn, st)
        ctx.cfg = (N, V, C, 1 if softmax else 0, lab_bytes)
        ctx.save_for_backward(s, lab, soft, coef)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        s, lab, soft, coef = ctx.saved_tensors
        N, V, C, softmax, lab_bytes = ctx.cfg
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        d_src = torch.empty_like(s)
        call('da_dice_bwd', ptr(s), ptr(lab), lab_bytes, ptr(soft), ptr(coef), ptr(gl), ptr(d_src), N, V, C, softmax, stream())
        return ncdhw(d_src), None, None, None, None, None, None


class SoftmaxFn(Function):
    """F.softmax(x, dim=1) on N x C x D x H x W (lib/loss.py:427; joint step: probabilities to warp)."""

    @staticmethod
    def forward(ctx, x):
        a = ndhwc(x)
        C = a.shape[-1]
        out = torch.empty_like(a)
        call('da_softmax_fwd', ptr(a), ptr(out), a.numel() // C, C, stream())
        ctx.save_for_backward(out)
        return ncdhw(out)

    @staticmethod
    def backward(ctx, gout):
        out, = ctx.saved_tensors
        C = out.shape[-1]
        g = ndhwc(gout)
        dx = torch.empty_like(out)
        call('da_softmax_bwd', ptr(g), ptr(out), ptr(dx), out.numel() // C, C, stream())
        return ncdhw(dx)


class NCCFn(Function):
    """NormalizedCrossCorrelationLoss.forward (lib/loss.py:493-501)."""

    @staticmethod
    def forward(ctx, x, y):
        nat.require_cuda(x, y)
        a = ndhwc(x) if x.dim() == 5 else x.contiguous()
        b = ndhwc(y) if y.dim() == 5 else y.contiguous()
        N = a.shape[0]
        V = a.numel() // N
        loss = _empty((1,), a)
        stats = torch.empty((N, 8), dtype=torch.float64, device=a.device)
        wp, wn = _ws(nat.lib().da_ncc_ws_bytes(N, V), a)
        call('da_ncc_fwd', ptr(a), ptr(b), N, V, ptr(loss), ptr(stats), wp, wn, stream())
        ctx.five = (x.dim() == 5, y.dim() == 5)
        ctx.save_for_backward(a, b, stats)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        a, b, stats = ctx.saved_tensors
        N = a.shape[0]
        V = a.numel() // N
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        call('da_ncc_bwd', ptr(a), ptr(b), ptr(stats), ptr(gl), ptr(dx), ptr(dy), N, V, stream())
        if dx is not None and ctx.five[0]:
            dx = ncdhw(dx)
        if dy is not None and ctx.five[1]:
            dy = ncdhw(dy)
        return dx, dy


class BendingFn(Function):
    """BendingEnergyLoss.forward, norm='L2' (lib/loss.py:687-730)."""

    @staticmethod
    def forward(ctx, disp, spacing, normalize, norm=2):
        u = ndhwc(disp)
        N, D, H, W, C = u.shape
        if C != 3:
            raise ValueError('bending energy expects a N x 3 x D x H x W displacement field')
        sp = (ctypes_float3(spacing))
        loss = _empty((1,), u)
        wp, wn = _ws(nat.lib().da_bending_ws_bytes(N, D, H, W), u)
        call('da_bending_fwd', ptr(u), N, D, H, W, sp, 1 if normalize else 0, int(norm), ptr(loss), wp, wn, stream())
        ctx.cfg = (tuple(float(s) for s in spacing), bool(normalize), int(norm))
        ctx.save_for_backward(u)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        u, = ctx.saved_tensors
        N, D, H, W, _ = u.shape
        spacing, normalize, norm = ctx.cfg
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        du = torch.empty_like(u)
        call('da_bending_bwd', ptr(u), ptr(gl), ptr(du), N, D, H, W, ctypes_float3(spacing), 1 if normalize else 0, norm, stream())
        return ncdhw(du), None, None, None


class XentFn(Function):
    """Cross-entropy family of the loss registry (lib/loss.py:739-761) on N x C x D x H x W logits: mode 0 nn.CrossEntropyLoss,
    1 FocalLoss.forward (lib/loss.py:181-213), 2 SoftCrossEntropy.forward with a probability target (:115-154)."""

    @staticmethod
    def forward(ctx, logits, labels, soft_target, alpha, mode, softmax, gamma, ignore_index, reduction):
        a = ndhwc(logits) if logits.dim() == 5 else logits.contiguous()
        nat.require_cuda(a)
        C = a.shape[-1]
        M = a.numel() // C
        lab = soft = None
        lab_bytes = 0
        if soft_target is not None:
            soft = ndhwc(soft_target) if soft_target.dim() == 5 else soft_target.contiguous()
        else:
            lab, lab_bytes = _labels(labels.reshape(-1))
            if lab.numel() != M:
                raise ValueError('target has %d elements for %d voxels' % (lab.numel(), M))
            if CHECK_LABELS:
                # torch.nn.CrossEntropyLoss raises on a target outside [0, C) that is not ignore_index; the kernels skip such voxels (zero loss,
                # zero gradient, not counted).  The check costs a device synchronisation, hence the switch (DA_CHECK_LABELS=1).
                bad = (lab != int(ignore_index)) & ((lab < 0) | (lab >= C)) if lab.dtype == torch.int64 else (lab >= C)
                if bool(bad.any()):
                    raise IndexError('Target %d is out of bounds.' % int(lab[bad][0].item()))
        al = alpha.detach().to(a.device, torch.float32).reshape(-1).contiguous() if alpha is not None else None
        loss, denom = _empty((1,), a), _empty((1,), a)
        wp, wn = _ws(nat.lib().da_xent_ws_bytes(), a)
        call('da_xent_fwd', ptr(a), ptr(lab), lab_bytes, ptr(soft), ptr(al), M, C, int(mode), 1 if softmax else 0, float(gamma),
             int(ignore_index), int(reduction), ptr(loss), ptr(denom), wp, wn, stream())
        ctx.cfg = (M, C, int(mode), 1 if softmax else 0, float(gamma), int(ignore_index), lab_bytes, logits.dim() == 5)
        ctx.save_for_backward(a, lab, soft, al, denom)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        a, lab, soft, al, denom = ctx.saved_tensors
        M, C, mode, softmax, gamma, ignore, lab_bytes, five = ctx.cfg
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(a)
        call('da_xent_bwd', ptr(a), ptr(lab), lab_bytes, ptr(soft), ptr(al), ptr(gl), ptr(denom), ptr(dx), M, C, mode, softmax, gamma, ignore, stream())
        return (ncdhw(dx) if five else dx), None, None, None, None, None, None, None, None


class LNCCFn(Function):
    """Local normalised cross-correlation over all-ones k^3 windows (dilation d, stride s): VoxelMorphLNCC.forward
    (lib/loss.py:599-617: k = filter_size, d = s = 1) and one scale of LNCCLoss.forward (:541-584).  Five box sums as three
    separable passes + fused cc reduction; backward through the adjoint box filter."""

    @staticmethod
    def forward(ctx, I, J, filter_size, eps, dilation=1, stride=1):
        if I.shape != J.shape or I.dim() != 5 or I.shape[1] != 1:
            raise ValueError('LNCC expects two N x 1 x D x H x W volumes of the same shape')
        nat.require_cuda(I); nat.require_cuda(J)
        a, b = I.detach().contiguous().float(), J.detach().contiguous().float()
        N, _, D, H, W = a.shape
        F_, d_, s_ = int(filter_size), int(dilation), int(stride)
        span = d_ * (F_ - 1) + 1
        if min(D, H, W) < span:
            raise RuntimeError('LNCC window %d (dilation %d) larger than the volume %s' % (F_, d_, (D, H, W)))
        od = tuple((L - span) // s_ + 1 for L in (D, H, W))
        loss = _empty((1,), a)
        sums = torch.empty((5, N) + od, dtype=torch.float32, device=a.device)
        wsb = nat.lib().da_lncc_ws_bytes(N, D, H, W, F_, d_, s_)
        wp, wn = _ws(wsb, a)
        call('da_lncc_fwd', ptr(a), ptr(b), N, D, H, W, F_, d_, s_, float(eps), ptr(loss), ptr(sums), wp, wn, stream())
        ctx.cfg = (F_, d_, s_, float(eps), wsb)
        ctx.save_for_backward(a, b, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        a, b, sums = ctx.saved_tensors
        F_, d_, s_, eps, wsb = ctx.cfg
        N, _, D, H, W = a.shape
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        dI = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        dJ = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        wp, wn = _ws(wsb, a)
        call('da_lncc_bwd', ptr(a), ptr(b), ptr(sums), ptr(gl), ptr(dI), ptr(dJ), N, D, H, W, F_, d_, s_, eps, wp, wn, stream())
        return dI, dJ, None, None, None, None


class GradLossFn(Function):
    """gradientLoss.forward (lib/loss.py:640-671), norm 'L2' or 'L1', with the reference's sign quirk along H and W."""

    @staticmethod
    def forward(ctx, disp, spacing, normalize, norm):
        u = ndhwc(disp)
        N, D, H, W, C = u.shape
        if C != 3:
            raise ValueError('gradientLoss expects a N x 3 x D x H x W displacement field')
        loss = _empty((1,), u)
        wp, wn = _ws(nat.lib().da_gradloss_ws_bytes(N, D, H, W), u)
        call('da_gradloss_fwd', ptr(u), N, D, H, W, ctypes_float3(spacing), 1 if normalize else 0, int(norm), ptr(loss), wp, wn, stream())
        ctx.cfg = (tuple(float(s) for s in spacing), bool(normalize), int(norm))
        ctx.save_for_backward(u)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        u, = ctx.saved_tensors
        N, D, H, W, _ = u.shape
        spacing, normalize, norm = ctx.cfg
        gl = gloss.detach().reshape(1).to(torch.float32).contiguous()
        du = torch.empty_like(u)
        call('da_gradloss_bwd', ptr(u), ptr(gl), ptr(du), N, D, H, W, ctypes_float3(spacing), 1 if normalize else 0, norm, stream())
        return ncdhw(du), None, None, None


def ctypes_float3(v):
    import ctypes
    arr = (ctypes.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))
    return ctypes.cast(arr, ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# non-differentiable utilities
# ------------------------------------------------------------------------------------------------
def one_hot(mask, n_classes):
    """lib/transforms.py:675-689 mask_to_one_hot for a B x 1 x ... index mask -> B x C x ... float (channels-last)."""
    nat.require_cuda(mask)
    if mask.shape[1] != 1:
        raise ValueError('mask must be B x 1 x ...')
    lab, nbytes = _labels(mask.reshape(mask.shape[0], -1))
    M = lab.numel()
    out = torch.empty((M, n_classes), dtype=torch.float32, device=mask.device)
    call('da_one_hot', ptr(lab), nbytes, ptr(out), M, n_classes, stream())
    spatial = list(mask.shape[2:])
    out = out.reshape([mask.shape[0]] + spatial + [n_classes])
    perm = [0, len(spatial) + 1] + list(range(1, len(spatial) + 1))
    return out.permute(perm)


def identity_grid(size, normalize=True, device='cuda'):
    """lib/utils.py:89-102 get_identity_transform: 3 x D x H x W."""
    D, H, W = int(size[0]), int(size[1]), int(size[2])
    out = torch.empty((3, D, H, W), dtype=torch.float32, device=device)
    call('da_identity_grid', ptr(out), D, H, W, 1 if normalize else 0, stream())
    return out


def argmax_dice_counts(logits, truth):
    """Eval path (models/segmentation.py:188-194): first-max argmax + exact integer overlap counts.
    Returns (counts[N][C][3] int64 = (|pred==c|, |truth==c|, |both|), pred uint8 N x D x H x W)."""
    a = ndhwc(logits)
    N, C = a.shape[0], a.shape[-1]
    V = a.numel() // (N * C)
    lab, nbytes = _labels(truth.reshape(N, -1))
    counts = torch.zeros((N, C, 3), dtype=torch.int64, device=a.device)
    pred = torch.empty((N, V), dtype=torch.uint8, device=a.device)
    call('da_argmax_dice_counts', ptr(a), ptr(lab), nbytes, N, V, C, ptr(counts), ptr(pred), stream())
    return counts, pred.reshape((N,) + tuple(a.shape[1:4]))


def label_overlap_counts(pred, truth, n_class):
    """integer overlap counts of two label maps (any shape with a leading batch axis).
    Returns counts[N][n_class][3] int64 = (|pred==c|, |truth==c|, |both|); labels outside [0, n_class) are ignored."""
    N = pred.shape[0]
    pl, pb = _labels(pred.reshape(N, -1))
    tl, tb = _labels(truth.reshape(N, -1))
    if pl.shape != tl.shape:
        raise ValueError('pred and truth must have the same number of voxels per sample')
    counts = torch.zeros((N, n_class, 3), dtype=torch.int64, device=pl.device)
    call('da_label_overlap_counts', ptr(pl), pb, ptr(tl), tb, N, pl.shape[1], n_class, ptr(counts), stream())
    return counts
