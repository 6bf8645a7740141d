"""ctypes binding of libdeepatlas_hip.so (C ABI declared in include/deepatlas_hip.h).

PyTorch is used here only as plumbing: device memory (torch.empty), the current HIP stream and
(elsewhere) torch.distributed.  There is NO fallback: if the shared library is missing the import of any
op fails loudly with the build command.
"""
import ctypes
import os
from ctypes import c_int, c_float, c_longlong, c_size_t, c_void_p, c_char_p, POINTER

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DA_LIB') or os.path.join(_HERE, 'csrc', 'libdeepatlas_hip.so')     # DA_LIB: A/B builds of the same ABI (tools/)

P = c_void_p
I = c_int
F = c_float
LL = c_longlong
SZ = c_size_t

# name -> (restype, argtypes); mirrors include/deepatlas_hip.h one to one
SIGNATURES = {
    'da_version': (I, []),
    'da_device_info': (I, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t), c_char_p, I]),
    'da_w_oik_to_tio': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_oik': (I, [P, P, I, I, I, P]),
    'da_w_iok_to_tio': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_iok': (I, [P, P, I, I, I, P]),
    'da_w_iok_flip_to_tio': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_iok_flip': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_oik_acc': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_iok_acc': (I, [P, P, I, I, I, P]),
    'da_w_tio_to_iok_flip_acc': (I, [P, P, I, I, I, P]),
    'da_conv3d_k3_ws_bytes': (SZ, [I, I, I, I, I, I, I]),
    'da_conv3d_k3_pack_bytes': (SZ, [I, I, I, I, I, I]),
    'da_conv3d_k3_dgrad_bst': (I, [P, P, P, I, P, I, I, I, I, I, I, P, P, F, P, I, P, P, SZ, P]),
    'da_conv3d_k3_prepack': (I, [P, I, I, I, I, I, I, I, I, P, SZ, P, SZ, P, P]),
    'da_conv3d_k3_use_prepacked': (None, [P, P, SZ, P, SZ]),
    'da_conv3d_k3_prepack_any': (I, [P, I, I, I, I, I, I, I, I, I, I, P, SZ, P, P, P, P, SZ, P]),
    'da_conv3d_k3_use_prepacked_any': (None, [P, P, SZ, I]),
    'da_conv3d_k3_prepack_many': (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    'da_conv3d_k3_fwd': (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, F, P, SZ, P]),
    'da_conv3d_k3_fwd_bnstats': (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, P, I, POINTER(c_int), P, SZ, P]),
    'da_conv3d_k3_dgrad': (I, [P, P, P, I, P, I, I, I, I, I, I, I, P, SZ, P]),
    'da_conv3d_k3_wgrad': (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_conv3d_k3_fwd_pro': (I, [P, I, P, P, F, P, I, P, P, F, P, P, P, I, I, I, I, I, F, P, I, POINTER(c_int), P, SZ, P]),
    'da_conv3d_k3_wgrad_pro': (I, [P, I, P, P, F, P, I, P, P, F, P, P, I, I, I, I, I, P, SZ, P]),
    'da_upconv3d_k3_supported': (I, [I, I, I]),
    'da_upconv3d_k3_ws_bytes': (SZ, [I, I, I, I, I, I]),
    'da_upconv3d_k3_fwd': (I, [P, I, P, I, P, P, P, I, I, I, I, I, F, P, SZ, P]),
    'da_upconv3d_k3_dgrad': (I, [P, P, P, I, P, I, I, I, I, I, I, P, SZ, P]),
    'da_upconv3d_k3_wgrad': (I, [P, I, P, I, P, P, I, I, I, I, I, P, SZ, P]),
    'da_set_conv_direct': (I, [I]),
    'da_set_matrix_bf16': (I, [I]),
    'da_set_matrix_mode': (I, [I]),
    'da_pointwise_ws_bytes': (SZ, [I, I, I]),
    'da_conv1x1_fwd': (I, [P, P, P, P, LL, I, I, P, SZ, P]),
    'da_conv1x1_fwd_pro': (I, [P, P, P, F, P, P, P, LL, I, I, P, SZ, P]),
    'da_conv1x1_wgrad_pro': (I, [P, P, P, F, P, P, P, LL, I, I, P, SZ, P]),
    'da_conv1x1_dgrad': (I, [P, P, P, LL, I, I, P, SZ, P]),
    'da_conv1x1_wgrad_ws_bytes': (SZ, [LL, I, I]),
    'da_conv1x1_wgrad': (I, [P, P, P, P, LL, I, I, P, SZ, P]),
    'da_deconv_k2s2_fwd': (I, [P, P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_deconv_k2s2_fwd_bnstats': (I, [P, P, P, P, I, I, I, I, I, I, P, I, POINTER(c_int), P, SZ, P]),
    'da_deconv_k2s2_dgrad': (I, [P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_deconv_k2s2_bn_bwd_ws_bytes': (SZ, [I, I, I, I, I, I]),
    'da_deconv_k2s2_bn_bwd': (I, [P, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, I, I, I, I, P, I, P, SZ, P]),
    'da_deconv_k2s2_wgrad_ws_bytes': (SZ, [I, I, I, I, I, I]),
    'da_deconv_k2s2_wgrad': (I, [P, P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_bn_ws_bytes': (SZ, [LL, I]),
    'da_bn_train_stats': (I, [P, LL, I, P, P, F, F, P, P, P, P, P, P, P, SZ, P]),
    'da_bn_train_stats_from_partials': (I, [P, I, LL, I, P, P, F, F, P, P, P, P, P, P, P]),
    'da_bn_eval_affine': (I, [P, P, P, P, F, I, P, P, P, P, P]),
    'da_bn_act_fwd': (I, [P, P, P, F, P, LL, I, P]),
    'da_bn_act_bwd': (I, [P, P, P, P, P, P, P, F, I, P, P, P, LL, I, P, SZ, P]),
    'da_bn_act_bwd_dbias': (I, [P, P, P, P, P, P, F, I, P, P, P, P, LL, I, P, SZ, P]),
    'da_bn_act_bwd_dbias_pre': (I, [P, P, P, P, P, P, F, I, P, P, P, P, LL, I, P, I, P, SZ, P]),
    'da_act_bwd': (I, [P, P, F, P, LL, P]),
    'da_act_bwd_add_dbias': (I, [P, P, P, F, P, P, LL, I, P, SZ, P]),
    'da_colsum': (I, [P, LL, I, P, P, SZ, P]),
    'da_act_bwd_add_partial': (I, [P, P, P, F, P, LL, I, P, SZ, POINTER(c_int), P]),
    'da_colsum_finish': (I, [P, I, I, P, I, P]),
    'da_maxpool2_fwd': (I, [P, P, I, I, I, I, I, P]),
    'da_maxpool2_fwd_pro': (I, [P, P, P, F, P, P, I, I, I, I, I, P]),
    'da_maxpool2_bwd': (I, [P, P, P, I, I, I, I, I, P]),
    'da_maxpool2_bwd_add': (I, [P, P, P, P, I, I, I, I, I, P]),
    'da_maxpool2_bwd_bst': (I, [P, P, P, P, I, I, I, I, I, P, F, P, I, P, P]),
    'da_upsample_nearest_fwd': (I, [P, P, I, I, I, I, I, I, I, I, P]),
    'da_upsample_nearest_bwd': (I, [P, P, I, I, I, I, I, I, I, I, P]),
    'da_warp_fwd': (I, [P, P, P, P, I, I, I, I, I, P]),
    'da_warp_bwd': (I, [P, P, P, P, P, I, I, I, I, I, P]),
    'da_label_warp_dice_ws_bytes': (SZ, [I, I]),
    'da_warp_dice_ws_bytes': (SZ, [I, I]),
    'da_warp_dice_fwd': (I, [P, P, P, I, I, I, I, I, I, I, I, F, P, P, P, SZ, P]),
    'da_softmax_dice_fwd': (I, [P, P, I, P, I, LL, I, I, I, F, P, P, P, SZ, P]),
    'da_label_warp_dice_fwd': (I, [P, I, P, I, P, I, I, I, I, I, I, I, F, P, P, P, SZ, P]),
    'da_label_warp_dice_bwd': (I, [P, I, P, I, P, P, P, P, I, I, I, I, I, P]),
    'da_warp_adjoint_labels': (I, [P, I, P, P, P, I, I, I, I, I, P]),
    'da_seg_anat_dlogits': (I, [P, P, I, P, P, P, P, P, P, P, I, LL, I, P]),
    'da_warp_bwd_dsrc_det_ws_bytes': (SZ, [I, I, I, I, I]),
    'da_warp_bwd_dsrc_det': (I, [P, P, P, I, I, I, I, I, P, SZ, P]),
    'da_identity_grid': (I, [P, I, I, I, I, P]),
    'da_warp_labels_fwd': (I, [P, I, P, P, I, I, I, I, I, P]),
    'da_warp_labels_bwd': (I, [P, P, I, P, P, I, I, I, I, I, P]),
    'da_dice_ws_bytes': (SZ, [I, LL, I]),
    'da_dice_fwd': (I, [P, P, I, P, I, LL, I, I, I, I, F, P, P, P, SZ, P]),
    'da_dice_bwd': (I, [P, P, I, P, P, P, P, I, LL, I, I, P]),
    'da_softmax_fwd': (I, [P, P, LL, I, P]),
    'da_softmax_bwd': (I, [P, P, P, LL, I, P]),
    'da_one_hot': (I, [P, I, P, LL, I, P]),
    'da_head_dice_ws_bytes': (SZ, [I, LL, I, I]),
    'da_head_dice_fwd': (I, [P, P, P, F, P, P, P, I, I, LL, I, I, I, I, F, P, P, P, SZ, P]),
    'da_head_dice_bwd': (I, [P, P, P, F, P, P, P, I, P, P, P, P, P, I, LL, I, I, P, SZ, P]),
    'da_head_dice_bwd_bst': (I, [P, P, P, F, P, P, P, P, I, P, P, P, P, P, I, LL, I, I, P, I, P, P, SZ, P]),
    'da_ncc_ws_bytes': (SZ, [I, LL]),
    'da_ncc_fwd': (I, [P, P, I, LL, P, P, P, SZ, P]),
    'da_ncc_bwd': (I, [P, P, P, P, P, P, I, LL, P]),
    'da_bending_ws_bytes': (SZ, [I, I, I, I]),
    'da_bending_fwd': (I, [P, I, I, I, I, P, I, I, P, P, SZ, P]),
    'da_bending_bwd': (I, [P, P, P, I, I, I, I, P, I, I, P]),
    'da_xent_ws_bytes': (SZ, []),
    'da_xent_fwd': (I, [P, P, I, P, P, LL, I, I, I, F, LL, I, P, P, P, SZ, P]),
    'da_xent_bwd': (I, [P, P, I, P, P, P, P, P, LL, I, I, I, F, LL, P]),
    'da_argmax_dice_counts': (I, [P, P, I, I, LL, I, P, P, P]),
    'da_label_overlap_counts': (I, [P, I, P, I, I, LL, I, P, P]),
    'da_conv_k2s2_fwd': (I, [P, P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_conv_k2s2_dgrad': (I, [P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_conv_k2s2_wgrad_ws_bytes': (SZ, [I, I, I, I, I, I]),
    'da_conv_k2s2_wgrad': (I, [P, P, P, P, I, I, I, I, I, I, P, SZ, P]),
    'da_upsample_trilinear2_fwd': (I, [P, P, I, I, I, I, I, P]),
    'da_upsample_trilinear2_bwd': (I, [P, P, I, I, I, I, I, P]),
    'da_clamp01_to_f32': (I, [P, I, P, LL, P]),
    'da_crop3d': (I, [P, P, I, LL, I, I, I, I, I, I, I, I, I, P]),
    'da_partition_tiles': (I, [P, P, I, I, I, I, P, P, P]),
    'da_assemble_tiles': (I, [P, P, I, I, I, I, P, P, I, P]),
    'da_synth_volume': (I, [P, P, I, I, I, I, I, I, F, ctypes.c_uint, I, P]),
    'da_lncc_ws_bytes': (SZ, [I, I, I, I, I, I, I]),
    'da_lncc_fwd': (I, [P, P, I, I, I, I, I, I, I, F, P, P, P, SZ, P]),
    'da_lncc_bwd': (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, SZ, P]),
    'da_gradloss_ws_bytes': (SZ, [I, I, I, I]),
    'da_gradloss_fwd': (I, [P, I, I, I, I, P, I, I, P, P, SZ, P]),
    'da_gradloss_bwd': (I, [P, P, P, I, I, I, I, P, I, I, P]),
    'da_adam_step': (I, [P, P, P, P, LL, F, F, F, F, I, F, P]),
    'da_adam_host_state': (I, [F, F, F, F, I, F, P]),
    'da_adam_step_dev': (I, [P, P, P, P, LL, P, P]),
}

# bf16 activation storage: `<name>_bf16` twins (include/deepatlas_hip.h, last section) have the argument list of `<name>`; the 3x3x3
# convolution twins take one more argument, the bit mask of their bf16 activation arguments
BF16_TWINS = ['da_bn_train_stats', 'da_bn_act_fwd', 'da_bn_act_bwd_dbias', 'da_act_bwd', 'da_act_bwd_add_dbias', 'da_act_bwd_add_partial', 'da_colsum',
              'da_maxpool2_fwd', 'da_maxpool2_fwd_pro', 'da_maxpool2_bwd', 'da_maxpool2_bwd_add', 'da_upsample_nearest_fwd', 'da_upsample_nearest_bwd',
              'da_deconv_k2s2_fwd', 'da_deconv_k2s2_fwd_bnstats', 'da_deconv_k2s2_dgrad', 'da_deconv_k2s2_wgrad',
              'da_conv1x1_fwd', 'da_conv1x1_fwd_pro', 'da_conv1x1_dgrad', 'da_conv1x1_wgrad', 'da_conv1x1_wgrad_pro', 'da_head_dice_fwd', 'da_head_dice_bwd']
BF16_MASKED_TWINS = ['da_conv3d_k3_fwd', 'da_conv3d_k3_fwd_bnstats', 'da_conv3d_k3_fwd_pro', 'da_conv3d_k3_wgrad_pro', 'da_conv3d_k3_dgrad', 'da_conv3d_k3_wgrad']
for _n in BF16_TWINS:
    SIGNATURES[_n + '_bf16'] = (SIGNATURES[_n][0], list(SIGNATURES[_n][1]))
for _n in BF16_MASKED_TWINS:
    SIGNATURES[_n + '_bf16'] = (SIGNATURES[_n][0], list(SIGNATURES[_n][1]) + [ctypes.c_uint])
SIGNATURES['da_cast_f32_to_bf16'] = (I, [P, P, LL, P])
SIGNATURES['da_cast_bf16_to_f32'] = (I, [P, P, LL, P])

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load libdeepatlas_hip.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise NativeError(
                "deepatlas_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for the hot path." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)            # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


_ERR = {-1: 'DA_ERR_BADARG', -2: 'DA_ERR_WS_SMALL', -3: 'DA_ERR_UNSUPPORTED'}


class CallProfiler:
    """Optional per-call timing with HIP events on the launch stream (torch's current stream is the stream every
    launcher is given).  key = (C-ABI name, tuple of the call's integer arguments = the layer shape)."""

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.records = {}

    def wants(self, name):
        return self.names is None or name in self.names

    def summary(self):
        """{key: (n_calls, total_ms)} -- call after torch.cuda.synchronize()."""
        out = {}
        for key, evs in self.records.items():
            out[key] = (len(evs), sum(a.elapsed_time(b) for a, b in evs))
        return out


profiler = None          # set to a CallProfiler to time calls
n_calls = 0              # C-ABI calls issued by this process (bench.py reports launches per step)


def call(name, *args):
    global n_calls
    n_calls += 1
    prof = profiler
    base = name[:-5] if name.endswith('_bf16') else name          # a bf16 twin is recorded under its entry's name, the flag in the key
    if prof is not None and prof.wants(base):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib(), name)(*args)
        b.record()
        key = (base, tuple(x for x in args if isinstance(x, int)) + (('bf16',) if base != name else ()))
        prof.records.setdefault(key, []).append((a, b))
    else:
        rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise NativeError('%s failed: %s' % (name, _ERR.get(rc, 'hipError_t %d' % rc)))


def call_supported(name, *args):
    """Like call(), for entry points that may decline a shape: returns False on DA_ERR_UNSUPPORTED (the caller then takes its
    documented alternative HIP entry), True on success; every other error raises."""
    try:
        call(name, *args)
        return True
    except NativeError as e:
        if 'DA_ERR_UNSUPPORTED' in str(e):
            return False
        raise


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("deepatlas_amd ops run only on the GPU (HIP kernels); got a CPU tensor. "
                              "The CPU restatement lives in oracle/ and is test infrastructure only.")


class _Workspace:
    """One growing scratch buffer per device and stream; kernels on one stream are ordered, so it is shared by all ops on it.
    While a HIP graph is being captured (graphs.GraphedStep sets `capture_tag`) the buffers are separate ones, allocated INSIDE the
    capture: they live in that graph's private memory pool and are kept until the owning GraphedStep is closed, so a replay never finds its
    scratch reallocated or released by someone else's larger request (or by the empty_cache() of a later capture)."""

    def __init__(self):
        self.buf = {}
        self.capture_tag = None

    def get(self, nbytes, device):
        nbytes = int(nbytes) + 256
        # one buffer per (device, stream): kernels on one stream are ordered, kernels on different streams are not
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
        if self.capture_tag is not None and torch.cuda.is_current_stream_capturing():
            key = key + (self.capture_tag, len([k for k in self.buf if k[:3] == key and len(k) == 5 and k[3] == self.capture_tag]))
            # (a larger request during the same capture gets ANOTHER buffer; the earlier one stays referenced by the nodes already captured)
            for k, b in self.buf.items():
                if len(k) == 5 and k[:4] == key[:4] and b.numel() >= nbytes:
                    return c_void_p(b.data_ptr()), c_size_t(b.numel())
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
            self.buf[key] = b
        return c_void_p(b.data_ptr()), c_size_t(b.numel())


    def release(self, tag):
        """Forget the buffers allocated under capture tag `tag` (graphs.GraphedStep.close())."""
        for k in [k for k in self.buf if len(k) == 5 and k[3] == tag]:
            del self.buf[k]


workspace = _Workspace()
