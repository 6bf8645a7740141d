"""Oracle: functional restatement of the reference's segmentation / registration nets.

Test infrastructure only (see oracle/__init__.py).  Parameters are plain dicts keyed by
the reference's ``state_dict`` names, so the same dict drives the reference import
(make_golden.py), this oracle, and the HIP modules under test.

Cites are into /root/reference.
"""
import math
import torch
import torch.nn.functional as F

# lib/network_factory/__init__.py:12-15  -- the 'UNet_light' spec
UNET_LIGHT = dict(encoders=[(8, 16), (16, 16, 32), (32, 32, 64), (64, 64, 64)],
                  decoders=[(64, 64, 64), (64, 32, 32), (32, 16, 16)],
                  slope=0.01)
# reduced-width spec used for small fixtures (same generator, smaller tuples)
UNET_TINY = dict(encoders=[(4, 8), (8, 8, 8), (8, 8, 16), (16, 16, 16)],
                 decoders=[(16, 16, 16), (16, 8, 8), (8, 8, 8)],
                 slope=0.01)
# UNet_generator option fixtures (unets.py:230-237,264,275): strided-conv down-samplers + trilinear up-samplers,
# and the residual variant (channel counts chosen so that `enc(x) + x` / `dec(.) + x` are well formed; the head maps to 16 classes)
UNET_OPT = dict(encoders=[(16, 16), (16, 16, 16), (32, 32, 32)], decoders=[(32, 16, 16), (16, 16, 16)], slope=0.0,
                maxpool=False, upsample=True, res=False)
UNET_RES = dict(encoders=[(16, 16), (16, 16, 16)], decoders=[(16, 16, 16)], slope=0.0, maxpool=True, upsample=False, res=True)

# lib/network_factory/voxel_morph.py:29-30
VM_ENC = (16, 32, 32, 32, 32)
VM_DEC = (32, 32, 32, 8, 8)


# ----------------------------------------------------------------------------------------
# parameter construction (names/shapes follow the reference's state_dict [probed])
# ----------------------------------------------------------------------------------------
def unet_param_shapes(in_channel, n_classes, encoders, decoders, bias=True, BN=True, maxpool=True, upsample=False):
    """Ordered {name: shape} as produced by UNet_generator (unets.py:199-252); maxpool=False adds the strided-conv
    `down_samplers.<i>` (unets.py:231-233), upsample=True removes the `up_samplers` parameters (unets.py:235-237)."""
    shapes = {}

    def conv_block(prefix, cin, cout, conv_name='conv', transposed=False, k=3):
        w = (cin, cout, k, k, k) if transposed else (cout, cin, k, k, k)
        shapes[f'{prefix}.{conv_name}.weight'] = w
        if bias:
            shapes[f'{prefix}.{conv_name}.bias'] = (cout,)
        if BN:
            shapes[f'{prefix}.BN.weight'] = (cout,)
            shapes[f'{prefix}.BN.bias'] = (cout,)
            shapes[f'{prefix}.BN.running_mean'] = (cout,)
            shapes[f'{prefix}.BN.running_var'] = (cout,)
            shapes[f'{prefix}.BN.num_batches_tracked'] = ()

    enc_last = None
    for i, enc in enumerate(encoders):
        if i == 0:
            enc = (in_channel,) + tuple(enc)
        for k in range(len(enc) - 1):
            conv_block(f'encoders.{i}.{k}', enc[k], enc[k + 1])
        enc_last = enc
    up, dec_blocks = {}, {}
    # unets.py:232-252: note `len(enc) - 1` uses the leaked loop variable (quirk, a6)
    nconv = len(enc_last) - 1
    dec_shapes_order = []
    for i, dec in enumerate(decoders):
        cin_up = encoders[-1][-1] if i == 0 else decoders[i - 1][-1]
        dec_shapes_order.append(('up', i, cin_up, dec[0]))
        full = (encoders[-(i + 2)][-1] + dec[0],) + tuple(dec[1:])
        for k in range(nconv):
            dec_shapes_order.append(('conv', i, k, full[k], full[k + 1]))
        if i == len(decoders) - 1:
            dec_shapes_order.append(('head', i, nconv, full[-1], n_classes))
    # registration order in the reference: encoders, decoders(ModuleList via add_module), down_samplers, up_samplers
    for item in dec_shapes_order:
        if item[0] == 'conv':
            _, i, k, cin, cout = item
            conv_block(f'decoders.decBlock{i}.{k}', cin, cout)
        elif item[0] == 'head':
            _, i, k, cin, cout = item
            shapes[f'decoders.decBlock{i}.{k}.weight'] = (cout, cin, 1, 1, 1)
            if bias:
                shapes[f'decoders.decBlock{i}.{k}.bias'] = (cout,)
    if not maxpool:
        for i in range(len(encoders) - 1):
            shapes[f'down_samplers.{i}.weight'] = (encoders[i + 1][0], encoders[i][-1], 2, 2, 2)
            if bias:
                shapes[f'down_samplers.{i}.bias'] = (encoders[i + 1][0],)
    for item in dec_shapes_order:
        if item[0] == 'up' and not upsample:
            _, i, cin, cout = item
            conv_block(f'up_samplers.{i}', cin, cout, conv_name='deconv', transposed=True, k=2)
    return shapes


def voxelmorph_param_shapes(input_channel=2, output_channel=3, enc=VM_ENC, dec=VM_DEC):
    """voxel_morph.py:43-57."""
    shapes = {}
    for i in range(len(enc)):
        cin = input_channel if i == 0 else enc[i - 1]
        shapes[f'encoders.{i}.conv.weight'] = (enc[i], cin, 3, 3, 3)
        shapes[f'encoders.{i}.conv.bias'] = (enc[i],)
    for i in range(len(dec)):
        if i == 0:
            cin = enc[-1]
        elif i < 4:
            cin = dec[i - 1] + enc[4 - i]
        else:
            cin = dec[i - 1]
        shapes[f'decoders.{i}.conv.weight'] = (dec[i], cin, 3, 3, 3)
        shapes[f'decoders.{i}.conv.bias'] = (dec[i],)
    shapes['flow.weight'] = (output_channel, dec[-1] + enc[0], 3, 3, 3)
    shapes['flow.bias'] = (output_channel,)
    return shapes


def closed_form_fill(shapes, seed=0, dtype=torch.float32):
    """Deterministic, RNG-free parameter fill (SURVEY.md §8c golden-vector plan).

    Conv/deconv weights: a*sin(b*i + phase) with a = xavier-like scale; biases small non-zero
    (so bias handling is pinned); BN weight ~1+0.1 sin, BN bias 0.05 cos, running stats (0,1).
    """
    out = {}
    for li, (name, shp) in enumerate(shapes.items()):
        n = int(math.prod(shp)) if len(shp) else 1
        i = torch.arange(n, dtype=torch.float64)
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith('running_mean'):
            out[name] = torch.zeros(shp, dtype=dtype)
        elif name.endswith('running_var'):
            out[name] = torch.ones(shp, dtype=dtype)
        elif '.BN.weight' in name:
            out[name] = (1.0 + 0.1 * torch.sin(0.7 * i + li + seed)).to(dtype).reshape(shp)
        elif '.BN.bias' in name:
            out[name] = (0.05 * torch.cos(1.3 * i + li + seed)).to(dtype).reshape(shp)
        elif name.endswith('bias'):
            out[name] = (0.02 * torch.sin(2.1 * i + 0.5 * li + seed)).to(dtype).reshape(shp)
        else:  # conv / deconv weight
            recf = int(math.prod(shp[2:]))
            fan = (shp[0] + shp[1]) * recf
            a = math.sqrt(2.0 / fan) * 1.7
            out[name] = (a * torch.sin(0.37 * i * (1 + 0.01 * li) + 1.1 * li + seed)).to(dtype).reshape(shp)
    return out


def closed_form_volume(shape, seed=0, dtype=torch.float32):
    """Deterministic image in [0,1] with D!=H!=W structure."""
    n = int(math.prod(shape))
    i = torch.arange(n, dtype=torch.float64)
    v = 0.5 + 0.5 * torch.sin(0.011 * i + 0.3 * seed) * torch.cos(0.00073 * i * (seed + 1) + seed)
    return v.to(dtype).reshape(shape)


def closed_form_labels(shape, n_classes, seed=0):
    """Blocky label map (N,D,H,W) uint8 in {0..n_classes-1}."""
    N, D, H, W = shape
    z = torch.arange(D).view(1, D, 1, 1)
    y = torch.arange(H).view(1, 1, H, 1)
    x = torch.arange(W).view(1, 1, 1, W)
    b = torch.arange(N).view(N, 1, 1, 1)
    lab = ((z // 3) * 7 + (y // 4) * 3 + (x // 5) + b * 5 + seed) % n_classes
    return lab.to(torch.uint8)


# ----------------------------------------------------------------------------------------
# forward restatements
# ----------------------------------------------------------------------------------------
# Optional emulation of the product's bf16 matrix mode (not a reference feature: train_seg.py is fp32-only).  When set to a
# callable, the input and the weight of every 3x3x3 convolution that the product runs on the matrix cores (Cin % 8 == 0,
# Cout >= 8, Cout % 4 == 0) are passed through it before the fp32 convolution, e.g. lambda t: t.bfloat16().float().
K3_OPERAND_ROUND = None

# Optional emulation of the product's bf16 ACTIVATION STORAGE (ops.set_activation_storage('bf16'); not a reference feature).  When set to a
# callable (e.g. lambda t: t.bfloat16().float()) every tensor the product stores between two layers -- convolution / transposed-convolution
# outputs and BatchNorm + activation outputs with >= 8 channels -- passes through it, and so does every gradient tensor the product stores
# (the gradient arriving at each consumer's input, and the summed gradient of a stored tensor).  BatchNorm then follows the product's
# arithmetic: batch statistics from the fp32 convolution result BEFORE it is rounded (they are accumulated in the convolution's epilogue;
# layers with <= 4 input channels run their statistics as a pass over the stored tensor), normalisation and backward on the ROUNDED tensor.
ACT_STORE_ROUND = None


class _StoreRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ACT_STORE_ROUND(x)

    @staticmethod
    def backward(ctx, g):
        return ACT_STORE_ROUND(g)


def _store(t, channels=None):
    """A stored tensor / a consumer's input edge under ACT_STORE_ROUND (identity otherwise; thin tensors -- the 1- or 2-channel network
    inputs, the 3-channel displacement field -- stay fp32 in the product)."""
    c = t.shape[1] if channels is None else channels
    if ACT_STORE_ROUND is None or not (c >= 8 and c % 4 == 0):
        return t
    return _StoreRound.apply(t)


class _BNGivenStats(torch.autograd.Function):
    """(x - mean) * rstd * gamma + beta with batch statistics computed elsewhere; the backward is the standard BatchNorm backward evaluated
    on x (norm_act.hip: bn_act_bwd_apply_kernel)."""

    @staticmethod
    def forward(ctx, x, mean, rstd, gamma, beta):
        sh = (1, -1, 1, 1, 1)
        xh = (x - mean.view(sh)) * rstd.view(sh)
        ctx.save_for_backward(xh, rstd, gamma)
        return xh * gamma.view(sh) + beta.view(sh)

    @staticmethod
    def backward(ctx, g):
        xh, rstd, gamma = ctx.saved_tensors
        sh = (1, -1, 1, 1, 1)
        dims = (0, 2, 3, 4)
        M = g.numel() // g.shape[1]
        s1 = g.sum(dims); s2 = (g * xh).sum(dims)
        dx = (gamma * rstd).view(sh) * (g - (s1 / M).view(sh) - xh * (s2 / M).view(sh))
        return dx, None, None, s2, s1


def _bn_train_stored(y, sd, prefix, momentum, eps, stats_from_stored):
    """Training-mode BatchNorm of a convolution result under ACT_STORE_ROUND (see there); returns the normalised ROUNDED tensor."""
    yr = _store(y)
    src = (yr if stats_from_stored else y).detach()
    dims = (0, 2, 3, 4)
    M = src.numel() // src.shape[1]
    mean = src.double().mean(dims)
    var = (src.double() ** 2).mean(dims) - mean ** 2
    var = var.clamp_min(0.0)
    rstd = (1.0 / torch.sqrt(var + eps)).float()
    rm, rv = sd[f'{prefix}.BN.running_mean'], sd[f'{prefix}.BN.running_var']
    with torch.no_grad():
        rm.mul_(1 - momentum).add_(momentum * mean.float())
        rv.mul_(1 - momentum).add_(momentum * (var * (M / max(M - 1, 1))).float())
    return _BNGivenStats.apply(yr, mean.float(), rstd, sd[f'{prefix}.BN.weight'], sd[f'{prefix}.BN.bias'])


def _conv_bn_act(x, sd, prefix, slope, training, conv_name='conv', stride=1, padding=1, momentum=0.1, eps=1e-5):
    """unets.py:24-39 convBlock: Conv3d -> BatchNorm3d -> LeakyReLU(0.01)."""
    w = sd[f'{prefix}.{conv_name}.weight']
    b = sd.get(f'{prefix}.{conv_name}.bias')
    if K3_OPERAND_ROUND is not None and tuple(w.shape[2:]) == (3, 3, 3) and w.shape[1] % 8 == 0 and w.shape[0] >= 8 and w.shape[0] % 4 == 0:
        x, w = K3_OPERAND_ROUND(x), K3_OPERAND_ROUND(w)
    y = F.conv3d(_store(x), w, b, stride=stride, padding=padding)
    if f'{prefix}.BN.weight' in sd:
        if ACT_STORE_ROUND is not None and training:
            y = _bn_train_stored(y, sd, prefix, momentum, eps, stats_from_stored=w.shape[1] <= 4)
        else:
            y = F.batch_norm(_store(y), sd[f'{prefix}.BN.running_mean'], sd[f'{prefix}.BN.running_var'],
                             sd[f'{prefix}.BN.weight'], sd[f'{prefix}.BN.bias'], training, momentum, eps)
        if training:
            sd[f'{prefix}.BN.num_batches_tracked'] += 1
    return _store(F.leaky_relu(y, slope) if slope else F.relu(y))


def _deconv_bn_act(x, sd, prefix, slope, training, momentum=0.1, eps=1e-5):
    """unets.py:42-58 deconvBlock: ConvTranspose3d(k2,s2) -> BN -> act."""
    w = sd[f'{prefix}.deconv.weight']
    b = sd.get(f'{prefix}.deconv.bias')
    y = F.conv_transpose3d(_store(x), w, b, stride=2)
    if f'{prefix}.BN.weight' in sd:
        if ACT_STORE_ROUND is not None and training:
            y = _bn_train_stored(y, sd, prefix, momentum, eps, stats_from_stored=False)
        else:
            y = F.batch_norm(_store(y), sd[f'{prefix}.BN.running_mean'], sd[f'{prefix}.BN.running_var'],
                             sd[f'{prefix}.BN.weight'], sd[f'{prefix}.BN.bias'], training, momentum, eps)
        if training:
            sd[f'{prefix}.BN.num_batches_tracked'] += 1
    return _store(F.leaky_relu(y, slope) if slope else F.relu(y))


def unet_forward(sd, x, spec, training=True):
    """UNetTemplate.forward, unets.py:259-278 (maxpool=True, upsample=False, res=False).

    sd: dict of tensors with the reference state_dict keys (BN running stats are updated
    in place when training, like nn.BatchNorm3d).  Returns raw logits N x n_classes x D x H x W.
    """
    encoders, decoders, slope = spec['encoders'], spec['decoders'], spec['slope']
    maxpool, upsample, res = spec.get('maxpool', True), spec.get('upsample', False), spec.get('res', False)
    levels = len(encoders)
    temp = []
    for i, enc in enumerate(encoders):
        nconv = len(enc) - (0 if i == 0 else 1)
        y = x
        for k in range(nconv):
            y = _conv_bn_act(y, sd, f'encoders.{i}.{k}', slope, training)
        x = (y + x) if res else y                                    # unets.py:264
        if i < levels - 1:
            temp.append(x)
            if maxpool:
                x = F.max_pool3d(_store(x), 2)                       # unets.py:230,267
            else:                                                    # strided conv k2 s2 p0, no BN / act (unets.py:231-233)
                x = F.conv3d(x, sd[f'down_samplers.{i}.weight'], sd.get(f'down_samplers.{i}.bias'), stride=2)
    nconv_dec = len(encoders[-1]) - (0 if levels == 1 else 1)        # leaked `enc` quirk unets.py:247
    for j in range(len(decoders)):
        if upsample:                                                  # nn.Upsample(scale_factor=2, mode="trilinear") (:236)
            x = F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=False)
        else:
            x = _deconv_bn_act(x, sd, f'up_samplers.{j}', slope, training)
        y = torch.cat((_store(x), _store(temp.pop())), dim=1)         # up-sampled first, skip second (:275)
        for k in range(nconv_dec):
            y = _conv_bn_act(y, sd, f'decoders.decBlock{j}.{k}', slope, training)
        if j == len(decoders) - 1:                                    # 1x1x1 head, no BN / act (:249-250)
            y = F.conv3d(_store(y), sd[f'decoders.decBlock{j}.{nconv_dec}.weight'],
                         sd.get(f'decoders.decBlock{j}.{nconv_dec}.bias'))
        x = (y + x) if res else y                                     # (:275)
    return x


def identity_transform(size, dtype=torch.float32):
    """lib/utils.py:89-102 get_identity_transform(normalize=True): 3 x D x H x W with channel
    0 = W-axis coord, 1 = H-axis, 2 = D-axis, each k/(size-1)*2-1."""
    D, H, W = size
    zz = (torch.arange(D, dtype=torch.float32) / (D - 1) * 2.0 - 1).view(D, 1, 1).expand(D, H, W)
    yy = (torch.arange(H, dtype=torch.float32) / (H - 1) * 2.0 - 1).view(1, H, 1).expand(D, H, W)
    xx = (torch.arange(W, dtype=torch.float32) / (W - 1) * 2.0 - 1).view(1, 1, W).expand(D, H, W)
    return torch.stack([xx, yy, zz]).to(dtype)


def warp_trilinear(source, deform):
    """voxel_morph.py:90-91: grid_sample(bilinear, zeros, align_corners=True) with the
    deformation field N x 3 x D x H x W permuted to N x D x H x W x 3 (x,y,z order)."""
    return F.grid_sample(source, deform.permute(0, 2, 3, 4, 1), mode='bilinear',
                         padding_mode='zeros', align_corners=True)


def _vm_conv(x, sd, prefix, stride):
    """modules.py:28-62 convBlock for the reg net: Conv3d(k3,p1,stride,bias) -> ReLU (no BN)."""
    w = sd[f'{prefix}.conv.weight']
    if K3_OPERAND_ROUND is not None and w.shape[1] % 8 == 0 and w.shape[0] >= 8 and w.shape[0] % 4 == 0:      # (see K3_OPERAND_ROUND)
        x, w = K3_OPERAND_ROUND(x), K3_OPERAND_ROUND(w)
    return _store(F.relu(F.conv3d(_store(x), w, sd[f'{prefix}.conv.bias'], stride=stride, padding=1)))


def voxelmorph_forward(sd, source, target):
    """VoxelMorphCVPR2018.forward, voxel_morph.py:62-92.  Returns (disp, warped_source, deform)."""
    e1 = _vm_conv(torch.cat((source, target), dim=1), sd, 'encoders.0', 1)
    e2 = _vm_conv(e1, sd, 'encoders.1', 2)
    e3 = _vm_conv(e2, sd, 'encoders.2', 2)
    e4 = _vm_conv(e3, sd, 'encoders.3', 2)
    e5 = _vm_conv(e4, sd, 'encoders.4', 2)
    d1 = _vm_conv(F.interpolate(e5, size=e4.shape[2:]), sd, 'decoders.0', 1)            # nearest (default)
    d2 = _vm_conv(F.interpolate(torch.cat((d1, e4), 1), size=e3.shape[2:]), sd, 'decoders.1', 1)
    d3 = _vm_conv(F.interpolate(torch.cat((d2, e3), 1), size=e2.shape[2:]), sd, 'decoders.2', 1)
    d4 = _vm_conv(torch.cat((d3, e2), 1), sd, 'decoders.3', 1)
    d5 = _vm_conv(F.interpolate(d4, size=e1.shape[2:]), sd, 'decoders.4', 1)
    disp = F.conv3d(_store(torch.cat((d5, e1), 1)), sd['flow.weight'], sd['flow.bias'], padding=1)
    deform = disp + identity_transform(source.shape[2:], disp.dtype)
    warped = warp_trilinear(source, deform)
    return disp, warped, deform


# ---- the fixed `UNet` (lib/network_factory/unets.py:70-179) -------------------------------------------------
UNET_FULL_ENC = [('ec0', None, 32), ('ec1', 32, 64), ('ec2', 64, 64), ('ec3', 64, 128), ('ec4', 128, 128), ('ec5', 128, 256),
                 ('ec6', 256, 256), ('ec7', 256, 512)]                                    # unets.py:75-82 (Conv3d k3 p1)
UNET_FULL_DEC = [('dc9', 512, 512, 2), ('dc8', 256 + 512, 256, 3), ('dc7', 256, 256, 3), ('dc6', 256, 256, 2),
                 ('dc5', 128 + 256, 128, 3), ('dc4', 128, 128, 3), ('dc3', 128, 128, 2), ('dc2', 64 + 128, 64, 3),
                 ('dc1', 64, 64, 3)]                                                      # unets.py:88-96 (ConvTranspose3d)


def unet_full_param_shapes(in_channel, n_classes, bias=True, BN=True):
    """state_dict keys / shapes of `UNet(in_channel, n_classes, bias, BN)`: nn.Sequential children are positional
    ('<name>.0' conv or transposed conv, '<name>.1' BatchNorm3d)."""
    shapes = {}

    def bn(prefix, c):
        if BN:
            shapes[prefix + '.1.weight'] = (c,); shapes[prefix + '.1.bias'] = (c,)
            shapes[prefix + '.1.running_mean'] = (c,); shapes[prefix + '.1.running_var'] = (c,)
            shapes[prefix + '.1.num_batches_tracked'] = ()
    for name, cin, cout in UNET_FULL_ENC:
        cin = in_channel if cin is None else cin
        shapes[name + '.0.weight'] = (cout, cin, 3, 3, 3)
        if bias:
            shapes[name + '.0.bias'] = (cout,)
        bn(name, cout)
    for name, cin, cout, k in UNET_FULL_DEC:
        shapes[name + '.0.weight'] = (cin, cout, k, k, k)                                  # ConvTranspose3d: [Cin][Cout][k^3]
        if bias:
            shapes[name + '.0.bias'] = (cout,)
        bn(name, cout)
    shapes['dc0.weight'] = (n_classes, 64, 1, 1, 1)                                          # unets.py:98
    if bias:
        shapes['dc0.bias'] = (n_classes,)
    return shapes


def closed_form_fill_positional(shapes, seed=0, dtype=torch.float32):
    """closed_form_fill for positional nn.Sequential keys: a 1-D 'weight' is a BatchNorm weight, a 'bias' whose sibling
    'running_mean' exists is a BatchNorm bias; everything else as in closed_form_fill."""
    renamed = {}
    for name, shp in shapes.items():
        stem = name.rsplit('.', 1)[0]
        is_bn = (stem + '.running_mean') in shapes
        if is_bn and name.endswith(('.weight', '.bias')):
            renamed[stem + '.BN.' + name.rsplit('.', 1)[1]] = (name, shp)
        else:
            renamed[name] = (name, shp)
    filled = closed_form_fill({k: v[1] for k, v in renamed.items()}, seed=seed, dtype=dtype)
    return {renamed[k][0]: v for k, v in filled.items()}


def unet_full_forward(sd, x, training=True, momentum=0.1, eps=1e-5):
    """UNet.forward, unets.py:141-179.  sd: reference state_dict keys; BN running stats updated in place when training."""
    def block(x, name, transposed, k=3):
        w, b = sd[name + '.0.weight'], sd.get(name + '.0.bias')
        if not transposed:
            y = F.conv3d(x, w, b, stride=1, padding=1)                                       # encoder :113-124
        elif k == 2:
            y = F.conv_transpose3d(x, w, b, stride=2)                                        # decoder k2 s2 :88,91,94
        else:
            y = F.conv_transpose3d(x, w, b, stride=1, padding=1)                             # decoder k3 s1 p1 :89-90,...
        if (name + '.1.weight') in sd:
            y = F.batch_norm(y, sd[name + '.1.running_mean'], sd[name + '.1.running_var'], sd[name + '.1.weight'],
                             sd[name + '.1.bias'], training, momentum, eps)
            if training:
                sd[name + '.1.num_batches_tracked'] += 1
        return F.relu(y)
    syn0 = block(block(x, 'ec0', False), 'ec1', False)
    syn1 = block(block(F.max_pool3d(syn0, 2), 'ec2', False), 'ec3', False)
    syn2 = block(block(F.max_pool3d(syn1, 2), 'ec4', False), 'ec5', False)
    e7 = block(block(F.max_pool3d(syn2, 2), 'ec6', False), 'ec7', False)
    d9 = torch.cat((block(e7, 'dc9', True, 2), syn2), dim=1)
    d7 = block(block(d9, 'dc8', True), 'dc7', True)
    d6 = torch.cat((block(d7, 'dc6', True, 2), syn1), dim=1)
    d4 = block(block(d6, 'dc5', True), 'dc4', True)
    d3 = torch.cat((block(d4, 'dc3', True, 2), syn0), dim=1)
    d1 = block(block(d3, 'dc2', True), 'dc1', True)
    return F.conv3d(d1, sd['dc0.weight'], sd.get('dc0.bias'))
