"""bf16 ACTIVATION STORAGE (BASELINE configs[4]; ops.set_activation_storage('bf16') + the bf16 matrix mode) against an oracle that rounds
where the product stores: oracle.nets.ACT_STORE_ROUND (+ K3_OPERAND_ROUND for the matrix operands).  The reference has no such mode; what
is pinned here is that the HIP path computes exactly "the reference's network with bf16 round-to-nearest-even at every stored tensor"."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

def dev():
    return torch.device('cuda:0')


def _bf16(t):
    return t.bfloat16().float()


@pytest.fixture()
def bf16_storage():
    from deepatlas_amd import ops
    pm = ops.set_matrix_precision('bf16')
    ps = ops.set_activation_storage('bf16')
    ops.bridged_calls.clear()
    yield ops
    ops.set_activation_storage(ps)
    ops.set_matrix_precision(pm)
    ops.BF16_FORCE_BRIDGE = False


def _seg_setup(spec_name, n_classes, shape, batch=2):
    from oracle import nets
    from deepatlas_amd.lib.network_factory import unets, get_network
    spec = getattr(nets, spec_name)
    shapes = nets.unet_param_shapes(1, n_classes, spec['encoders'], spec['decoders'])
    sd = nets.closed_form_fill(shapes, seed=1)
    cls = get_network('UNet_light') if spec_name == 'UNET_LIGHT' else unets.UNet_generator(
        encoders=spec['encoders'], decoders=spec['decoders'], act='LeakyReLU', maxpool=True, upsample=False, res=False)
    model = cls(in_channel=1, n_classes=n_classes, bias=True, BN=True)
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    x = nets.closed_form_volume((batch, 1) + shape, seed=2)
    y = nets.closed_form_labels((batch,) + shape, n_classes, seed=3)
    return model.to(dev()), sd, spec, x, y


def _device_seg_step(model, x, y, n_classes, fused_head):
    from deepatlas_amd.lib.loss import get_loss_function
    from deepatlas_amd.optim import FlatAdam
    crit = get_loss_function('dice')(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    model.train()
    model.lazy_head = bool(fused_head)
    opt.zero_grad()
    out = model(x.to(dev()))
    loss = crit(out, y.to(dev()).long())
    loss.backward()
    from deepatlas_amd import ops
    ops.join_side_stream()
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    logits = ops.materialize_logits(out).detach().float().cpu() if not fused_head else None
    return float(loss.item()), logits, grads


def _oracle_seg_step(sd, spec, x, y, n_classes, store_round, operand_round=True):
    from oracle import nets, steps
    sd = {k: v.clone() for k, v in sd.items()}
    nets.K3_OPERAND_ROUND = _bf16 if operand_round else None
    nets.ACT_STORE_ROUND = _bf16 if store_round else None
    try:
        loss, logits, grads = steps.seg_step(sd, steps.Adam(steps.trainable(sd)), x, y, spec, n_classes)
    finally:
        nets.K3_OPERAND_ROUND = None
        nets.ACT_STORE_ROUND = None
    return float(loss.item()), logits, grads


@pytest.mark.parametrize('spec_name,n_classes,shape', [('UNET_LIGHT', 32, (32, 32, 32)), ('UNET_TINY', 5, (16, 24, 32))])
def test_seg_step_bf16_storage_vs_rounding_oracle(bf16_storage, spec_name, n_classes, shape):
    ops = bf16_storage
    model, sd, spec, x, y = _seg_setup(spec_name, n_classes, shape)
    loss, logits, grads = _device_seg_step(model, x, y, n_classes, fused_head=False)
    o_loss, o_logits, o_grads = _oracle_seg_step(sd, spec, x, y, n_classes, store_round=True)
    p_loss, p_logits, p_grads = _oracle_seg_step(sd, spec, x, y, n_classes, store_round=False)
    e_logits, e_plain = rel_l2(logits.numpy(), o_logits.numpy()), rel_l2(logits.numpy(), p_logits.numpy())
    a = np.concatenate([grads[n].numpy().reshape(-1) for n in grads])
    b = np.concatenate([o_grads[n].numpy().reshape(-1) for n in grads])
    c = np.concatenate([p_grads[n].numpy().reshape(-1) for n in grads])
    e_g, e_gp = rel_l2(a, b), rel_l2(a, c)
    print('bf16 storage %s: loss %.6f oracle %.6f (plain %.6f); logits rel-l2 %.2e (oracle without storage rounding %.2e); grads %.2e (%.2e); bridged %r'
          % (spec_name, loss, o_loss, p_loss, e_logits, e_plain, e_g, e_gp, dict(ops.bridged_calls)))
    # a bf16 rounding decision flips wherever the two fp32 computations straddle a rounding boundary (a 2^-9 relative step on that element);
    # those flips, not the arithmetic, set the floor of this comparison
    assert abs(loss - o_loss) < 2e-4 * max(1.0, abs(o_loss))
    assert e_logits < 4e-3, e_logits
    assert e_plain > 2 * e_logits, (e_plain, e_logits)          # the emulation is what makes them agree
    assert e_g < 5e-2, e_g


def test_bf16_twins_equal_the_conversion_route(bf16_storage):
    """Every `_bf16` twin against (convert -> fp32 entry -> convert) at network level: the same training step with the twins and with
    ops.BF16_FORCE_BRIDGE must produce the same loss and gradients (same kernels on the same values; only the staging differs)."""
    ops = bf16_storage
    res = []
    for force in (False, True):
        ops.BF16_FORCE_BRIDGE = force
        ops.bridged_calls.clear()
        model, sd, spec, x, y = _seg_setup('UNET_LIGHT', 32, (32, 32, 32))
        res.append(_device_seg_step(model, x, y, 32, fused_head=True) + (dict(ops.bridged_calls),))
    ops.BF16_FORCE_BRIDGE = False
    (l0, _, g0, b0), (l1, _, g1, b1) = res
    assert sum(b1.values()) > sum(b0.values()) + 20, (b0, b1)          # the forced run really took the conversion route
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l1)), (l0, l1)
    a = np.concatenate([g0[n].numpy().reshape(-1) for n in g0])
    b = np.concatenate([g1[n].numpy().reshape(-1) for n in g0])
    assert rel_l2(a, b) < 1e-5, rel_l2(a, b)


def test_bf16_storage_tensors_really_are_bf16(bf16_storage):
    """The activations between the layers are bf16 tensors (half the bytes), the logits and every parameter gradient fp32."""
    model, sd, spec, x, y = _seg_setup('UNET_LIGHT', 32, (32, 32, 32))
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, name=n: seen.__setitem__(name, out.dtype if torch.is_tensor(out) else None))
             for n, m in model.named_modules() if n.startswith('encoders.') and n.count('.') == 2]
    model.train()
    out = model(x.to(dev()))
    for h in hooks:
        h.remove()
    assert out.dtype == torch.float32
    dts = [v for v in seen.values() if v is not None]
    assert dts and all(v == torch.bfloat16 for v in dts), seen
