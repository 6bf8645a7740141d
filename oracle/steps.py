"""Oracle: training-step restatements (callers of the hot path).

Test infrastructure only (see oracle/__init__.py).  Cites are into /root/reference.
"""
import math
import torch
import torch.nn.functional as F

from . import nets, losses


def trainable(sd):
    """Names of tensors that are nn.Parameters in the reference (everything except BN buffers)."""
    return [k for k in sd if not (k.endswith('running_mean') or k.endswith('running_var')
                                  or k.endswith('num_batches_tracked'))]


class Adam:
    """torch.optim.Adam defaults as used at models/segmentation.py:91 (lr, betas=(0.9,0.999),
    eps=1e-8, no weight decay, no amsgrad), restated explicitly."""

    def __init__(self, names, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.names, self.lr, self.b1, self.b2, self.eps = list(names), lr, b1, b2, eps
        self.t = 0
        self.m, self.v = {}, {}

    @torch.no_grad()
    def step(self, sd, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for n in self.names:
            g = grads[n]
            if n not in self.m:
                self.m[n] = torch.zeros_like(g)
                self.v[n] = torch.zeros_like(g)
            self.m[n].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[n].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
            sd[n].addcdiv_(self.m[n], denom, value=-self.lr / bc1)


def _grads(loss, sd, names):
    gs = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(sd[n])) for n, g in zip(names, gs)}


def seg_step(sd, opt, images, truths, spec, n_classes, loss_kw=None):
    """models/segmentation.py:141-157: train(); zero_grad(); out=model(x); loss=crit(out, y.long());
    backward(); Adam.step().  Returns (loss, logits, grads)."""
    loss_kw = loss_kw or dict(weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
    names = opt.names
    for n in names:
        sd[n].requires_grad_(True)
    logits = nets.unet_forward(sd, images, spec, training=True)
    loss = losses.dice_loss(logits, truths.long(), n_classes, **loss_kw)
    grads = _grads(loss, sd, names)
    for n in names:
        sd[n].requires_grad_(False)
    opt.step(sd, grads)
    return loss.detach(), logits.detach(), grads


def reg_step(sd, opt, source, target, lam_reg=1.0):
    """Registration step from the reference's parts: VoxelMorph forward (voxel_morph.py:62-92),
    loss = NCC(warped, target) (loss.py:493-501) + lam_reg * bending(disp) (loss.py:687-730)."""
    names = opt.names
    for n in names:
        sd[n].requires_grad_(True)
    disp, warped, deform = nets.voxelmorph_forward(sd, source, target)
    l_sim = losses.ncc_loss(warped, target)
    l_reg = losses.bending_energy_loss(disp)
    loss = l_sim + lam_reg * l_reg
    grads = _grads(loss, sd, names)
    for n in names:
        sd[n].requires_grad_(False)
    opt.step(sd, grads)
    return loss.detach(), (disp.detach(), warped.detach(), deform.detach()), grads, (l_sim.detach(), l_reg.detach())


def joint_step(seg_sd, seg_opt, reg_sd, reg_opt, im_m, im_t, seg_m, seg_t, spec, n_classes,
               lam_sim=1.0, lam_reg=1.0, lam_anat=1.0, lam_sp=1.0, reduce_grads=None):
    """Joint DeepAtlas alternating step (build-defined from the reference's parts).

    reg phase (seg net frozen): L = lam_sim*NCC(warp(Im), It) + lam_reg*Bending(disp)
                                    + lam_anat*Dice(warp(onehot(seg_m)), onehot(seg_t))   [soft 5-D target path]
    seg phase (reg net frozen): L = lam_sp*Dice(S(Im), seg_m) + lam_anat*Dice(warp(softmax(S(Im)), phi.detach()), onehot(seg_t))
    seg_m=None (the moving image has no manual segmentation): the reg phase warps softmax(S(Im)).detach() -- the segmentation net in
    eval mode, no state change -- instead of onehot(seg_m), and the seg phase has no supervised term.
    reduce_grads(grads, phase) (optional; phase 'reg' | 'seg'): applied to each phase's gradient dict right before its optimiser
    step -- where data-parallel training averages the gradients over the replicas (tests/test_dp_gloo.py).
    Returns dict of losses (+ 'grads_reg' / 'grads_seg': the gradients each optimiser step consumed).
    """
    onehot_t = losses.mask_to_one_hot(seg_t.long().unsqueeze(1), n_classes)
    if seg_m is not None:
        onehot_m = losses.mask_to_one_hot(seg_m.long().unsqueeze(1), n_classes)
    else:
        with torch.no_grad():
            onehot_m = F.softmax(nets.unet_forward(seg_sd, im_m, spec, training=False), dim=1)
    # ---- reg phase
    rn = reg_opt.names
    for n in rn:
        reg_sd[n].requires_grad_(True)
    disp, warped, deform = nets.voxelmorph_forward(reg_sd, im_m, im_t)
    warped_seg = nets.warp_trilinear(onehot_m, deform)
    l_sim = losses.ncc_loss(warped, im_t)
    l_reg = losses.bending_energy_loss(disp)
    l_anat = losses.dice_loss(warped_seg, onehot_t, n_classes, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)
    loss_r = lam_sim * l_sim + lam_reg * l_reg + lam_anat * l_anat
    g = _grads(loss_r, reg_sd, rn)
    for n in rn:
        reg_sd[n].requires_grad_(False)
    if reduce_grads is not None:
        g = reduce_grads(g, 'reg')
    reg_opt.step(reg_sd, g)
    deform = deform.detach()
    # ---- seg phase
    sn = seg_opt.names
    for n in sn:
        seg_sd[n].requires_grad_(True)
    logits = nets.unet_forward(seg_sd, im_m, spec, training=True)
    l_sp = (losses.dice_loss(logits, seg_m.long(), n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
            if seg_m is not None else torch.zeros((), dtype=logits.dtype))
    prob = F.softmax(logits, dim=1)
    l_anat2 = losses.dice_loss(nets.warp_trilinear(prob, deform), onehot_t, n_classes,
                               weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)
    loss_s = lam_sp * l_sp + lam_anat * l_anat2
    g2 = _grads(loss_s, seg_sd, sn)
    for n in sn:
        seg_sd[n].requires_grad_(False)
    if reduce_grads is not None:
        g2 = reduce_grads(g2, 'seg')
    seg_opt.step(seg_sd, g2)
    return dict(loss_reg=loss_r.detach(), loss_seg=loss_s.detach(), sim=l_sim.detach(), bend=l_reg.detach(),
                anat_reg=l_anat.detach(), sup=l_sp.detach(), anat_seg=l_anat2.detach(), grads_reg=g, grads_seg=g2)
