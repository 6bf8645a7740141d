"""Registration and joint DeepAtlas training steps built from the reference's parts.

The reference lists registration and joint training as TODO (README.md:15-19); only the components exist: the
registration net (lib/network_factory/voxel_morph.py), the segmentation net, Dice with soft targets
(lib/loss.py:435-436), NCC / bending losses and the one-hot transform.  The step definitions below are the
build's composition of those parts, identical to oracle/steps.py (reg_step / joint_step), which the GPU tests
compare against:

  reg phase (seg net frozen):  L = l_sim*NCC(warp(Im), It) + l_reg*Bending(disp) + l_anat*Dice(warp(onehot(Sm)), onehot(St))
  seg phase (reg net frozen):  L = l_sp*Dice(S(Im), Sm) + l_anat*Dice(warp(softmax(S(Im)), phi.detach()), onehot(St))
  Sm = None (moving image without a manual segmentation): the reg phase warps softmax(S(Im)).detach() (segmentation net in eval
  mode under no_grad: no state change) instead of onehot(Sm), and the seg phase drops its supervised term.
"""
import os

import torch

from .. import ops, parallel, trace
from ..lib.loss import DiceLossMultiClass, NormalizedCrossCorrelationLoss, BendingEnergyLoss


class RegistrationStep:
    """One registration optimisation step: VoxelMorph forward -> NCC + lambda * bending -> backward -> Adam."""

    def __init__(self, reg_model, optimizer, lam_reg=1.0):
        self.model, self.opt, self.lam_reg = reg_model, optimizer, lam_reg
        self.ncc, self.bend = NormalizedCrossCorrelationLoss(), BendingEnergyLoss()

    def gradients(self, source, target):
        """zero_grad -> forward -> losses -> backward (device work only: capturable in a HIP graph, graphs.GraphedStep)."""
        self.model.train()
        self.opt.zero_grad()
        with trace.range('reg/forward'):
            disp, warped, deform = self.model(source, target)
        with trace.range('reg/loss'):
            l_sim = self.ncc(warped, target)
            l_reg = self.bend(disp)
            loss = l_sim + self.lam_reg * l_reg
        with trace.range('reg/backward'):
            loss.backward()
        return dict(loss=loss.detach(), disp=disp.detach(), warped=warped.detach(), deform=deform.detach(), sim=l_sim.detach(), bend=l_reg.detach())

    def segments(self, source, target):
        """(segments, between, optimizers) for graphs.GraphedStep: the gradient all-reduce sits between the two segments."""
        return ([lambda: self.gradients(source, target), lambda: self.opt.step()],
                [lambda: parallel.allreduce_gradients(self.opt)], [self.opt])

    def __call__(self, source, target):
        r = self.gradients(source, target)
        with trace.range('reg/allreduce'):
            parallel.allreduce_gradients(self.opt)
        with trace.range('reg/adam'):
            self.opt.step()
        return r['loss'], (r['disp'], r['warped'], r['deform']), (r['sim'], r['bend'])


class DeepAtlasJointStep:
    """Alternating joint step (one reg phase + one seg phase per image pair)."""

    def __init__(self, seg_model, seg_opt, reg_model, reg_opt, n_classes,
                 lam_sim=1.0, lam_reg=1.0, lam_anat=1.0, lam_sp=1.0, fused=True):
        self.fused = fused             # fused anatomy losses (ops.LabelWarpDiceFn / ops.SegPhaseLossFn); False: the op-by-op composition
        self.seg, self.seg_opt, self.reg, self.reg_opt = seg_model, seg_opt, reg_model, reg_opt
        self.n_classes = n_classes
        self.lam = dict(sim=lam_sim, reg=lam_reg, anat=lam_anat, sp=lam_sp)
        self.ncc, self.bend = NormalizedCrossCorrelationLoss(), BendingEnergyLoss()
        self.dice_logits = DiceLossMultiClass(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=True, eps=1e-6)
        self.dice_prob = DiceLossMultiClass(n_class=n_classes, weight_type='Uniform', no_bg=False, softmax=False, eps=1e-6)

    # The segmentation net's FORWARD pass does not depend on the registration phase (with a manual segmentation of the moving image: the registration phase warps
    # one-hot(seg_m), and the deformation is only needed by the segmentation phase's LOSS).  It is issued first, on its own stream, and runs beside the registration
    # phase -- a chain of small latency-bound kernels that leaves most of the GPU idle -- instead of after it.  Same kernels, same order inside either network:
    # results are those of the sequential step.  DA_JOINT_OVERLAP=0 (or overlap_phases=False) runs the phases one after the other.
    overlap_phases = os.environ.get('DA_JOINT_OVERLAP', '1') == '1'
    _phase_stream = None
    _ev_disp = None

    def _seg_forward_ahead(self, im_m):
        main = torch.cuda.current_stream()
        if self._phase_stream is None:
            self._phase_stream = torch.cuda.Stream(priority=int(os.environ.get('DA_JOINT_PS_PRIO', '-1')))      # high priority: the segmentation phase is the step's long chain (-0.14 ms; 0 = normal)
        ps = self._phase_stream
        ps.wait_stream(main)                       # (the previous step's segmentation update, the input)
        with torch.cuda.stream(ps):
            self.seg.train()
            self.seg_opt.zero_grad()
            with trace.range('joint/seg_phase/forward'):
                logits = ops.materialize_logits(self.seg(im_m))
        return logits

    def __call__(self, im_m, im_t, seg_m, seg_t):
        ahead = None
        if (self.overlap_phases and seg_m is not None and im_m.is_cuda and not torch.cuda.is_current_stream_capturing()):
            ahead = self._seg_forward_ahead(im_m)
        r = self.reg_gradients(im_m, im_t, seg_m, seg_t)
        parallel.allreduce_gradients(self.reg_opt)
        self.reg_opt.step()
        s = self.seg_gradients(im_m, seg_m, seg_t, r['disp'], ahead)
        parallel.allreduce_gradients(self.seg_opt)
        self.seg_opt.step()
        r.pop('disp')
        r.update(s)
        return r

    def segments(self, im_m, im_t, seg_m, seg_t):
        """(segments, between, optimizers) for graphs.GraphedStep: reg gradients | reg update + seg gradients | seg update, with the
        two flat-bucket all-reduces in the gaps."""
        st = {}

        def first():
            st.update(self.reg_gradients(im_m, im_t, seg_m, seg_t))
            return {k: v for k, v in st.items() if k != 'disp'}

        def second():
            self.reg_opt.step()
            return self.seg_gradients(im_m, seg_m, seg_t, st['disp'])

        return ([first, second, lambda: self.seg_opt.step()],
                [lambda: parallel.allreduce_gradients(self.reg_opt), lambda: parallel.allreduce_gradients(self.seg_opt)],
                [self.reg_opt, self.seg_opt])

    def reg_gradients(self, im_m, im_t, seg_m, seg_t):
        """registration phase up to its gradients (segmentation net frozen)."""
        lam = self.lam
        # Dice against one-hot(seg_t): the fused kernel takes the index mask directly (t in {0,1} either way), so the target
        # one-hot is never materialised
        self.reg.train()
        self.reg_opt.zero_grad()
        if seg_m is None:
            with torch.no_grad():
                self.seg.eval()
                prob_m = ops.SoftmaxFn.apply(ops.materialize_logits(self.seg(im_m)))
        with trace.range('joint/reg_phase/forward'):
            disp, warped, deform = self.reg(im_m, im_t)
        if disp.is_cuda and not torch.cuda.is_current_stream_capturing():
            self._ev_disp = torch.cuda.Event()
            self._ev_disp.record()                 # the segmentation phase's losses (phase stream) wait for this, not for the rest of the registration phase
        fused = self.fused and ops.fused_anatomy_supported(self.n_classes)
        trace.mark('joint/reg_phase/losses')
        l_sim = self.ncc(warped, im_t)
        l_reg = self.bend(disp)
        if seg_m is not None and fused:
            # Dice(warp(one_hot(seg_m)), one_hot(seg_t)) straight from the two label maps: no 32-channel tensor in either direction
            l_anat = ops.LabelWarpDiceFn.apply(seg_m, seg_t, disp, self.n_classes, 'Uniform', False, 1e-6)
        else:
            if seg_m is not None:
                warped_seg = ops.WarpLabelsFn.apply(seg_m, disp, self.n_classes)      # = warp(one_hot(seg_m)), one-hot never materialised
            else:
                warped_seg, _ = ops.WarpFn.apply(prob_m, disp)                          # gradient flows to disp only (prob_m is a constant)
            l_anat = self.dice_prob(warped_seg, seg_t)
        loss_r = lam['sim'] * l_sim + lam['reg'] * l_reg + lam['anat'] * l_anat
        with trace.range('joint/reg_phase/backward'):
            loss_r.backward()
        return dict(loss_reg=loss_r.detach(), sim=l_sim.detach(), bend=l_reg.detach(), anat_reg=l_anat.detach(), disp=disp.detach())

    def seg_gradients(self, im_m, seg_m, seg_t, disp, logits_ahead=None):
        """segmentation phase up to its gradients (deformation fixed).  logits_ahead: the forward pass was already issued on the phase stream (_seg_forward_ahead)."""
        lam = self.lam
        fused = self.fused and ops.fused_anatomy_supported(self.n_classes)
        if logits_ahead is not None:
            # the whole phase stays on the phase stream: its losses wait for the deformation (an event behind the registration forward), not for the registration
            # phase's backward pass and update, and run beside them
            main, ps = torch.cuda.current_stream(), self._phase_stream
            ps.wait_event(self._ev_disp)
            disp.record_stream(ps)                 # (allocated on the main stream, read on the phase stream)
            with torch.cuda.stream(ps):
                out = self._seg_losses_backward(logits_ahead, seg_m, seg_t, disp, fused)
            main.wait_stream(ps)                   # the update (this stream) comes after the backward pass; weight gradients: FlatAdam.step() joins the side stream
            for v in out.values():
                v.record_stream(main)
            return out
        self.seg.train()
        self.seg_opt.zero_grad()
        with trace.range('joint/seg_phase/forward'):
            logits = ops.materialize_logits(self.seg(im_m))
        return self._seg_losses_backward(logits, seg_m, seg_t, disp, fused)

    def _seg_losses_backward(self, logits, seg_m, seg_t, disp, fused):
        lam = self.lam
        trace.mark('joint/seg_phase/losses')
        if fused and not ops.DETERMINISTIC:
            # both Dice terms as one node: structured adjoint warp + one pass to the logit gradient (ops.SegPhaseLossFn); its scatter
            # uses float atomics, so deterministic runs take the composed path below (fixed-point accumulation in WarpFn)
            l_sp, l_anat2 = ops.SegPhaseLossFn.apply(logits, seg_m, disp, seg_t, 'Uniform', False, 1e-6)
        else:
            l_sp = self.dice_logits(logits, seg_m) if seg_m is not None else torch.zeros((), device=logits.device)
            prob = ops.SoftmaxFn.apply(logits)
            warped_prob, _ = ops.WarpFn.apply(prob, disp)
            l_anat2 = self.dice_prob(warped_prob, seg_t)
        loss_s = lam['sp'] * l_sp + lam['anat'] * l_anat2
        with trace.range('joint/seg_phase/backward'):
            loss_s.backward()
        return dict(loss_seg=loss_s.detach(), sup=l_sp.detach(), anat_seg=l_anat2.detach())
