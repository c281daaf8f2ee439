"""Autograd nodes around the fused loss kernels (value and gradient come out of one pass; backward just scales)."""
import torch

from .. import ops


class AdpitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label):
        loss, dpred = ops.adpit_loss(pred.contiguous().float(), label.contiguous().float())
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


class MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        loss, dpred = ops.mse_loss(pred.contiguous().float(), target.contiguous().float())
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None


class TpitFn(torch.autograd.Function):
    """returns (loss_all, loss_sed, loss_doa); only loss_all carries a gradient (as the reference back-propagates
    loss_dict['loss_all'])."""

    @staticmethod
    def forward(ctx, sed, doa, sed_label, doa_label, beta):
        loss3, dsed, ddoa = ops.tpit_loss(sed.contiguous().float(), doa.contiguous().float(),
                                          sed_label.contiguous().float(), doa_label.contiguous().float(), beta)
        ctx.save_for_backward(dsed, ddoa)
        ctx.mark_non_differentiable
        return loss3[0], loss3[1].detach(), loss3[2].detach()

    @staticmethod
    def backward(ctx, g_all, g_sed, g_doa):
        dsed, ddoa = ctx.saved_tensors
        return dsed * g_all, ddoa * g_all, None, None, None


class AggPitFn(torch.autograd.Function):
    """returns (loss_all, loss_agg, loss_accdoa); only loss_all carries a gradient."""

    @staticmethod
    def forward(ctx, sed, doa, sed_label, doa_label, w_agg, w_acc, l1):
        loss3, dsed, ddoa = ops.agg_pit_loss(sed.contiguous().float(), doa.contiguous().float(),
                                             sed_label.contiguous().float(), doa_label.contiguous().float(), w_agg, w_acc, l1)
        ctx.save_for_backward(dsed, ddoa)
        return loss3[0], loss3[1].detach(), loss3[2].detach()

    @staticmethod
    def backward(ctx, g_all, g_agg, g_acc):
        dsed, ddoa = ctx.saved_tensors
        return dsed * g_all, ddoa * g_all, None, None, None, None, None
