"""`losswrapper` of reference codes/network/loss/losses.py:21-50 on the HIP loss kernels."""
import torch

from ... import ops


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, pred_p, pred_l, target, factors, reg_l2, use_mask):
        pred, pred_p, pred_l, target = (t.contiguous() for t in (pred, pred_p, pred_l, target))
        ctx.save_for_backward(pred, pred_p, pred_l, target)
        ctx.cfg = (factors, reg_l2, use_mask)
        return ops.loss_fwd(pred, pred_p, pred_l, target, factors, reg_l2, use_mask)

    @staticmethod
    def backward(ctx, g):
        pred, pred_p, pred_l, target = ctx.saved_tensors
        factors, reg_l2, use_mask = ctx.cfg
        g_pred, g_p, g_l = ops.loss_bwd(pred, pred_p, pred_l, target, g.contiguous(), factors, reg_l2, use_mask)
        return g_pred, g_p, g_l, None, None, None, None


def losswrapper(predict, predict_shuffle_p, predict_shuffle_l, target, cfg, rest_out=None, rest_view=None,
                loss1_gt=None, loss2_gt=None):
    """Returns (loss, f0*loss1, f1*loss2, f2*loss3[, loss_unsperv]) as 0-dim tensors; `loss` supports .backward().
    The Standin terms compare against the detached prediction (OurLoss1, losses.py:5-18)."""
    if cfg.SOLVER.reg_loss == 'l2_loss':
        reg_l2 = True
    elif cfg.SOLVER.reg_loss == 'l1_loss':
        reg_l2 = False
    else:
        raise NotImplementedError
    if loss1_gt is not None or loss2_gt is not None:
        raise NotImplementedError("loss1_gt / loss2_gt are only produced by model variants outside Nef-Net")
    using = cfg.SOLVER.loss_using
    use_mask = (1 if 1 in using else 0) | (2 if 2 in using else 0) | (4 if 3 in using else 0)
    factors = tuple(float(f) for f in cfg.SOLVER.loss_factor)
    target = target.to(torch.float32).expand_as(predict)
    L4 = _LossFn.apply(predict, predict_shuffle_p, predict_shuffle_l, target, factors, reg_l2, use_mask)
    result = (L4[0], L4[1].detach(), L4[2].detach(), L4[3].detach())
    if rest_out is not None and rest_view is not None:
        ro = rest_out.detach().to(torch.float32).contiguous()
        rv = rest_view.detach().to(torch.float32).contiguous()
        unsup = ops.loss_fwd(ro, ro, ro, rv, (0.0, 0.0, 1.0), reg_l2, 4)[3]
        return result + (unsup,)
    return result


class MSELead(torch.nn.Module):
    """Reference losses.py:53-64: the mean over leads of the per-lead MSE (unused by the Nef-Net path).  Every lead has
    the same number of elements, so this is the MSE over the whole tensor -- one pass of the loss kernels."""

    def forward(self, input, target):
        target = target.to(torch.float32).expand_as(input)
        # element [0] (the total) is the one _LossFn back-propagates through; with factors (0, 0, 1) and only the
        # reconstruction term enabled it IS the MSE (element [3] carries the same number but no gradient)
        return _LossFn.apply(input, input, input, target, (0.0, 0.0, 1.0), True, 4)[0]
