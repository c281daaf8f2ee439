"""EINV2 losses on MI355X — mirror of the reference's `loss/einv2.py`: track-wise PIT (Losses_pit :30-116) and the AGG loss
(Losses_agg_pit :118-188)."""
from ._fn import AggPitFn, TpitFn


class Losses_pit(object):
    def __init__(self, loss_fn, loss_type, method, loss_beta):
        if loss_fn['sed'] != 'bce' or loss_fn['doa'] != 'mse' or method != 'tPIT':
            raise NotImplementedError("built for configs/loss/einv2_pit.yaml: {sed: bce, doa: mse}, method tPIT")
        self.max_ov = 3
        self.beta = loss_beta
        self.loss_type = loss_type
        self.PIT_type = method
        self.names = ['loss_all', 'loss_BCEWithLogits', 'loss_MSE']
        self.loss_dict_keys = ['loss_all', 'loss_sed', 'loss_doa', 'loss_other']

    def __call__(self, pred, target, epoch_it=0):
        sed_l = target['sed_label'][:, :, :self.max_ov, :]
        doa_l = target['doa_label'][:, :, :self.max_ov, :]
        loss_all, loss_sed, loss_doa = TpitFn.apply(pred['sed'], pred['doa'], sed_l, doa_l, self.beta)
        return {'loss_all': loss_all, 'loss_sed': loss_sed, 'loss_doa': loss_doa, 'loss_other': 0.}


class Losses_agg_pit(object):
    def __init__(self, loss_fn, loss_type, loss_alpha, method):
        if loss_fn not in ('mse', 'l1'):
            raise NotImplementedError(f"loss_fn {loss_fn}: the reference defines mse and l1 (einv2.py:121-126)")
        self.l1 = loss_fn == 'l1'
        self.max_ov = 3
        self.loss_type = loss_type
        self.alpha = loss_alpha
        self.method = method
        self.names = ['loss_all']
        self.loss_dict_keys = ['loss_all', 'loss_agg', 'loss_accdoa', 'loss_other']

    def weights(self):
        """(w_agg, w_acc) of loss_all = w_agg * loss_agg + w_acc * loss_accdoa (einv2.py:145-157)."""
        if self.method == 'mACCDOA_pit':
            return 1.0, 0.0
        if self.method == 'ACCDOA':
            return 0.0, 1.0
        return float(self.alpha), 1.0 - float(self.alpha)

    def __call__(self, pred, target, epoch_it=0):
        w_agg, w_acc = self.weights()
        loss_all, loss_agg, loss_acc = AggPitFn.apply(pred['sed'], pred['doa'], target['sed_label'], target['doa_label'],
                                                      w_agg, w_acc, self.l1)
        return {'loss_all': loss_all, 'loss_agg': loss_agg if w_agg else 0., 'loss_accdoa': loss_acc if w_acc else 0.,
                'loss_other': 0.}
