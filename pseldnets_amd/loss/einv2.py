"""EINV2 track-wise PIT loss on MI355X — mirror of the reference's `loss/einv2.py` (Losses_pit :30-116)."""
from ._fn import TpitFn


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
