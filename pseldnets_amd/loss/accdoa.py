"""ACCDOA loss on MI355X — mirror of the reference's `loss/accdoa.py` (Losses :3-22)."""
from ._fn import MseFn


class Losses(object):
    def __init__(self, loss_fn, loss_type):
        super().__init__()
        if loss_fn != 'mse':
            raise NotImplementedError("only loss_fn='mse' (the shipped configs/loss/accdoa.yaml) is built")
        self.loss_type = loss_type
        self.names = ['loss_MSE']
        self.loss_dict_keys = ['loss_all', 'loss_accdoa', 'loss_other']

    def __call__(self, pred, target):
        loss = MseFn.apply(pred['accdoa'], target['accdoa_label'])
        return {'loss_all': loss + 0.0, 'loss_accdoa': loss, 'loss_other': 0.}
