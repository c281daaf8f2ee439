"""ADPIT loss on MI355X — mirror of the reference's `loss/multi_accdoa.py` (Losses :5-105): same constructor,
attributes (.names, .loss_type, .loss_dict_keys) and returned dict; the 13 candidate targets, the class-wise
minimum and the gradient are one fused kernel (pseld_adpit_loss) instead of 13 full-size temporaries."""
from ._fn import AdpitFn


class Losses(object):
    def __init__(self, loss_fn, loss_type):
        super().__init__()
        if loss_fn != 'mse':
            raise NotImplementedError("ADPIT is built with the MSE criterion (the reference hard-codes nn.MSELoss)")
        self.names = ['loss_all', 'loss_adpit', 'loss_other']
        self.loss_type = loss_type
        self.loss_dict_keys = ['loss_all', 'loss_adpit', 'loss_other']

    def __call__(self, output, target, epoch_it=0):
        """
        output: {'multi_accdoa': [batch_size, frames, num_track*num_axis*num_class]}
        target: {'adpit_label': [batch_size, frames, num_track_dummy=6, num_axis=4, num_class]}
        """
        loss = AdpitFn.apply(output['multi_accdoa'], target['adpit_label'])
        return {'loss_all': loss + 0., 'loss_adpit': loss, 'loss_other': 0.}
