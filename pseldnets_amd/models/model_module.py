"""Task logic around the hot path — mirror of the reference's `models/model_module.py` (SELDModelModule.setup :22-45,
common_step :47-68, training_step :70-81) and `models/components/model_module.py` (standardize :123-126,
configure_optimizers :128-146, configure_loss :171-175), without Lightning: the object exposes the same hooks and
can be driven either hook-by-hook (autograd + torch optimiser, as Lightning would) or through `fused_training_step`
(FusedTrainer: backward, bucketed all-reduce, clip and AdamW inside the MI355X path).
The device part of the validation / test path (prediction incl. ACS, rank gather, moving average, thresholding and
15-degree unification -> DCASE dictionaries) is mirrored too, and `update_metrics` accumulates the SELD scores
(pseldnets_amd/utils/seld_scores.py, host numpy as in the reference)."""
import importlib
import math
import random
from itertools import combinations

import torch

from .. import models
from ..trainer import FusedTrainer
from ..utils.config import get_afextractor

ModelMoodule = {            # (sic) the reference's registry name, models/model_module.py:13-17
    'accdoa': models.accdoa,
    'einv2': models.einv2,
    'multi_accdoa': models.multi_accdoa,
}
_LOSS_KIND = {'accdoa': 'mse', 'multi_accdoa': 'adpit', 'einv2': 'tpit'}


def _get(cfg, dotted, default=None):
    cur = cfg
    for part in dotted.split('.'):
        try:
            cur = cur[part]
        except (KeyError, TypeError):
            return default
    return cur


def instantiate(node):
    """hydra.utils.instantiate subset: {_target_: 'pkg.mod.Class', **kwargs}. Reference loss targets
    (`loss.multi_accdoa.Losses`, ...) are mapped onto this package."""
    kwargs = {k: v for k, v in dict(node).items() if k != '_target_'}
    target = node['_target_']
    if target.startswith('loss.') or target.startswith('augment.'):
        target = 'pseldnets_amd.' + target
    mod, cls = target.rsplit('.', 1)
    return getattr(importlib.import_module(mod), cls)(**kwargs)


class SELDModelModule:
    def __init__(self, cfg, dataset, valid_meta=None, test_meta=None):
        self.cfg = cfg
        self.num_classes = dataset.num_classes
        self.method = _get(cfg, 'model.method')
        self.net = None
        self.training = True
        self.af_extractor = get_afextractor(cfg)
        self.loss = instantiate(_get(cfg, 'model.loss'))
        # Data augmentations (components/model_module.py:61-78): instantiated only when configured
        self.label_res = 0.1
        xy_ratio = _get(cfg, 'data.sample_rate') / _get(cfg, 'data.hoplen') * self.label_res
        types = list(_get(cfg, 'augment.type', []) or [])
        self.data_aug = {'type': types, 'AugMix': bool(_get(cfg, 'augment.AugMix', False))}
        for name in ('trackmix', 'wavmix', 'rotate', 'freqshift', 'crop', 'specaug'):
            node = _get(cfg, 'augment.' + name)
            if node is not None:
                extra = {'xy_ratio': xy_ratio} if name == 'specaug' else {}
                self.data_aug[name] = instantiate(dict(node, **extra))
        missing = [t for t in types if t not in self.data_aug]
        if missing:
            raise KeyError(f"augment.type lists {missing} but cfg.augment has no such node")
        aug_TF = [t for t in types if t not in ('rotate', 'wavmix')]
        self.aug_TF_comb = []
        for n in range(1, len(aug_TF) + 1):
            self.aug_TF_comb += combinations(aug_TF, n)
        self._trainer = None
        self.step_system_outputs = []
        # components/model_module.py:41-53: predictions per test chunk, the per-recording padded frame count, the meta of the split
        self.num_preds_per_chunk = int(round((_get(cfg, 'data.test_chunklen_sec', 10) or 10) / self.label_res))
        self.valid_paths_dict = self.valid_gt_dcase_format = self.test_paths_dict = self.paths_dict = None
        if valid_meta is not None:
            self.valid_paths_dict, self.valid_gt_dcase_format = valid_meta
            self.paths_dict = self.valid_paths_dict
        if test_meta is not None:
            self.test_paths_dict = test_meta
            self.paths_dict = self.test_paths_dict
        self.metrics = None

    def get_num_frames(self, x):
        return int(math.ceil(x / self.num_preds_per_chunk) * self.num_preds_per_chunk)

    def setup(self, stage='fit', device='cuda'):
        feature = _get(self.cfg, 'data.audio_feature')
        in_channels = {'logmelIV': 7, 'salsa': 7, 'salsalite': 7, 'logmelgcc': 10, 'logmel': 1}[feature]
        kwargs = dict(_get(self.cfg, 'model.kwargs', {}))
        self.net = vars(ModelMoodule[self.method])[_get(self.cfg, 'model.backbone')](
            self.cfg, self.num_classes, in_channels, **kwargs)
        if str(_get(self.cfg, 'trainer.precision', '32-true')).startswith('bf16'):
            self.net.compute_dtype = torch.bfloat16
        self.net.to(device)
        if self.af_extractor is not None:
            self.af_extractor.to(device)
        return self

    def standardize(self, batch_x):
        return self.af_extractor(batch_x) if self.af_extractor is not None else batch_x

    def forward(self, x):
        return self.net(x)

    # -- augmentation plumbing (components/model_module.py:83-121) ------------------------------------------------------
    def data_copy(self, batch_x, batch_target):
        batch_x = torch.cat([batch_x] * 3, dim=0)
        batch_target = dict(batch_target)
        for key, value in batch_target.items():
            batch_target[key] = torch.cat([value] * 3, dim=0) if isinstance(value, torch.Tensor) else list(value) * 3
        return batch_x, batch_target

    def augmix_data(self, batch_x, batch_target):
        N = len(batch_x) // 3
        parts_x, parts_t = [batch_x[:N]], [{k: v[:N] for k, v in batch_target.items()}]
        for i in (1, 2):
            x, t = self.augment_data(batch_x[i * N:(i + 1) * N], {k: v[i * N:(i + 1) * N] for k, v in batch_target.items()})
            parts_x.append(x); parts_t.append(t)
        out = {}
        for key in batch_target:
            if 'label' in key:
                out[key] = torch.cat([t[key] for t in parts_t], dim=0)
            else:
                out[key] = list(parts_t[0][key]) + list(parts_t[1][key]) + list(parts_t[2][key])
        return torch.cat(parts_x, dim=0), out

    def augment_data(self, batch_x, batch_y=None):
        if self.data_aug['type'] and self.aug_TF_comb:
            aug_methods = list(random.choice(self.aug_TF_comb))
            random.shuffle(aug_methods)
            for aug_method in aug_methods:
                batch_x, batch_y = self.data_aug[aug_method](batch_x, batch_y)
        return batch_x, batch_y

    def augment_step(self, batch_x, batch_y):
        """The part of common_step (models/model_module.py:47-65) in front of the network: waveform augmentations, feature
        extraction, feature augmentations. Returns (features, targets)."""
        if self.training:
            if self.data_aug['AugMix']:
                batch_x, batch_y = self.data_copy(batch_x, batch_y)
            if 'rotate' in self.data_aug['type']:
                batch_x, batch_y = self.data_aug['rotate'](batch_x, batch_y)
            if 'wavmix' in self.data_aug['type']:
                batch_x, batch_y = self.data_aug['wavmix'](batch_x, batch_y)
        batch_x = self.standardize(batch_x)
        if self.training:
            if self.data_aug['AugMix']:
                batch_x, batch_y = self.augmix_data(batch_x, batch_y)
            else:
                batch_x, batch_y = self.augment_data(batch_x, batch_y)
        return batch_x, batch_y

    def common_step(self, batch_x, batch_y=None):
        batch_x, batch_y = self.augment_step(batch_x, batch_y)
        return self.forward(batch_x), batch_y

    def train(self, mode=True):
        """Lightning toggles the module's `training` flag and the network's together around every fit / validation / test
        loop; without Lightning the hooks below do it themselves (a validation epoch used to leave `training`
        False and silently switch every augmentation off for the rest of the run)."""
        self.training = bool(mode)
        if self.net is not None:
            self.net.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def training_step(self, batch_sample, batch_idx=0):
        """Reference semantics: returns the scalar to back-propagate (loss_dict[loss.loss_type])."""
        self.train()
        batch_target = {k: v for k, v in batch_sample.items() if 'data' not in k}
        pred, target = self.common_step(batch_sample['data'], batch_target)
        loss_dict = self.loss(pred, target)
        return loss_dict[self.loss.loss_type]

    # -- validation / test (models/model_module.py:83-145, components/model_module.py:177-241) ---------------------------
    def predict(self, batch_data):
        """Eval-mode prediction of one batch; cfg.post_processing == 'ACS' runs the 16-pass test-time augmentation."""
        from .. import inference
        was_training = self.training
        self.eval()
        try:
            with torch.no_grad():
                if _get(self.cfg, 'post_processing') == 'ACS':
                    return inference.acs_predict(batch_data, self.standardize, self.forward, output_format=self.method)
                return self.common_step(batch_data)[0]
        finally:
            self.training = was_training          # the next training step sets the network's own flag itself

    def validation_step(self, batch_sample, batch_idx=0):
        batch_target = {k: v for k, v in batch_sample.items() if 'label' in k}
        batch_pred = self.predict(batch_sample['data'])
        self.step_system_outputs.append(batch_pred)
        with torch.no_grad():
            return self.loss({k: v.float() for k, v in batch_pred.items()}, batch_target)

    def test_step(self, batch_sample, batch_idx=0):
        self.step_system_outputs.append(self.predict(batch_sample['data']))

    def pred_aggregation(self, process_group=None, paths_dict=None):
        """All ranks' step outputs in the sampler's order (rank-interleaved, components/model_module.py:178-184), optionally
        stitched by the moving average; returns the prediction tensor(s) on the device, [frames, D] per output key."""
        from .. import inference
        outs, self.step_system_outputs = self.step_system_outputs, []
        merged = {k: torch.cat([o[k].float() for o in outs], dim=0) for k in outs[0]}
        if process_group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(process_group)
            for k, v in merged.items():
                parts = [torch.empty_like(v) for _ in range(world)]
                dist.all_gather(parts, v.contiguous(), group=process_group)
                merged[k] = torch.stack(parts, 0).transpose(0, 1).reshape(-1, *v.shape[1:])
        if self.method == 'multi_accdoa' and _get(self.cfg, 'post_processing') == 'move_avg':
            merged['multi_accdoa'] = inference.move_avg(merged['multi_accdoa'], list((paths_dict or {}).values()),
                                                        _get(self.cfg, 'data.test_chunklen_sec'), _get(self.cfg, 'data.test_hoplen_sec'),
                                                        self.label_res)
        return {k: v.reshape(-1, v.shape[-1]) if k in ('accdoa', 'multi_accdoa') else v.reshape(-1, *v.shape[2:]) for k, v in merged.items()}

    def convert_to_dcase_format_polar(self, pred_frames):
        """{frame: [[class, azimuth_deg, elevation_deg], ...]} for a slice of aggregated prediction frames (one recording);
        einv2: pred_frames = (sed logits [frames, 3, C], doa [frames, 3, 3])."""
        from .. import inference
        thr = _get(self.cfg, 'sed_threshold', 0.5)
        if self.method == 'multi_accdoa':
            return inference.multi_accdoa_to_dcase_polar(pred_frames, self.num_classes, thr)
        if self.method == 'accdoa':
            return inference.accdoa_to_dcase_polar(pred_frames, self.num_classes, thr)
        sed, doa = pred_frames
        return inference.einv2_to_dcase(sed, doa, thr)

    def update_metrics(self, pred_dcase_format, gt_dcase_format, num_frames, metrics=None):
        """components/model_module.py:243-262: accumulate the SELD scores of one recording (DCASE dictionaries in degrees)."""
        from ..utils.seld_scores import SeldScores
        if metrics is None:
            if getattr(self, 'metrics', None) is None:
                self.metrics = SeldScores(doa_threshold=_get(self.cfg, 'doa_threshold', 20), nb_classes=self.num_classes,
                                          label_resolution=self.label_res)
            metrics = self.metrics
        metrics.update(pred_dcase_format, gt_dcase_format, num_frames)
        return metrics

    def _per_recording(self, paths_dict, process_group):
        """(path, label frames, DCASE dictionary) of every recording from the aggregated predictions (model_module.py:114-128,
        165-176): recording i owns get_num_frames(frames_i) consecutive prediction frames, of which the first frames_i count."""
        agg = self.pred_aggregation(process_group, paths_dict)
        frame_ind = 0
        for path, loc_frames in paths_dict.items():
            if self.method == 'einv2':
                frames = (agg['sed'][frame_ind:frame_ind + loc_frames], agg['doa'][frame_ind:frame_ind + loc_frames])
            else:
                frames = agg[self.method][frame_ind:frame_ind + loc_frames]
            yield path, loc_frames, self.convert_to_dcase_format_polar(frames)
            frame_ind += self.get_num_frames(loc_frames)

    def on_validation_epoch_end(self, process_group=None):
        """models/model_module.py:111-145: SELD scores of the validation split, {'macro': {...}, 'micro': {...}}."""
        from ..utils.seld_scores import SeldScores
        self.metrics = SeldScores(doa_threshold=_get(self.cfg, 'doa_threshold', 20), nb_classes=self.num_classes,
                                  label_resolution=self.label_res)
        for path, loc_frames, pred in self._per_recording(self.valid_paths_dict, process_group):
            self.update_metrics(pred, self.valid_gt_dcase_format[path], loc_frames)
        return {'macro': self.metrics.compute('macro'), 'micro': self.metrics.compute('micro')}

    def on_test_epoch_end(self, submissions_dir, process_group=None):
        """models/model_module.py:165-179: one DCASE CSV per recording (`<stem>.csv`) under submissions_dir."""
        from pathlib import Path
        from .. import inference
        out = Path(submissions_dir)
        out.mkdir(parents=True, exist_ok=True)
        written = []
        for path, _, pred in self._per_recording(self.test_paths_dict, process_group):
            csv_path = out.joinpath(Path(path).stem + '.csv')
            inference.write_output_format_file(csv_path, pred)
            written.append(csv_path)
        return written

    def configure_optimizers(self):
        opt_cfg, sch_cfg = _get(self.cfg, 'model.optimizer'), _get(self.cfg, 'model.lr_scheduler')
        optimizer = vars(torch.optim)[opt_cfg['method']](self.net.parameters(), **dict(opt_cfg['kwargs']))
        scheduler = vars(torch.optim.lr_scheduler)[sch_cfg['method']](optimizer, **dict(sch_cfg['kwargs']))
        return [optimizer], [scheduler]

    # -- fused path -----------------------------------------------------------------------------------------------
    def fused_trainer(self, process_group=None):
        if self._trainer is None:
            # the fused step is clip + AdamW + StepLR (configs/model/*.yaml as shipped); anything else the config asks for
            # must not train silently with different dynamics
            om, sm = _get(self.cfg, 'model.optimizer.method', 'AdamW'), _get(self.cfg, 'model.lr_scheduler.method', 'StepLR')
            opt = dict(_get(self.cfg, 'model.optimizer.kwargs', {}))
            if om != 'AdamW' or sm != 'StepLR' or opt.get('amsgrad', False) or _get(self.cfg, 'model.optimizer.multi_opt', False):
                raise NotImplementedError(f"fused training step: optimizer {om} (amsgrad={opt.get('amsgrad', False)}, multi_opt="
                                          f"{_get(self.cfg, 'model.optimizer.multi_opt', False)}) / scheduler {sm} is not built on the MI355X "
                                          "path; use training_step + configure_optimizers (torch optimisers) for it")
            sch = dict(_get(self.cfg, 'model.lr_scheduler.kwargs', {}))
            kind, agg = _LOSS_KIND[self.method], {}
            if hasattr(self.loss, 'weights'):                       # loss.einv2.Losses_agg_pit (configs/loss/einv2_pit_agg.yaml)
                kind, agg = 'agg_pit', dict(agg_weights=self.loss.weights(), agg_l1=self.loss.l1)
            self._trainer = FusedTrainer(
                self.net, self.af_extractor, kind, lr=opt.get('lr', 1e-4),
                max_norm=_get(self.cfg, 'trainer.gradient_clip_val', 1.0), weight_decay=opt.get('weight_decay', 0.01),
                betas=tuple(opt.get('betas', (0.9, 0.999))), eps=opt.get('eps', 1e-8),
                step_size=sch.get('step_size', 20), gamma=sch.get('gamma', 0.1), process_group=process_group,
                sync_bn=bool(_get(self.cfg, 'trainer.sync_batchnorm', False)),
                loss_beta=getattr(self.loss, 'beta', 0.5), **agg)
        return self._trainer

    def fused_training_step(self, batch_sample, process_group=None, next_batch=None):
        """next_batch: the batch the caller will pass next (a loader with one batch of look-ahead): its features are extracted on
        a second stream during this step (trainer.py:prefetch_features). Ignored on the augmentation path, which extracts
        features inside the augmentation chain."""
        batch_target = {k: v for k, v in batch_sample.items() if 'data' not in k}
        self.train()
        if self.data_aug['type'] or self.data_aug['AugMix']:
            feats, batch_target = self.augment_step(batch_sample['data'], batch_target)
            return self.fused_trainer(process_group).training_step(feats, batch_target, is_features=True)
        return self.fused_trainer(process_group).training_step(batch_sample['data'], batch_target,
                                                               next_x=None if next_batch is None else next_batch['data'])
