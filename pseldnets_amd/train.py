"""`python -m pseldnets_amd.train experiment=synth_maccdoa [a.b=c ...]` — minimal mirror of the reference's
`src/train.py:19-75` for the hot path: seed -> dataset descriptor -> model module -> fused MI355X train loop on
SYNTHETIC in-memory batches (the data layer, Hydra composition, Lightning callbacks/loggers are out of scope; the
config keys are the reference's, defaults from configs/{data/default,model/htsat,experiment/synth_maccdoa}.yaml)."""
import copy
import json
import os
import sys
import time

import torch

DEFAULT_CFG = {
    'seed': 2024, 'compile': False,
    'data': {'sample_rate': 24000, 'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64,
             'audio_feature': 'logmelIV', 'train_chunklen_sec': 10, 'num_classes': 170},
    'model': {'method': 'multi_accdoa', 'backbone': 'HTSAT', 'batch_size': 32,
              'kwargs': {'spec_size': 256, 'patch_size': 4, 'patch_stride': [4, 4], 'embed_dim': 96,
                         'depths': [2, 2, 6, 2], 'num_heads': [4, 8, 16, 32], 'window_size': 8, 'mlp_ratio': 4,
                         'qkv_bias': True, 'drop_rate': 0., 'attn_drop_rate': 0., 'drop_path_rate': 0.1, 'ape': False,
                         'patch_norm': True, 'norm_before_mlp': 'ln', 'audioset_pretrain': True, 'pretrained_path': None},
              'loss': {'_target_': 'loss.multi_accdoa.Losses', 'loss_fn': 'mse', 'loss_type': 'loss_all'},
              'optimizer': {'method': 'AdamW', 'kwargs': {'lr': 1e-4, 'amsgrad': False}},
              'lr_scheduler': {'method': 'StepLR', 'kwargs': {'step_size': 20, 'gamma': 0.1}}},
    # configs/augment/default.yaml (the `augment=augmix` group option = configs/augment/augmix.yaml)
    'augment': {'type': [], 'AugMix': False,
                'trackmix': {'_target_': 'augment.TrackMix', 'alpha': 0.5},
                'wavmix': {'_target_': 'augment.WavMix', 'alpha': 0.5, 'p': 0.5},
                'rotate': {'_target_': 'augment.Rotation', 'p': 0.8, 'rotation_type': 48},
                'specaug': {'_target_': 'augment.SpecAugment', 'T': 40, 'F': 8, 'mT': 4, 'mF': 2},
                'crop': {'_target_': 'augment.Crop', 'T': 8, 'F': 4, 'mC': 4},
                'freqshift': {'_target_': 'augment.FreqShift', 'p': 0.5, 'shift_range': 15, 'direction': 'None', 'mode': 'reflect'}},
    'trainer': {'max_epochs': 1, 'gradient_clip_val': 1.0, 'precision': 'bf16-mixed', 'sync_batchnorm': False,
                'limit_train_batches': 10},
    'adapt': {},
}
ADAPT_GROUPS = {      # configs/adapt/{default,adapter,mono_adapter}.yaml
    'default': {'method': 'none'},
    'adapter': {'method': 'adapter', 'adapt_kwargs': {'position': ['MlpAdapter', 'SpatialAdapter'], 'type': 'adapter', 'mlp_ratio': 0.5,
                                                      'adapter_scalar': 0.1, 'act_layer': 'gelu'}},
    'lora': {'method': 'lora', 'linear_kwargs': {'r': 16, 'lora_alpha': 1, 'lora_dropout': 0., 'fan_in_fan_out': False, 'merge_weights': True},
             'conv_kwargs': {'r': 16, 'lora_alpha': 1}},
    'mono_adapter': {'method': 'mono_adapter', 'adapt_kwargs': {'position': ['MlpAdapter', 'SpatialAdapter'], 'type': 'adapter',
                                                                'mlp_ratio': 0.5, 'act_layer': 'gelu', 'adapter_scalar': 0.1}},
}
MODEL_GROUPS = {      # configs/model/{htsat,passt,crnn}.yaml: backbone + kwargs (+ decoder keys); `model=<name>` replaces them
    'htsat': {'backbone': 'HTSAT', 'kwargs': copy.deepcopy(DEFAULT_CFG['model']['kwargs'])},
    'passt': {'backbone': 'PASST', 'decoder': None, 'num_decoder_layers': 2, 'ps_gap': 2,
              'kwargs': {'u_patchout': 0, 's_patchout_t': 0, 's_patchout_f': 0, 'img_size': [64, 1001], 'patch_size': 16, 'stride': 10,
                         'embed_dim': 768, 'depth': 7, 'num_heads': 12, 'mlp_ratio': 4, 'qkv_bias': True, 'representation_size': None,
                         'distilled': True, 'drop_rate': 0., 'drop_path_rate': 0., 'norm_layer': None, 'act_layer': None,
                         'audioset_pretrain': True, 'pretrained_path': None}},
    'crnn': {'backbone': 'CRNN', 'decoder': 'conformer', 'num_decoder_layers': 1,
             'kwargs': {'encoder': 'CNN12', 'num_features': [64, 128, 256, 512, 1024, 2048], 'audioset_pretrain': True,
                        'pretrained_path': None}},
}
AUGMENT_GROUPS = {
    'default': {},
    'augmix': {'type': ['specaug', 'crop', 'freqshift', 'rotate', 'trackmix', 'wavmix'], 'AugMix': True},
}
EXPERIMENTS = {      # configs/experiment/synth_*.yaml bodies over their `override /loss:` file (trainer.max_epochs stays this entry's own: synthetic loop)
    'synth_maccdoa': {'model': {'batch_size': 32, 'optimizer': {'kwargs': {'lr': 1e-4}}, 'lr_scheduler': {'kwargs': {'step_size': 20}}}},
    'synth_accdoa': {'model': {'method': 'accdoa', 'batch_size': 40, 'loss': {'_target_': 'loss.accdoa.Losses', 'loss_fn': 'mse', 'loss_type': 'loss_all'},
                               'optimizer': {'kwargs': {'lr': 1e-4}}, 'lr_scheduler': {'kwargs': {'step_size': 20}}}},
    # configs/experiment/synth_einv2_agg.yaml (einv2.HTSAT) / synth_seddoa_agg.yaml (model.backbone=HTSAT_SEDDOA) over
    # configs/loss/einv2_pit_agg.yaml
    'synth_einv2_agg': {'model': {'method': 'einv2', 'batch_size': 17, 'thresh_unify': 10,
                                  'loss': {'_target_': 'loss.einv2.Losses_agg_pit', 'loss_fn': 'mse', 'loss_type': 'loss_all',
                                           'loss_alpha': 0., 'method': 'mACCDOA_pit'},
                                  'optimizer': {'method': 'AdamW', 'kwargs': {'lr': 5e-5}}, 'lr_scheduler': {'kwargs': {'step_size': 6}}}},
    'synth_seddoa_agg': {'model': {'method': 'einv2', 'backbone': 'HTSAT_SEDDOA', 'batch_size': 40, 'thresh_unify': 10,
                                   'loss': {'_target_': 'loss.einv2.Losses_agg_pit', 'loss_fn': 'mse', 'loss_type': 'loss_all',
                                            'loss_alpha': 0., 'method': 'mACCDOA_pit'},
                                   'optimizer': {'method': 'AdamW', 'kwargs': {'lr': 1e-4}}, 'lr_scheduler': {'kwargs': {'step_size': 10}}}},
    # configs/experiment/synth_einv2.yaml:7-13 over configs/loss/einv2_pit.yaml
    'synth_einv2': {'model': {'method': 'einv2', 'batch_size': 17,
                              'loss': {'_target_': 'loss.einv2.Losses_pit', 'loss_fn': {'sed': 'bce', 'doa': 'mse'},
                                       'loss_type': 'loss_all', 'method': 'tPIT', 'loss_beta': 0.5},
                              'optimizer': {'method': 'AdamW', 'kwargs': {'lr': 5e-5}}, 'lr_scheduler': {'kwargs': {'step_size': 6}}}},
}


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def _fill_missing(dst, src):
    for k, v in src.items():
        if k not in dst:
            dst[k] = copy.deepcopy(v)
        elif isinstance(v, dict) and isinstance(dst[k], dict):
            _fill_missing(dst[k], v)


def compose(argv):
    """`[--config-dir DIR [--config-name NAME]] key=value ...`. With --config-dir (or PSELD_CONFIG_DIR) the configuration is composed
    from that Hydra-style tree (utils/hydra_lite.py: defaults lists, `# @package _global_`, `override /group:`, interpolations,
    key=value / +key=value / ~key overrides) - point it at the reference's own `configs/` and `experiment=synth_maccdoa` resolves
    exactly as `python src/train.py experiment=synth_maccdoa` does; the keys the synthetic loop of this entry needs and that tree
    does not define (data.num_classes, trainer.limit_train_batches) are filled from the built-in defaults. Without it the
    built-in tables below (the same groups, transcribed) are used."""
    argv = list(argv)
    config_dir, config_name = os.environ.get('PSELD_CONFIG_DIR'), 'train'
    rest = []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ('--config-dir', '--config-name') and i + 1 < len(argv):
            if a == '--config-dir':
                config_dir = argv[i + 1]
            else:
                config_name = argv[i + 1]
            i += 2
            continue
        if a.startswith('--config-dir='):
            config_dir = a.partition('=')[2]
        elif a.startswith('--config-name='):
            config_name = a.partition('=')[2]
        else:
            rest.append(a)
        i += 1
    if config_dir:
        from .utils.hydra_lite import compose as compose_tree
        cfg = dict(compose_tree(config_dir, config_name, rest))
        _fill_missing(cfg, {'data': {'num_classes': DEFAULT_CFG['data']['num_classes']},
                            'trainer': {'limit_train_batches': DEFAULT_CFG['trainer']['limit_train_batches']}, 'adapt': {}})
        return AttrDict(cfg)
    cfg = copy.deepcopy(DEFAULT_CFG)
    # Hydra gives a command-line group choice precedence over the experiment's `override /group` whatever the argument order
    # (and the --config-dir path above does the same): the experiment is applied first, then the explicit group choices, then
    # the dotted overrides (`augment=default experiment=synth_maccdoa` must not end with AugMix on)
    #  a `model=` / `adapt=` choice SUBSTITUTES the group the experiment's defaults list would pick, and the experiment
    # BODY (`_self_` last in its defaults) is merged over the chosen group - so the groups go first, then the experiment, then the
    # augment choice (which must beat the experiment's `override /augment`), then the dotted overrides
    order = {'model': 0, 'adapt': 0, 'experiment': 1, 'augment': 2}
    rest = sorted(rest, key=lambda a: order.get(a.partition('=')[0], 3))          # stable: equal ranks keep their order
    explicit_augment = any(a.partition('=')[0] == 'augment' for a in rest)
    for arg in rest:
        key, _, val = arg.partition('=')
        if key == 'experiment':
            _merge(cfg, copy.deepcopy(EXPERIMENTS[val]))
            if not explicit_augment:
                _merge(cfg['augment'], copy.deepcopy(AUGMENT_GROUPS['augmix']))  # every synth_* experiment: `override /augment: augmix.yaml`
            continue
        if key == 'model':
            group = copy.deepcopy(MODEL_GROUPS[val])
            cfg['model']['kwargs'] = {}
            for k in ('decoder', 'num_decoder_layers', 'ps_gap'):
                cfg['model'].pop(k, None)
            _merge(cfg['model'], group)
            continue
        if key == 'augment':
            if val == 'default':
                cfg['augment'].update(type=[], AugMix=False)
            _merge(cfg['augment'], copy.deepcopy(AUGMENT_GROUPS[val]))
            continue
        if key == 'adapt':
            cfg['adapt'] = copy.deepcopy(ADAPT_GROUPS[val])
            continue
        try:
            val = json.loads(val)
        except json.JSONDecodeError:
            pass
        cur = cfg
        parts = key.split('.')
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = val
    return AttrDict(cfg)


class SyntheticDataset:
    max_ov, label_resolution = 3, 0.1

    def __init__(self, cfg):
        self.num_classes = cfg.data.num_classes


def synthetic_batch(cfg, method, device, gen):
    B, C = cfg.model.batch_size, cfg.data.num_classes
    L = int(cfg.data.train_chunklen_sec * cfg.data.sample_rate)
    batch = {'data': 0.1 * torch.randn(B, 4, L, device=device, generator=gen)}
    act = (torch.rand(B, 100, C, device=device, generator=gen) < 0.02).float()
    doa = torch.randn(B, 100, 3, C, device=device, generator=gen)
    doa = doa / doa.norm(dim=2, keepdim=True).clamp_min(1e-6) * act.unsqueeze(2)
    if method == 'multi_accdoa':
        lab = torch.zeros(B, 100, 6, 4, C, device=device)
        lab[:, :, 0, 0], lab[:, :, 0, 1:] = act, doa
        batch['adpit_label'] = lab
        batch['ov'] = ['1'] * B                      # one source per synthetic chunk (TrackMix / WavMix pair these)
    elif method == 'accdoa':
        batch['accdoa_label'] = doa.reshape(B, 100, 3 * C)
        batch['ov'] = ['1'] * B
    elif method == 'einv2':                          # track-wise labels: the events on track 0 (one class per frame), tracks 1-2 silent
        first = ((act.cumsum(-1) == 1) & (act > 0)).float()
        sed = torch.zeros(B, 100, 3, C, device=device)
        sed[:, :, 0] = first
        dl = torch.zeros(B, 100, 3, 3, device=device)
        dl[:, :, 0] = (doa * first.unsqueeze(2)).sum(-1)
        batch['sed_label'], batch['doa_label'] = sed, dl
        batch['ov'] = ['1'] * B
    else:
        raise NotImplementedError(method)
    return batch


def main(argv=None):
    from .models.model_module import SELDModelModule
    cfg = compose(sys.argv[1:] if argv is None else argv)
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.manual_seed(cfg.seed)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)
        group = dist.group.WORLD
    module = SELDModelModule(cfg, SyntheticDataset(cfg)).setup('fit', device)
    if world > 1:
        import torch.distributed as dist
        for p in module.net.parameters():
            dist.broadcast(p.data, 0)
    gen = torch.Generator(device=device).manual_seed(cfg.seed + rank)
    trainer = module.fused_trainer(group)
    batches, n_batches = None, cfg.trainer.limit_train_batches
    if cfg.data.get('wav_dir'):
        # recordings + DCASE metadata on disk -> HBM-resident split, batches drawn by the reference's rank-strided sampler
        # (data/components/sampler.py) and assembled on the device (data/ingest.py:DeviceSELDDataset)
        from pathlib import Path
        from .data.components.sampler import UserDistributedBatchSampler
        from .data.ingest import DeviceClipStore, DeviceSELDDataset
        store = DeviceClipStore(device, 4)
        # recordings: *.wav (RIFF PCM16) and *.flac (the reference's synthetic datasets, data/components/data.py:81); labels: the DCASE
        # metadata CSVs beside them (data.meta_dir), or the HDF5 label file the reference's preprocessing wrote (data.label_h5:
        # .../{adpit,accdoa,track}/<type>/<dataset>.h5); rows: the dataset's own index, or the reference's index CSV (data.index_csv)
        wavs = sorted(list(Path(cfg.data.wav_dir).glob('*.wav')) + list(Path(cfg.data.wav_dir).glob('*.flac')))
        for w in wavs:
            store.add_audio(w, sample_rate=cfg.data.sample_rate)
        label_h5 = cfg.data.get('label_h5')
        metas = None if label_h5 else {str(w): Path(cfg.data.get('meta_dir') or cfg.data.wav_dir) / (w.stem + '.csv') for w in wavs}
        ds = DeviceSELDDataset(store, metas, cfg.model.method, cfg.data.num_classes, cfg.data.sample_rate, cfg.data.train_chunklen_sec,
                               cfg.data.get('train_hoplen_sec', cfg.data.train_chunklen_sec),
                               mono_adapter=(cfg.adapt or {}).get('method') == 'mono_adapter',
                               index_csv=cfg.data.get('index_csv'), label_h5=label_h5)
        sampler = UserDistributedBatchSampler(len(ds), cfg.model.batch_size, seed=cfg.seed)
        batches, n_batches = iter(sampler), min(n_batches, len(sampler)) if n_batches else len(sampler)
    # -- save / resume: the counterpart of Lightning's ModelCheckpoint + `ckpt_path=...` (reference: configs/train.yaml:32-33,
    #    configs/callbacks/default.yaml, src/train.py:49-50 `trainer.fit(..., ckpt_path=cfg.get("ckpt_path"))`). Rank 0 writes
    #    <paths.output_dir>/checkpoints/last.ckpt at the end of every epoch (and epoch_NNN.ckpt every `trainer.save_every_n_epochs`);
    #    `ckpt_path=FILE` makes every rank load it before the loop: weights, AdamW moments + step, StepLR epoch, and the data order -
    #    the sampler's pointer / permutation / RandomState and the synthetic generator's state, so a resumed run draws the batches the
    #    uninterrupted run would have drawn.
    out_dir = (cfg.get('paths') or {}).get('output_dir') or cfg.get('output_dir') or os.path.join(os.getcwd(), 'outputs')
    ckpt_dir = os.path.join(out_dir, 'checkpoints')
    first_epoch = 0
    if cfg.get('ckpt_path'):
        state = torch.load(cfg.ckpt_path, map_location=device, weights_only=False)
        trainer.load_state_dict(state['trainer'])
        first_epoch = int(state['epoch']) + 1
        per_rank = state['data'][rank] if rank < len(state['data']) else None
        if per_rank is not None:
            gen.set_state(per_rank['generator'].cpu())
            torch.set_rng_state(per_rank['torch_rng'].cpu())                   # DropPath / dropout / augmentation draws continue where they stopped
            torch.cuda.set_rng_state(per_rank['cuda_rng'].cpu(), device)
            if batches is not None and per_rank.get('sampler') is not None:
                sampler.pointer, sampler.indices = int(per_rank['sampler']['pointer']), per_rank['sampler']['indices'].copy()
                if sampler.shuffle:
                    sampler.random_state.set_state(per_rank['sampler']['random_state'])
        if rank == 0:
            print(f"resumed from {cfg.ckpt_path}: epoch {first_epoch}, optimiser step {state['trainer']['optimizer']['step']}")

    def save_checkpoint(epoch):
        data = {'generator': gen.get_state(), 'torch_rng': torch.get_rng_state(), 'cuda_rng': torch.cuda.get_rng_state(device),
                'sampler': None if batches is None else {'pointer': sampler.pointer, 'indices': sampler.indices.copy(),
                                                         'random_state': sampler.random_state.get_state() if sampler.shuffle else None}}
        if world > 1:                          # every rank's data-order state travels to rank 0 (small host objects)
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, data)
        else:
            gathered = [data]
        if rank != 0:
            return
        os.makedirs(ckpt_dir, exist_ok=True)
        state = {'epoch': epoch, 'trainer': trainer.state_dict(), 'data': gathered, 'world': world}
        tmp = os.path.join(ckpt_dir, 'last.ckpt.tmp')
        torch.save(state, tmp)
        os.replace(tmp, os.path.join(ckpt_dir, 'last.ckpt'))           # (never a half-written last.ckpt)
        every = int(cfg.trainer.get('save_every_n_epochs', 0) or 0)
        if every and (epoch + 1) % every == 0:
            torch.save(state, os.path.join(ckpt_dir, f'epoch_{epoch:03d}.ckpt'))

    for epoch in range(first_epoch, cfg.trainer.max_epochs):
        t0 = time.perf_counter()
        draw = lambda: ds.batch(next(batches)) if batches is not None else synthetic_batch(cfg, cfg.model.method, device, gen)
        nxt = draw()
        for it in range(n_batches):
            batch, nxt = nxt, (draw() if it + 1 < n_batches else None)       # one batch of look-ahead: its features are prefetched
            loss = module.fused_training_step(batch, group, next_batch=nxt)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"epoch {epoch}: loss_all {loss['loss_all'].item():.5f}  lr {trainer.lr:.2e}  "
                  f"{n_batches * cfg.model.batch_size * world / (time.perf_counter() - t0):.1f} chunks/s")
        trainer.end_epoch()
        if cfg.trainer.get('enable_checkpointing', True):
            save_checkpoint(epoch)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
