"""The config composer (pseldnets_amd/utils/hydra_lite.py) on a SYNTHETIC mini tree (tests/mini_configs: invented content, the
mechanisms of the reference's configs/train.yaml:3-23 and configs/experiment/*.yaml): defaults lists with _self_ ordering,
`# @package _global_`, `override /group:`, nested defaults inside a group, `${...}` interpolation, command-line overrides."""
import os

import pytest

from pseldnets_amd.utils.hydra_lite import ConfigError, compose

TREE = os.path.join(os.path.dirname(__file__), 'mini_configs')


def test_defaults_self_order_and_group_packages():
    c = compose(TREE, 'train')
    assert (c.model.method, c.model.backbone, c.model.batch_size) == ('accdoa', 'Tiny', 4)      # model/small.yaml under `model`
    assert c.model.loss == {'_target_': 'loss.accdoa.Losses', 'loss_fn': 'mse'}                # loss/plain.yaml: @package _global_
    assert c.augment == {'type': [], 'AugMix': False} and c.trainer.max_epochs == 1
    assert c.seed == 7 and c.compile is True and 'experiment' not in c and 'debug' not in c      # `group: null` entries add nothing
    # callbacks/default.yaml pulls its sibling file first and overrides one key of it with its own content (_self_ last)
    assert c.callbacks.checkpoint == {'monitor': 'val/score', 'save_top_k': 1}


def test_experiment_overrides_groups_and_values():
    c = compose(TREE, 'train', ['experiment=exp1'])
    # `override /model: big.yaml` + `override /augment: mix.yaml` re-point the primary defaults; the experiment body wins over them
    assert (c.model.backbone, c.model.kwargs.embed_dim, c.model.batch_size) == ('Large', 64, 32)
    assert c.model.optimizer.kwargs == {'lr': 0.0001, 'amsgrad': False}                         # dict merge keeps untouched keys
    assert c.augment.AugMix is True and c.augment.type == ['specaug', 'rotate']
    assert c.trainer.max_epochs == 25 and c.trainer.gradient_clip_val == 1.0 and c.seed == 2024
    assert c.task_name == 'multi_accdoa_Large'            # interpolation sees the composed values (model/big.yaml sets the method)
    # the command line beats the experiment's group override; value overrides beat everything
    c = compose(TREE, 'train', ['experiment=exp1', 'model=small', 'loss=multi', 'model.kwargs.embed_dim=96', '+extra.flag=true', '~compile'])
    assert (c.model.backbone, c.model.kwargs.embed_dim, c.model.method) == ('Tiny', 96, 'multi_accdoa')
    assert c.model.loss._target_ == 'loss.multi_accdoa.Losses' and c.extra.flag is True and 'compile' not in c
    assert c.task_name == 'multi_accdoa_Tiny' and c.run_dir == '/tmp/runs/multi_accdoa_Tiny'


def test_interpolation_and_resolvers(monkeypatch):
    c = compose(TREE, 'train')
    assert c.data.frames_per_second == 24000 and isinstance(c.data.frames_per_second, int)      # whole-value reference keeps its type
    assert c.token == 'none' and c.unresolved == '${hydra:runtime.output_dir}' and len(c.stamp) == 4
    monkeypatch.setenv('PSELD_TEST_TOKEN', 'abc')
    assert compose(TREE, 'train').token == 'abc'
    assert compose(TREE, 'train', resolve=False).task_name == '${model.method}_${model.backbone}'


def test_nested_groups_and_errors():
    c = compose(TREE, 'train', ['+data/site=roomA'])                                          # a nested group the defaults do not list
    assert c.data.site == {'name': 'roomA', 'channels': 4} and c.data.sample_rate == 24000
    with pytest.raises(ConfigError):
        compose(TREE, 'train', ['model=does_not_exist'])
    with pytest.raises(ConfigError):
        compose(TREE, 'train', ['model.no_such_key=1'])                                         # plain overrides must hit an existing key
    with pytest.raises(ConfigError):
        compose(TREE, 'train', ['justaword'])
    assert compose(TREE, 'train', ['+model.no_such_key=1']).model.no_such_key == 1
    # a RELATIVE group inside a group file resolves against that file's group (Hydra), and the `hydra` node is stripped
    c = compose(TREE, 'train', ['data=with_site'])
    assert c.data.site == {'name': 'roomA', 'channels': 4} and c.data.sample_rate == 16000 and 'site' not in c
    assert 'hydra' not in c


def test_builtin_table_gives_the_command_line_group_choice_precedence_over_the_experiment():
    """`augment=default experiment=...` must compose like `experiment=... augment=default` (Hydra: a command-line group
    choice beats the experiment's `override /augment`, whatever the argument order)."""
    from pseldnets_amd.train import compose as train_compose
    a = train_compose(['augment=default', 'experiment=synth_maccdoa'])
    b = train_compose(['experiment=synth_maccdoa', 'augment=default'])
    assert a.augment == b.augment and not a.augment.AugMix and list(a.augment.type) == []
    assert train_compose(['experiment=synth_maccdoa']).augment.AugMix


def test_train_entry_composes_from_a_tree():
    from pseldnets_amd.train import compose as train_compose
    c = train_compose(['--config-dir', TREE, 'experiment=exp1', 'model.batch_size=8'])
    assert c.model.batch_size == 8 and c.model.backbone == 'Large' and c.data.num_classes == 170 and c.trainer.limit_train_batches == 10
    # the built-in tables: every synth_* experiment carries the reference's `override /augment: augmix.yaml`
    b = train_compose(['experiment=synth_maccdoa'])
    assert b.augment.AugMix is True and 'wavmix' in b.augment.type
    assert train_compose(['experiment=synth_maccdoa', 'augment=default']).augment.AugMix is False


# ---- the reference's own experiments: expected values READ from its YAML files (file:line cited), held here as data ----------------------------
AUGMIX_TYPES = ['specaug', 'crop', 'freqshift', 'rotate', 'trackmix', 'wavmix']            # configs/augment/augmix.yaml:5-11
EXPECTED = {
    # configs/experiment/synth_maccdoa.yaml:3-4 (override /augment: augmix, /loss: multi_accdoa), :8 batch_size, :10 lr, :12 step_size,
    # :15 max_epochs, :19 seed; configs/loss/multi_accdoa.yaml:5-11; configs/trainer/default.yaml:31 gradient_clip_val;
    # configs/model/htsat.yaml:3,29-36 (backbone, amsgrad, StepLR gamma, warm-up steps)
    'synth_maccdoa': {'model.batch_size': 32, 'model.optimizer.method': 'AdamW', 'model.optimizer.kwargs.lr': 1e-4,
                      'model.optimizer.kwargs.amsgrad': False, 'model.lr_scheduler.method': 'StepLR',
                      'model.lr_scheduler.kwargs.step_size': 20, 'model.lr_scheduler.kwargs.gamma': 0.1,
                      'model.method': 'multi_accdoa', 'model.backbone': 'HTSAT', 'model.loss._target_': 'loss.multi_accdoa.Losses',
                      'model.loss.loss_fn': 'mse', 'model.loss.loss_type': 'loss_all', 'model.kwargs.embed_dim': 96,
                      'model.kwargs.depths': [2, 2, 6, 2], 'model.kwargs.num_heads': [4, 8, 16, 32], 'model.kwargs.drop_path_rate': 0.1,
                      'augment.AugMix': True, 'augment.type': AUGMIX_TYPES, 'augment.rotate.p': 0.8, 'augment.specaug.T': 40,
                      'trainer.gradient_clip_val': 1.0, 'seed': 2024},
    # configs/experiment/synth_einv2.yaml:3-4,8 batch_size 17, :11 lr 5e-5 (a YAML-1.2 float: PyYAML alone reads the string '5e-5'), :13
    # step_size 6, :16 max_epochs; configs/loss/einv2_pit.yaml:5-12
    'synth_einv2': {'model.batch_size': 17, 'model.optimizer.method': 'AdamW', 'model.optimizer.kwargs.lr': 5e-5,
                    'model.optimizer.kwargs.amsgrad': False, 'model.lr_scheduler.kwargs.step_size': 6, 'model.lr_scheduler.kwargs.gamma': 0.1,
                    'model.method': 'einv2', 'model.backbone': 'HTSAT', 'model.loss._target_': 'loss.einv2.Losses_pit',
                    'model.loss.loss_fn': {'sed': 'bce', 'doa': 'mse'}, 'model.loss.method': 'tPIT', 'model.loss.loss_beta': 0.5,
                    'augment.AugMix': True, 'augment.type': AUGMIX_TYPES, 'trainer.gradient_clip_val': 1.0, 'seed': 2024},
}
# what only the reference's tree defines (the built-in tables of pseldnets_amd/train.py keep their own trainer block: synthetic loop)
EXPECTED_TREE_ONLY = {
    'synth_maccdoa': {'trainer.max_epochs': 25, 'trainer.num_sanity_val_steps': -1, 'trainer.precision': '32-true', 'model.num_warmup_steps': 5,
                      'task_name': 'multi_accdoa_HTSAT', 'data.sample_rate': 24000, 'data.n_mels': 64, 'data.audio_feature': 'logmelIV'},
    'synth_einv2': {'trainer.max_epochs': 8, 'task_name': 'einv2_HTSAT'},
}
REF_TREE = '/root/reference/configs'


def _get(cfg, dotted):
    cur = cfg
    for part in dotted.split('.'):
        cur = cur[part]
    return cur


def _check(cfg, expected, where):
    for key, want in expected.items():
        got = _get(cfg, key)
        if isinstance(want, float):
            assert isinstance(got, float) and abs(got - want) <= 1e-12 * abs(want), (where, key, got, want)
        elif isinstance(want, (dict, list)):
            assert got == want, (where, key, got, want)
        else:
            assert got == want and type(got) is type(want), (where, key, got, want)


@pytest.mark.parametrize("exp", sorted(EXPECTED))
def test_builtin_tables_give_the_reference_experiments_values(exp):
    from pseldnets_amd.train import compose as train_compose
    _check(train_compose([f'experiment={exp}']), EXPECTED[exp], f'built-in {exp}')


@pytest.mark.skipif(not os.path.isdir(REF_TREE), reason="the reference's configs/ tree exists only in the build container")
@pytest.mark.parametrize("exp", sorted(EXPECTED))
def test_reference_tree_composes_to_the_expected_values(exp):
    """hydra_lite on the reference's real configs/ tree (it composed all 21 experiments but no test pinned a value)."""
    cfg = compose(REF_TREE, 'train', [f'experiment={exp}'])
    _check(cfg, EXPECTED[exp], f'tree {exp}')
    _check(cfg, EXPECTED_TREE_ONLY[exp], f'tree {exp}')
    # the DDP recipe: configs/trainer/gpu.yaml:4-10 on top of the experiment
    ddp = compose(REF_TREE, 'train', [f'experiment={exp}', 'trainer=gpu'])
    assert (ddp.trainer.strategy, ddp.trainer.devices, ddp.trainer.sync_batchnorm, ddp.trainer.use_distributed_sampler) == ('ddp', 2, True, False)
    assert ddp.trainer.gradient_clip_val == 1.0 and ddp.trainer.max_epochs == EXPECTED_TREE_ONLY[exp]['trainer.max_epochs']


@pytest.mark.skipif(not os.path.isdir(REF_TREE), reason="the reference's configs/ tree exists only in the build container")
def test_every_reference_experiment_composes_and_its_learning_rate_is_a_number():
    base = os.path.join(REF_TREE, 'experiment')
    n = 0
    for root, _dirs, files in os.walk(base):
        for f in sorted(files):
            if f.endswith('.yaml'):
                rel = os.path.relpath(os.path.join(root, f), base)[:-5]
                cfg = compose(REF_TREE, 'train', [f'experiment={rel}'])
                assert isinstance(cfg.model.optimizer.kwargs.lr, float) and 0 < cfg.model.optimizer.kwargs.lr < 1, rel
                assert isinstance(cfg.model.batch_size, int) and cfg.trainer.gradient_clip_val == 1.0, rel
                n += 1
    assert n == 21


def test_yaml_1_2_floats_load_as_numbers():
    from pseldnets_amd.utils.hydra_lite import _yaml_load
    d = _yaml_load("a: 5e-5\nb: 1e3\nc: 0.1\nd: 12\ne: abc\nf: '5e-5'\ng: -2E+3\nh: 3e")
    assert d == {'a': 5e-5, 'b': 1000.0, 'c': 0.1, 'd': 12, 'e': 'abc', 'f': '5e-5', 'g': -2000.0, 'h': '3e'}
    assert isinstance(d['a'], float) and isinstance(d['d'], int) and isinstance(d['f'], str)
