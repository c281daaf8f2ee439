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
