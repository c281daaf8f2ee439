"""Minimal composer for a Hydra-style `configs/` tree (PyYAML only; hydra / omegaconf are not in the image).

Resolves what the reference's `configs/train.yaml:3-23` and its experiment files use, so that
`python -m pseldnets_amd.train --config-dir <path to a configs/ tree> experiment=synth_maccdoa model.kwargs.embed_dim=96`
composes the same dictionary `python src/train.py experiment=...` would hand to the LightningModule:

  * `defaults:` lists, in order, with `_self_`, `group: option`, `group: null`, `optional group: option`, bare file names
    (relative to the file's own group) and nested groups (`data/starss23: foo`);
  * `# @package _global_` headers (the file merges at the root instead of under its group) and `override /group: option`
    entries inside such files (they re-point an already listed default, whatever their position);
  * command-line overrides: `group=option` and `+group=option` (when `group` is a directory of the tree, nested ones as
    `data/site=roomA`), `a.b.c=value`, `+a.b=value`, `~a.b`;
  * `${a.b.c}` interpolations (resolved after composition, type-preserving when the whole value is one reference),
    `${oc.env:VAR}` / `${oc.env:VAR,default}` and `${now:%fmt}`; other resolvers (`${hydra:...}`) are left as written.

Not a re-implementation of Hydra: no multirun, no structured configs, no `_target_` instantiation (that is
models/model_module.py:instantiate), no package relocation other than `_global_`.
"""
import copy
import datetime
import os
import re

import yaml


class ConfigError(ValueError):
    pass


class _Loader(yaml.SafeLoader):
    """PyYAML's YAML-1.1 float rule needs a '.', so `lr: 5e-5` (configs/experiment/synth_einv2.yaml:11) loads as the STRING '5e-5';
    OmegaConf - what Hydra loads its files with - registers the YAML-1.2 float pattern, and the reference's optimizer receives a
    float. Same pattern here (omegaconf's documented resolver: digits with optional fraction and exponent, .inf, .nan)."""


_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                    |\.[0-9_]+(?:[eE][-+][0-9]+)?
                    |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
                    |[-+]?\.(?:inf|Inf|INF)
                    |\.(?:nan|NaN|NAN))$""", re.X),
    list('-+0123456789.'))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


class AttrDict(dict):
    """dict with attribute access (cfg.model.method), as the reference's code reads its DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: _to_attr(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_to_attr(v) for v in x]
    return x


def _merge(dst, src):
    """Recursive dict merge, src wins; lists and scalars are replaced."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg, dotted, value, create=True):
    parts = dotted.split('.')
    cur = cfg
    for p in parts[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            if not create:
                raise ConfigError(f"override '{dotted}': '{p}' does not exist (use +{dotted}=... to add it)")
            cur[p] = {}
        cur = cur[p]
    if not create and parts[-1] not in cur:
        raise ConfigError(f"override '{dotted}': key does not exist (use +{dotted}=... to add it)")
    cur[parts[-1]] = value


def _del_path(cfg, dotted):
    parts = dotted.split('.')
    cur = cfg
    for p in parts[:-1]:
        cur = cur[p]
    del cur[parts[-1]]


def _get_path(cfg, dotted):
    cur = cfg
    for p in dotted.split('.'):
        if isinstance(cur, list):
            cur = cur[int(p)]
        else:
            cur = cur[p]
    return cur


def _load(path):
    with open(path) as f:
        text = f.read()
    is_global = bool(re.search(r'^\s*#\s*@package\s+_global_\s*$', text, flags=re.M))
    data = _yaml_load(text) or {}
    if not isinstance(data, dict):
        raise ConfigError(f'{path}: a config file must hold a mapping')
    return data, is_global


def _find(config_dir, group, option):
    """`group/option(.yaml)` inside the tree."""
    name = option if option.endswith(('.yaml', '.yml')) else option + '.yaml'
    path = os.path.join(config_dir, group, name) if group else os.path.join(config_dir, name)
    if not os.path.isfile(path):
        raise ConfigError(f"config '{os.path.join(group, name)}' not found under {config_dir}")
    return path


class _Composer:
    def __init__(self, config_dir, group_choices):
        self.dir = config_dir
        self.choices = dict(group_choices)         # group -> option chosen on the command line
        self.overrides = {}                        # group -> option of an `override /group:` entry of an included file
        self.cfg = {}

    def _entry(self, entry, own_group):
        """One item of a defaults list -> (kind, group, option)."""
        if isinstance(entry, str):
            if entry == '_self_':
                return ('self', None, None)
            return ('file', own_group, entry)          # a sibling file of the same group
        if isinstance(entry, dict) and len(entry) == 1:
            (k, v), = entry.items()
            k = k.strip()
            optional = k.startswith('optional ')
            if optional:
                k = k[len('optional '):].strip()
            if k.startswith('override '):
                return ('override', k[len('override '):].strip().lstrip('/'), v)
            # Hydra resolves a RELATIVE group (`- site: roomA` inside data/default.yaml) against the including file's own group
            # (data/site/roomA, packaged under data.site); `/group` is absolute
            group = k.lstrip('/') if (k.startswith('/') or not own_group) else own_group + '/' + k
            return ('optional' if optional else 'group', group, v)
        raise ConfigError(f'cannot read defaults entry {entry!r}')

    def collect_overrides(self, path, own_group):
        """First pass: `override /group: option` entries anywhere in the tree of defaults re-point earlier choices."""
        data, _ = _load(path)
        for entry in data.get('defaults', []) or []:
            kind, group, option = self._entry(entry, own_group)
            if kind == 'override':
                self.overrides[group] = option
            elif kind in ('group', 'optional'):
                opt = self._option(group, option)
                if opt is None:
                    continue
                try:
                    self.collect_overrides(_find(self.dir, group, str(opt)), group)
                except ConfigError:
                    if kind != 'optional':
                        raise
            elif kind == 'file':
                self.collect_overrides(_find(self.dir, own_group, option), own_group)

    def include(self, path, own_group, package):
        """Merge file `path` (and, through its defaults list, what it pulls in) into the config; `package`: dotted key the
        file's own content lands under ('' = root)."""
        data, is_global = _load(path)
        body = {k: v for k, v in data.items() if k != 'defaults'}
        target = '' if is_global else package
        defaults = data.get('defaults', None)
        entries = list(defaults) if defaults else []
        if not any(self._entry(e, own_group)[0] == 'self' for e in entries):
            entries.append('_self_')                       # Hydra 1.1+: an unlisted _self_ goes last (the file overrides its defaults)
        for entry in entries:
            kind, group, option = self._entry(entry, own_group)
            if kind == 'self':
                self._merge_at(target, body)
            elif kind == 'override':
                continue                                   # applied through self.overrides in the first pass
            elif kind == 'file':
                self.include(_find(self.dir, own_group, option), own_group, package)
            else:
                opt = self._option(group, option)
                if opt is None:
                    continue
                try:
                    sub = _find(self.dir, group, str(opt))
                except ConfigError:
                    if kind == 'optional':
                        continue
                    raise
                self.include(sub, group, group.replace('/', '.'))

    def _option(self, group, listed):
        """command line > `override /group:` of an included file > the option the defaults list names"""
        if group in self.choices:
            return self.choices[group]
        if group in self.overrides:
            return self.overrides[group]
        return listed

    def _merge_at(self, package, body):
        if not package:
            _merge(self.cfg, body)
            return
        cur = self.cfg
        for p in package.split('.'):
            cur = cur.setdefault(p, {})
        _merge(cur, body)


_REF = re.compile(r'\$\{([^${}]+)\}')


def _resolve_value(cfg, value, stack):
    if not isinstance(value, str) or '${' not in value:
        return value

    def lookup(expr):
        expr = expr.strip()
        if expr.startswith('oc.env:'):
            name, _, default = expr[len('oc.env:'):].partition(',')
            if name in os.environ:
                return os.environ[name]
            if default != '':
                return default
            return None                                     # left unresolved (the tokens the reference's loggers read)
        if expr.startswith('now:'):
            return datetime.datetime.now().strftime(expr[len('now:'):])
        if ':' in expr:
            return None                                     # other resolvers (hydra:...) are left as written
        if expr in stack:
            raise ConfigError(f"interpolation cycle through '{expr}'")
        try:
            return _resolve_value(cfg, _get_path(cfg, expr), stack + (expr,))
        except (KeyError, IndexError, TypeError):
            return None

    m = _REF.fullmatch(value.strip())
    if m:                                                   # the whole value is one reference: keep its type
        v = lookup(m.group(1))
        return value if v is None else v
    prev = None
    while prev != value:                                    # innermost references first (nested ${a.${b}} forms)
        prev = value
        value = _REF.sub(lambda mm: (lambda v: mm.group(0) if v is None else str(v))(lookup(mm.group(1))), value)
    return value


def _resolve_tree(cfg, node):
    if isinstance(node, dict):
        for k in list(node.keys()):
            node[k] = _resolve_tree(cfg, node[k])
        return node
    if isinstance(node, list):
        return [_resolve_tree(cfg, v) for v in node]
    return _resolve_value(cfg, node, ())


def _parse_scalar(text):
    try:
        return _yaml_load(text)
    except yaml.YAMLError:
        return text


def compose(config_dir, config_name='train', overrides=(), resolve=True):
    """The composed configuration of `<config_dir>/<config_name>.yaml` under the command-line `overrides` (an AttrDict)."""
    config_dir = os.path.abspath(config_dir)
    if not os.path.isdir(config_dir):
        raise ConfigError(f'{config_dir} is not a directory')
    choices, sets, adds, dels, appended = {}, [], [], [], []
    for ov in overrides:
        if ov.startswith('~'):
            dels.append(ov[1:].split('=')[0])
            continue
        if '=' not in ov:
            raise ConfigError(f"override '{ov}' is not of the form key=value")
        key, _, val = ov.partition('=')
        add = key.startswith('+')
        key = key.lstrip('+')
        if '.' not in key and os.path.isdir(os.path.join(config_dir, key)):
            if add:
                appended.append((key, val))                 # `+group=option`: a group the defaults list does not name
            else:
                choices[key] = None if val in ('null', 'None', '') else val
        else:
            (adds if add else sets).append((key, _parse_scalar(val)))
    comp = _Composer(config_dir, choices)
    root = _find(config_dir, '', config_name)
    comp.collect_overrides(root, '')
    comp.include(root, '', '')
    for group, option in appended:
        comp.include(_find(config_dir, group, option), group, group.replace('/', '.'))
    cfg = comp.cfg
    for key, val in sets:
        _set_path(cfg, key, val, create=False)
    for key, val in adds:
        _set_path(cfg, key, val, create=True)
    for key in dels:
        _del_path(cfg, key)
    if resolve:
        _resolve_tree(cfg, cfg)
    cfg.pop('hydra', None)                                   # Hydra strips its own node from the composed job config
    return _to_attr(cfg)
