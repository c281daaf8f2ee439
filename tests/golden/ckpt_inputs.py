"""Synthetic checkpoints shared by the checkpoint-loading golden generator and the tests (no reference code involved): every
tensor is a closed-form function of its key name and shape (oracle/htsat.py:formula_tensor), so the generator (which takes the
key / shape lists from the reference's modules) and the tests (which take them from this package's modules) build identical
files for the keys they share."""
import torch

from oracle.htsat import formula_tensor


def ck_tensor(name, shape):
    if name.endswith('num_batches_tracked'):
        return torch.tensor(7, dtype=torch.long)
    return formula_tensor('ck.' + name, tuple(shape)).float()


def _bn0(mel=64):
    return {f'bn0.{leaf}': ck_tensor(f'bn0.{leaf}', () if leaf == 'num_batches_tracked' else (mel,))
            for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')}


def htsat_audioset(enc_shapes):
    """{'state_dict': {'sed_model.<encoder key>': ...}}: one-channel patch embedding, bn0 (accdoa.py:172-190, einv2.py:239-263)."""
    sd = {}
    for k, shp in enc_shapes.items():
        if k == 'patch_embed.proj.weight':
            shp = (shp[0], 1, shp[2], shp[3])
        sd['sed_model.' + k] = ck_tensor(k, shp)
    sd.update({'sed_model.' + k: v for k, v in _bn0().items()})
    return {'state_dict': sd}


def passt_audioset(enc_shapes, extra_t=7, extra_f=2):
    """Flat dict; one-channel patch embedding; longer time / frequency positional embeddings (centre-cropped by the loader,
    accdoa.py:273-300); the classifier rows head.1.* that the loader drops."""
    sd = {}
    for k, shp in enc_shapes.items():
        shp = tuple(shp)
        if k == 'patch_embed.proj.weight':
            shp = (shp[0], 1, shp[2], shp[3])
        elif k == 'time_new_pos_embed':
            shp = shp[:3] + (shp[3] + extra_t,)
        elif k == 'freq_new_pos_embed':
            shp = shp[:2] + (shp[2] + extra_f, shp[3])
        sd[k] = ck_tensor(k, shp)
    E = enc_shapes['norm.weight'][0]
    sd['head.1.weight'], sd['head.1.bias'] = ck_tensor('head.1.weight', (527, E)), ck_tensor('head.1.bias', (527,))
    return sd


def cnn14_audioset(enc_shapes):
    """{'model': {...}}: PANNs CNN14 with a one-channel first convolution and bn0 (accdoa.py:44-55, einv2.py:69-86)."""
    sd = {}
    for k, shp in enc_shapes.items():
        if k == 'conv_block1.conv1.weight':
            shp = (shp[0], 1, 3, 3)
        sd[k] = ck_tensor(k, shp)
    sd.update(_bn0())
    return {'model': sd}


def pseld(all_shapes, compiled=False):
    """A PSELDNets (Lightning) checkpoint of the same network: {'state_dict': {'net.<key>': ...}} (+ '_orig_mod.' when compiled)."""
    pre = 'net._orig_mod.' if compiled else 'net.'
    return {'state_dict': {pre + k: ck_tensor('net.' + k, shp) for k, shp in all_shapes.items()}}


def shapes(state_dict, prefix=''):
    return {k[len(prefix):]: tuple(v.shape) for k, v in state_dict.items() if k.startswith(prefix)}


def checksums(state_dict, keys):
    """[sum, abs-sum, first, last] per key (float64)."""
    out = []
    for k in keys:
        v = state_dict[k].detach().double().reshape(-1)
        out.append([v.sum().item(), v.abs().sum().item(), v[0].item(), v[-1].item()])
    return out


def keep_index_buffers(ckpt, sd0, prefix, kind, pseld_prefix=None):
    """Integer buffers (relative-position indices) are not synthetic: the checkpoint carries the network's own."""
    inner = ckpt.get('state_dict', ckpt.get('model', ckpt))
    for k, v in sd0.items():
        if v.is_floating_point() or 'num_batches_tracked' in k:
            continue
        if pseld_prefix is not None:
            inner[pseld_prefix + k] = v.clone()
        elif k.startswith(prefix):
            inner[('sed_model.' if kind == 'htsat' else '') + k[len(prefix):]] = v.clone()
    return ckpt


def checked_keys(sd0):
    return [k for k, v in sd0.items() if (v.is_floating_point() or 'num_batches_tracked' in k) and 'attn_mask' not in k]
