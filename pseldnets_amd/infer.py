"""`python -m pseldnets_amd.infer mode=valid|test [wav_dir=DIR] [meta_dir=DIR] [ckpt_path=FILE] [out_dir=DIR] [a.b=c ...]` —
minimal mirror of the reference's `src/infer.py:19-99` for the device part of the validation / test path: the recordings
of the split live in HBM as 16-bit PCM (`DeviceClipStore`), test chunks are cut on the device from the reference's index
rows (`utils/data_utilities.py:6-64`), predicted in eval mode (optionally with the ACS test-time augmentation or the
moving-average stitching, `post_processing=ACS|move_avg`), aggregated, decoded to DCASE dictionaries and either scored
(mode=valid: ER / F / LE / LR against `meta_dir/<stem>.csv`) or written as `out_dir/<stem>.csv` (mode=test).
Hydra composition, Lightning, loggers and the HDF5 / FLAC readers are out of scope; the config keys are the reference's
(`configs/infer.yaml`: mode, sed_threshold, ckpt_path). Without `wav_dir` the split is `n_clips` synthetic recordings."""
import math
import os
import sys
from collections import OrderedDict
from pathlib import Path

import torch

from . import inference
from .data.ingest import DeviceClipStore
from .train import SyntheticDataset, compose

INFER_DEFAULTS = ['mode=valid', 'sed_threshold=0.5', 'ckpt_path=null', 'wav_dir=null', 'meta_dir=null', 'out_dir=submissions',
                  'n_clips=3', 'clip_sec=23', 'data.test_chunklen_sec=10', 'data.test_hoplen_sec=10']


def load_checkpoint(net, path):
    """A PSELDNets (Lightning) checkpoint: {'state_dict': {'net.<key>': tensor}} (keys of a compiled module carry
    '_orig_mod.', models/model_module.py:101-109), or a plain state dict."""
    ck = torch.load(path, map_location='cpu')
    sd = ck['state_dict'] if isinstance(ck, dict) and 'state_dict' in ck else ck
    sd = {k.replace('_orig_mod.', '').removeprefix('net.'): v for k, v in sd.items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    allowed = ('relative_position_index', 'attn_mask')
    bad = [k for k in missing if not any(a in k for a in allowed)]
    if bad or unexpected:
        raise KeyError(f'checkpoint {path}: missing {bad[:5]}, unexpected {list(unexpected)[:5]}')


def build_split(cfg, device):
    """(DeviceClipStore, index rows, paths_dict {recording: label frames}) of the split (data/components/data.py:72-88)."""
    sr = cfg.data.sample_rate
    store = DeviceClipStore(device, 4)
    if cfg.wav_dir:
        for p in sorted(Path(cfg.wav_dir).glob('*.wav')):
            store.add_wav(p, sample_rate=sr)
    else:
        g = torch.Generator().manual_seed(cfg.seed)
        for i in range(cfg.n_clips):
            n = int((cfg.clip_sec + 1.7 * i) * sr)
            store.add_clip(f'synthetic/mix_{i:03d}.wav', (torch.randn(n, 4, generator=g) * 3000).to(torch.int16))
    if not store.names:
        raise FileNotFoundError(f'no recordings under {cfg.wav_dir}')
    # test index rows always zero-pad the trailing chunk (preproc/preprocess.py:468), so chunk k of a recording starts at k * hop
    rows = store.index_rows(int(cfg.data.test_chunklen_sec * sr), int(cfg.data.test_hoplen_sec * sr), last_frame_always_paddding=True)
    points_per_prediction = int(sr * 0.1)
    paths = OrderedDict()
    for r in rows:
        paths[r[0]] = int(math.ceil(r[2] / points_per_prediction))
    return store, rows, paths


def main(argv=None):
    from .models.model_module import SELDModelModule
    cfg = compose(INFER_DEFAULTS + list(sys.argv[1:] if argv is None else argv))
    torch.manual_seed(cfg.seed)
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    store, rows, paths = build_split(cfg, device)
    gts = None
    if cfg.mode == 'valid':
        gts = OrderedDict()
        for path in paths:
            meta = Path(cfg.meta_dir, Path(path).stem + '.csv') if cfg.meta_dir else None
            gts[path] = inference.load_output_format_file(meta) if meta is not None and meta.exists() else {}
    module = SELDModelModule(cfg, SyntheticDataset(cfg), valid_meta=(paths, gts) if gts is not None else None,
                             test_meta=paths if cfg.mode == 'test' else None).setup('test', device)
    if cfg.ckpt_path:
        load_checkpoint(module.net, cfg.ckpt_path)
    chunk_len = int(cfg.data.test_chunklen_sec * cfg.data.sample_rate)
    bs = cfg.model.batch_size
    for i in range(0, len(rows), bs):
        module.test_step({'data': store.chunks(rows[i:i + bs], chunk_len)})
    if cfg.mode == 'valid':
        scores = module.on_validation_epoch_end()
        for avg in ('macro', 'micro'):
            d = scores[avg]
            print(f"val/{avg}: ER20 {d['ER']:.4f}  F20 {d['F']:.4f}  LE {d['LE']:.2f}  LR {d['LR']:.4f}  SELD {d['SELD_scr']:.4f}")
        return scores
    written = module.on_test_epoch_end(cfg.out_dir)
    print(f'{len(written)} DCASE files written to {cfg.out_dir}')
    return written


if __name__ == '__main__':
    main()
