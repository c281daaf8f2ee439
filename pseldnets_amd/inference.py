"""Inference-side post-processing on the MI355X — mirror of the device-relevant part of the reference's validation / test
path: `BaseModelModule.post_processing` (ACS 16-pass test-time augmentation and the moving average
over overlapping test chunks, models/components/model_module.py:269-329), `pred_aggregation` (:177-221: thresholding of
the ACCDOA / multi-ACCDOA outputs) and `convert_to_dcase_format_polar` (:223-241 with utils/data_utilities.py:197-388).
The rotations, the averaging, the activity thresholds and the 15-degree unification run as HIP kernels on the gathered
predictions; only the final, variable-length DCASE dictionaries ({frame: [[class, azimuth, elevation], ...]}) and the CSV
files are assembled on the host; the SELD scores on them are `pseldnets_amd.utils.seld_scores`.
"""
import math

import numpy as np
import torch

from . import ops

# components/model_module.py:272-275
ACS_TRANS = {(0, 1, 2): (1, 2, 3), (1, 0, 2): (3, 2, 1)}
ACS_SIGNS = [[1, 1, 1], [-1, 1, 1], [1, -1, 1], [-1, -1, 1], [1, 1, -1], [-1, 1, -1], [1, -1, -1], [-1, -1, -1]]


def acs_predict(batch, standardize, forward, output_format='multi_accdoa'):
    """post_processing(method='ACS'): batch f32 [B, 4, L] FOA waveforms; `standardize` = the feature extractor,
    `forward` = the network (returns {output_format: [B, T, (3*)3*C]}). Returns {output_format: mean over the 16 passes}."""
    if output_format not in ('multi_accdoa', 'accdoa'):
        raise NotImplementedError(output_format)
    B, dev = batch.shape[0], batch.device
    batch = batch.contiguous().float()
    acc = None
    for sx, sy, sz in ACS_SIGNS:
        for (xx, yy, zz), (s_x, s_y, s_z) in ACS_TRANS.items():
            src = torch.tensor([[s_x, s_y, s_z]] * B, dtype=torch.int32, device=dev)
            sign = torch.tensor([[sy, sz, sx]] * B, dtype=torch.float32, device=dev)
            y = forward(standardize(ops.aug_rotate_wave(batch, src, sign)))[output_format].contiguous().float()
            T = y.shape[1]
            s_axis = (sx, sy, sz)
            lsrc = torch.tensor([[xx, yy, zz]] * B, dtype=torch.int32, device=dev)
            lsign = torch.tensor([[s_axis[xx], s_axis[yy], s_axis[zz]]] * B, dtype=torch.float32, device=dev)
            tracks = 3 if output_format == 'multi_accdoa' else 1
            y = ops.aug_rotate_label(y, lsrc, lsign, T * tracks, 3, y.shape[2] // (3 * tracks), 0)
            acc = y if acc is None else ops.axpby(acc, y, 1.0, 1.0)
    return {output_format: ops.axpby(acc, acc, 1.0 / 16.0, 0.0)}


def move_avg(preds, seg_lens, chunklen_sec, hoplen_sec, label_res=0.1):
    """post_processing(method='move_avg'): preds f32 [sum of chunks, chunk frames, D] in recording order, seg_lens = label
    frames per recording. Returns [1, sum of padded lengths, D] like the reference."""
    if chunklen_sec % hoplen_sec != 0:
        raise AssertionError("test_chunklen_sec % test_hoplen_sec == 0")
    hop = int(hoplen_sec / label_res)
    per_chunk = int(chunklen_sec / label_res)
    preds = preds.contiguous().float()
    outs, ind = [], 0
    for seg_len in seg_lens:
        num_chunks = int(np.ceil((seg_len - chunklen_sec / label_res) / hop)) + 1
        valid = int(np.ceil(seg_len / hop))
        tgt = int(np.ceil(seg_len / per_chunk) * per_chunk)
        outs.append(ops.move_avg(preds[ind:ind + num_chunks], hop, valid * hop, tgt))
        ind += num_chunks
    return torch.cat(outs, 0).unsqueeze(0)


def multi_accdoa_to_dcase_polar(pred, nb_classes, sed_threshold=0.5, unify_deg=15.0):
    """get_multi_accdoa_labels + multi_accdoa_to_dcase_format + convert_output_format_cartesian_to_polar for the frames of
    pred f32 [frames, 9C]: {frame: [[class, azimuth_deg, elevation_deg], ...]}."""
    events, counts = ops.decode_maccdoa(pred.contiguous().float(), nb_classes, sed_threshold, unify_deg)
    ev, cn = events.cpu().numpy(), counts.cpu().numpy()
    out = {}
    frames, classes = np.nonzero(cn)
    for f, c in zip(frames, classes):
        lst = out.setdefault(int(f), [])
        for k in range(cn[f, c]):
            x, y, z = ev[f, c, k]
            lst.append([int(c), math.atan2(y, x) * 180 / math.pi, math.atan2(z, math.sqrt(x * x + y * y)) * 180 / math.pi])
    return out


def accdoa_to_dcase_polar(pred, nb_classes, sed_threshold=0.5, max_ov=3):
    """get_accdoa_labels + accdoa_label_to_dcase_format + cartesian -> polar for pred f32 [frames, 3C]."""
    pred = pred.contiguous().float()
    sed = ops.decode_accdoa(pred, nb_classes, sed_threshold, max_ov).cpu().numpy()
    p = pred.cpu().numpy()
    out = {}
    for f, c in zip(*np.nonzero(sed)):
        x, y, z = p[f, c], p[f, c + nb_classes], p[f, c + 2 * nb_classes]
        out.setdefault(int(f), []).append([int(c), math.atan2(y, x) * 180 / math.pi, math.atan2(z, math.sqrt(x * x + y * y)) * 180 / math.pi])
    return out


def einv2_to_dcase(pred_sed, pred_doa, sed_threshold=0.5):
    """pred_aggregation's einv2 branch (components/model_module.py:191-204) + convert_to_dcase_format_polar (:229-233) +
    track_to_dcase_format (utils/data_utilities.py:154-177): sed logits [frames, 3, C], doa [frames, 3, 3] ->
    {frame: [[class, azimuth_deg, elevation_deg], ...]} (integer degrees, tracks in order). The sigmoid / arg-max / threshold
    are evaluated on the device; per track only the top class can be active."""
    p = torch.sigmoid(pred_sed.float())
    top_v, top_i = p.max(dim=-1)                                         # [frames, 3]
    active = (top_v > sed_threshold).cpu().numpy()
    cls = top_i.cpu().numpy()
    d = pred_doa.float().cpu().numpy()
    azi = np.arctan2(d[..., 1], d[..., 0])
    ele = np.arctan2(d[..., 2], np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2))
    out = {}
    for f, t in zip(*np.nonzero(active)):
        out.setdefault(int(f), []).append([int(cls[f, t]), int(np.around(azi[f, t] * 180 / np.pi)), int(np.around(ele[f, t] * 180 / np.pi))])
    return out


def load_output_format_file(path):
    """utils/data_utilities.py:67-88: DCASE CSV -> {frame: [[class, azimuth, elevation], ...]}; rows of 4 fields
    (frame, class, azi, ele) or 5 / 6 / 7 fields (frame, class, track, azi, ele[, distance[, mids]])."""
    out = {}
    with open(path) as f:
        for line in f:
            item = [v for v in line.strip().split(',') if v != '']
            if not item:
                continue
            frame = int(float(item[0]))
            if len(item) == 4:
                ev = [int(float(item[1])), float(item[2]), float(item[3])]
            elif len(item) in (5, 6, 7):
                ev = [int(float(item[1])), float(item[3]), float(item[4])]
            else:
                continue
            out.setdefault(frame, []).append(ev)
    return out


def write_output_format_file(path, output_dict):
    """utils/data_utilities.py:91-104: DCASE CSV rows `frame,class,azimuth,elevation` (integers). Frames are written in
    ascending order (the reference writes them in its dictionary's insertion order)."""
    with open(path, 'w') as f:
        for frame in output_dict:
            for value in output_dict[frame]:
                f.write('{},{},{},{}\n'.format(int(frame), int(value[0]), int(value[1]), int(value[2])))
