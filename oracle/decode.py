"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy / plain loops) of the reference's inference-side decoding
(/root/reference/src/utils/data_utilities.py:197-244,273-388 and models/components/model_module.py:269-329). Imported only
by tests/. Pinned by tests/golden/decode.npz (tests/golden/make_golden.py:gen_decode calls the reference's own functions and
its BaseModelModule.post_processing with a stand-in `self`)."""
import numpy as np
import torch


def ang_dist_deg(a, b):
    """data_utilities.py:212-231 on fp32 triples."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    n1 = np.sqrt((a ** 2).sum() + np.float32(1e-10)); n2 = np.sqrt((b ** 2).sum() + np.float32(1e-10))
    d = np.clip(((a / n1) * (b / n2)).sum(), -1, 1)
    return np.arccos(d) * 180 / np.pi


def decode_multi_accdoa(pred, nb_classes, sed_threshold=0.5, unify=15):
    """get_multi_accdoa_labels (:273-300) + multi_accdoa_to_dcase_format (:302-388): pred [frames, 9C] ->
    {frame: [[class, x, y, z], ...]} with classes ascending and, inside a class, the reference's event order."""
    pred = np.asarray(pred, np.float32)
    C = nb_classes
    out = {}
    for f in range(pred.shape[0]):
        ev_f = []
        for c in range(C):
            e = []
            for k in range(3):
                v = pred[f, [(3 * k) * C + c, (3 * k + 1) * C + c, (3 * k + 2) * C + c]]
                if np.sqrt((v ** 2).sum()) > sed_threshold:
                    e.append(v)
            if len(e) == 1:
                ev_f.append([c, *e[0]])
            elif len(e) == 2:
                if ang_dist_deg(e[0], e[1]) < unify:
                    ev_f.append([c, *((e[0] + e[1]) / 2)])
                else:
                    ev_f += [[c, *e[0]], [c, *e[1]]]
            elif len(e) == 3:
                s01, s12, s02 = (int(ang_dist_deg(e[0], e[1]) < unify), int(ang_dist_deg(e[1], e[2]) < unify),
                                 int(ang_dist_deg(e[0], e[2]) < unify))
                s = s01 + s12 + s02
                if s == 0:
                    ev_f += [[c, *e[0]], [c, *e[1]], [c, *e[2]]]
                elif s == 1:
                    if s01:
                        ev_f += [[c, *((e[0] + e[1]) / 2)], [c, *e[2]]]
                    elif s12:
                        ev_f += [[c, *e[0]], [c, *((e[1] + e[2]) / 2)]]
                    else:
                        ev_f += [[c, *e[0]], [c, *((e[0] + e[2]) / 2)]]          # :371-377, kept as the reference has it
                else:
                    ev_f.append([c, *((e[0] + e[1] + e[2]) / 3)])
        if ev_f:
            out[f] = ev_f
    return out


def cartesian_to_polar(d):
    """convert_output_format_cartesian_to_polar (:197-210): degrees."""
    out = {}
    for f, evs in d.items():
        out[f] = [[e[0], np.arctan2(e[2], e[1]) * 180 / np.pi, np.arctan2(e[3], np.sqrt(e[1] ** 2 + e[2] ** 2)) * 180 / np.pi] for e in evs]
    return out


def decode_accdoa(pred, nb_classes, sed_threshold=0.5, max_ov=3):
    """get_accdoa_labels (:234-244): bool [frames, C]."""
    pred = torch.as_tensor(pred)
    C = nb_classes
    sed = torch.sqrt(pred[..., :C] ** 2 + pred[..., C:2 * C] ** 2 + pred[..., 2 * C:] ** 2)
    top_v, top_i = torch.topk(sed, max_ov, dim=-1, largest=True)
    z = torch.zeros_like(sed)
    z.scatter_(-1, top_i, top_v)
    return (z > sed_threshold).numpy()


ACS_TRANS = {(0, 1, 2): (1, 2, 3), (1, 0, 2): (3, 2, 1)}
ACS_SIGNS = [[1, 1, 1], [-1, 1, 1], [1, -1, 1], [-1, -1, 1], [1, 1, -1], [-1, 1, -1], [1, -1, -1], [-1, -1, -1]]


def acs(batch, standardize, forward, output_format):
    """components/model_module.py:271-300: 16 rotations / reflections of the FOA input, outputs mapped back and averaged."""
    outs = []
    for sx, sy, sz in ACS_SIGNS:
        for (xx, yy, zz), (s_x, s_y, s_z) in ACS_TRANS.items():
            x = torch.stack((batch[:, 0], sy * batch[:, s_x], sz * batch[:, s_y], sx * batch[:, s_z]), 1)
            y = forward(standardize(x))[output_format]
            B, T = y.shape[:2]
            y = y.reshape(B, T, 3, 3, -1) if output_format == 'multi_accdoa' else y.reshape(B, T, 3, -1)
            y = torch.stack((sx * y[..., 0, :], sy * y[..., 1, :], sz * y[..., 2, :]), -2)
            y = torch.stack((y[..., xx, :], y[..., yy, :], y[..., zz, :]), -2)
            outs.append(y.reshape(B, T, -1))
    return {output_format: torch.stack(outs).mean(0)}


def move_avg(preds, seg_lens, chunklen_sec, hoplen_sec, label_res=0.1):
    """components/model_module.py:302-329: preds [sum of chunks, chunk frames, D]; one stitched tensor per recording."""
    chunk_len = int(hoplen_sec / label_res)
    per_chunk = int(chunklen_sec / label_res)
    outs, ind = [], 0
    for seg_len in seg_lens:
        num_chunks = int(np.ceil((seg_len - chunklen_sec / label_res) / chunk_len)) + 1
        valid = int(np.ceil(seg_len / chunk_len))
        tgt = int(np.ceil(seg_len / per_chunk) * per_chunk)
        local = preds[ind:ind + num_chunks]
        blocks = []
        for i in range(valid):
            lo, hi = int(max(0, i - chunklen_sec // hoplen_sec + 1)), int(min(i + 1, num_chunks))
            blocks.append(torch.stack([local[j, (i - j) * chunk_len:(i - j + 1) * chunk_len] for j in range(lo, hi)], 0).mean(0))
        res = torch.cat(blocks, 0)
        res = torch.cat([res, torch.zeros(tgt - res.shape[0], *res.shape[1:])], 0) if res.shape[0] < tgt else res[:tgt]
        outs.append(res)
        ind += num_chunks
    return outs
