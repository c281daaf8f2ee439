"""HBM-resident clip store and device-side chunking — the device part of the reference's data layer:
`utils/data_utilities.py:6-64` segment_index (the index rows `begin,end,pad_before,pad_after` of the chunk CSVs),
`data/data.py:7-15` load_audio + the padding of :75-77, and the label synthesis of :87-93 / :207-213.

The reference reads a float32 slice per chunk from FLAC/WAV files through soundfile in DataLoader workers; at the MI355X
step rate (≈ 7 500 ten-second chunks/s per GPU) that is ≈ 29 GB/s of fp32 audio per GPU. Here the clips are decoded ONCE
to 16-bit PCM and kept in HBM (STARSS23 dev: 5 GB); a training batch is cut, padded and converted by one kernel launch
from a table of index rows. The index CSV of `extract_index` is written and read here (write_index_csv / read_index_csv, pinned to
rows the reference's own function wrote: tests/golden/index.npz). Recordings come from RIFF PCM16 WAV (read_wav_pcm16) or FLAC (data/flac.py:
own decoder); labels from the DCASE metadata CSVs (data/labels.py) or from the reference's HDF5 label files (data/hdf5_lite.py: own reader).
"""
import struct

import numpy as np
import torch

from .. import _lib


def segment_index(x_len, chunklen, hoplen, last_frame_always_paddding=False):
    """utils/data_utilities.py:6-64 on a length instead of an array: ([(begin, end), ...], [(pad_before, pad_after), ...]).
    Full chunks at multiples of hoplen; a trailing remainder of at least half a chunk becomes a zero-padded chunk, a
    shorter one is covered by a last chunk aligned to the end (unless `last_frame_always_paddding`)."""
    if x_len < chunklen:
        return [(0, x_len)], [(0, chunklen - x_len)]
    n_frames = 1 + (x_len - chunklen) // hoplen
    idx = [(n * hoplen, n * hoplen + chunklen) for n in range(n_frames)]
    pad = [(0, 0)] * n_frames
    if (n_frames - 1) * hoplen + chunklen == x_len:
        return idx, pad
    rest = x_len - n_frames * hoplen
    if last_frame_always_paddding or rest >= chunklen // 2:
        idx.append((n_frames * hoplen, x_len)); pad.append((0, chunklen - rest))
    else:
        idx.append((x_len - chunklen, x_len)); pad.append((0, 0))
    return idx, pad


def write_index_csv(path, recordings, chunklen, hoplen, last_frame_always_paddding=False):
    """The index file of preproc/preprocess.py:430-479 (`extract_index`, data_type 'wav'): one row `recording path,begin,end,pad_before,
    pad_after` per chunk, recordings in the given order (the reference sorts its glob). recordings: [(path, frames), ...]; chunklen /
    hoplen in samples. The reference writes the train file with the short-remainder rule and the test file with
    last_frame_always_paddding=True (:433)."""
    with open(path, 'w') as f:
        for rec, frames in recordings:
            idx, pad = segment_index(int(frames), chunklen, hoplen, last_frame_always_paddding)
            for (b, e), (pb, pa) in zip(idx, pad):
                f.write(f'{rec},{b},{e},{pb},{pa}\n')


def read_index_csv(path):
    """Rows of an index file as data/components/data.py:38-45 reads them: [(recording path, begin, end, pad_before, pad_after), ...]
    (the four integers are the LAST four comma-separated fields: a path may itself contain commas)."""
    rows = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            line = line.rstrip('\n')
            if not line:
                continue
            parts = line.rsplit(',', 4)
            if len(parts) != 5:
                raise ValueError(f'{path}:{ln}: expected `path,begin,end,pad_before,pad_after`, got {line!r}')
            try:
                b, e, pb, pa = (int(v) for v in parts[1:])
            except ValueError:
                raise ValueError(f'{path}:{ln}: non-integer index field in {line!r}') from None
            if b < 0 or e < b or pb < 0 or pa < 0:
                raise ValueError(f'{path}:{ln}: inconsistent index row {line!r}')
            rows.append((parts[0], b, e, pb, pa))
    return rows


def read_wav_pcm16(path):
    """Minimal RIFF/WAVE reader for 16-bit PCM: (int16 [frames, channels], sample_rate)."""
    with open(path, 'rb') as f:
        data = f.read()
    if data[:4] != b'RIFF' or data[8:12] != b'WAVE':
        raise ValueError(f'{path}: not a RIFF/WAVE file')
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack('<I', data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b'fmt ':
            fmt = struct.unpack('<HHIIHH', body[:16])
        elif cid == b'data':
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f'{path}: missing fmt / data chunk')
    tag, channels, rate, _, _, bits = fmt
    if tag not in (1, 0xFFFE) or bits != 16:
        raise NotImplementedError(f'{path}: only 16-bit PCM is supported (format tag {tag}, {bits} bits)')
    return np.frombuffer(pcm, dtype='<i2').reshape(-1, channels).copy(), rate


class DeviceClipStore:
    """All clips of a dataset split as ONE int16 tensor [total_frames, channels] in HBM plus per-clip frame offsets."""

    def __init__(self, device, channels=4):
        self.device, self.channels = torch.device(device), channels
        self._parts, self.offsets, self.lengths, self.names = [], [], [], {}
        self._total, self.pcm = 0, None

    def add_clip(self, name, pcm):
        """pcm: int16 array / tensor [frames, channels]."""
        pcm = torch.as_tensor(np.ascontiguousarray(pcm) if isinstance(pcm, np.ndarray) else pcm)
        if pcm.dtype != torch.int16 or pcm.ndim != 2 or pcm.shape[1] != self.channels:
            raise ValueError(f'clip must be int16 [frames, {self.channels}]')
        self.names[name] = len(self.offsets)
        self.offsets.append(self._total); self.lengths.append(pcm.shape[0])
        self._parts.append(pcm)
        self._total += pcm.shape[0]
        self.pcm = None

    def add_wav(self, path, sample_rate=None):
        """sample_rate: cfg.data.sample_rate — a recording at another rate would be chunked and labelled on the wrong time base."""
        pcm, rate = read_wav_pcm16(path)
        if sample_rate is not None and int(rate) != int(sample_rate):
            raise ValueError(f'{path}: sample rate {rate} Hz, expected cfg.data.sample_rate = {sample_rate} Hz (no resampler on this path)')
        self.add_clip(str(path), pcm)

    def add_flac(self, path, sample_rate=None):
        """A FLAC recording (the reference's synthetic datasets: data/components/data.py:81; read there by soundfile). Decoded on the host
        by libpseld_host.so (data/flac.py: frame CRCs and the stream's MD5 signature checked); 16-bit streams only - the store holds
        PCM16, and `sf.read(dtype='float32')` of a 16-bit stream is exactly int16 / 32768, the scaling the chunk kernel applies."""
        from . import flac
        with open(path, 'rb') as f:
            data = f.read()
        pcm, info = flac.decode_flac(data)
        if info['bits_per_sample'] != 16:
            raise NotImplementedError(f"{path}: {info['bits_per_sample']}-bit FLAC (the clip store holds 16-bit PCM)")
        if sample_rate is not None and int(info['sample_rate']) != int(sample_rate):
            raise ValueError(f"{path}: sample rate {info['sample_rate']} Hz, expected cfg.data.sample_rate = {sample_rate} Hz (no resampler on this path)")
        self.add_clip(str(path), pcm.astype(np.int16))

    def add_audio(self, path, sample_rate=None):
        """By extension: .flac -> add_flac, anything else -> add_wav (RIFF PCM16)."""
        return self.add_flac(path, sample_rate) if str(path).lower().endswith('.flac') else self.add_wav(path, sample_rate)

    def finalize(self):
        if self.pcm is None:
            _lib.require_gpu()
            self.pcm = torch.cat([p.to(self.device) for p in self._parts], 0).contiguous()
            self._parts = [self.pcm]
        return self

    def index_rows(self, chunklen, hoplen, last_frame_always_paddding=False):
        """The rows of the reference's `{dataset}_{chunk}sChunklen_{hop}sHoplen_*.csv`: (clip name, begin, end, pad_before,
        pad_after) for every clip of the store."""
        rows = []
        for name, i in self.names.items():
            idx, pad = segment_index(self.lengths[i], chunklen, hoplen, last_frame_always_paddding)
            rows += [(name, b, e, pb, pa) for (b, e), (pb, pa) in zip(idx, pad)]
        return rows

    def chunks(self, rows, chunk_len):
        """rows: [(clip name, begin, end, pad_before, pad_after), ...] -> f32 [n, channels, chunk_len] on the device."""
        self.finalize()
        seg = torch.tensor([[self.offsets[self.names[r[0]]], r[1], r[2], r[3], r[4]] for r in rows], dtype=torch.int64, device=self.device)
        for r in rows:
            if r[3] + (r[2] - r[1]) + r[4] != chunk_len or r[1] < 0 or r[2] > self.lengths[self.names[r[0]]]:
                raise ValueError(f'bad index row {r} for chunk length {chunk_len}')
        out = torch.empty((len(rows), self.channels, chunk_len), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().pseld_pcm16_chunks(_lib.ptr(self.pcm), _lib.ptr(seg), _lib.ptr(out), len(rows), self.channels, chunk_len,
                                                 _lib.stream_ptr()), "pseld_pcm16_chunks")
        return out


def polar_labels(se, azi, ele):
    """(se bool/u8, azi int16, ele int8) of shape [T, tracks, C] (ADPIT) or [T, C] (ACCDOA) on the device ->
    f32 [T, tracks, 4, C] (data.py:207-213) or [T, 4*C] (data.py:87-93: concatenation of se, x, y, z)."""
    if not se.is_cuda:
        raise _lib.PseldError("labels must live on the MI355X (no CPU fallback)")
    se8, a16, e8 = se.to(torch.uint8).contiguous(), azi.to(torch.int16).contiguous(), ele.to(torch.int8).contiguous()
    C = se.shape[-1]
    rows_tracks = se.numel() // C
    out = torch.empty((rows_tracks, 4, C), dtype=torch.float32, device=se.device)
    _lib.check(_lib.lib().pseld_polar_labels(_lib.ptr(se8), _lib.ptr(a16), _lib.ptr(e8), _lib.ptr(out), rows_tracks, C, _lib.stream_ptr()),
               "pseld_polar_labels")
    return out.view(*se.shape[:-1], 4, C) if se.ndim == 3 else out.view(se.shape[0], 4 * C)


def generate_spatial_samples(audio, method, rng=None, **labels):
    """Batch-wise device mirror of data/data.py:17-59 (the mono_adapter recipe, single-source targets only): every mono clip
    becomes the FOA encoding of a source at a random direction and its label is rewritten to that direction.
    audio f32 [N, L] or [N, ch, L] (channel 0 = the mono signal) on the device; labels: sed_label [N,T,3,C] + doa_label
    (einv2), accdoa_label [N,T,4C] (accdoa), adpit_label [N,T,6,4,C] (multi_accdoa). `rng` (default numpy's global generator,
    as the reference) is asked for randint(-180, 180) then randint(-90, 90) per sample, in batch order.
    Returns (foa [N,4,L], label...) like the reference."""
    rng = np.random if rng is None else rng
    N = audio.shape[0]
    mono = audio if audio.ndim == 2 else audio[:, 0]
    if not mono.is_cuda or mono.dtype != torch.float32 or mono.stride(-1) != 1:
        raise _lib.PseldError("generate_spatial_samples: audio must be fp32 on the MI355X with unit stride along time")
    L = mono.shape[-1]
    ang = np.array([[rng.randint(-180, 180), rng.randint(-90, 90)] for _ in range(N)], np.float64).reshape(N, 2)
    azi, ele = np.deg2rad(ang[:, 0]), np.deg2rad(ang[:, 1])
    xyz = np.stack((np.cos(azi) * np.cos(ele), np.sin(azi) * np.cos(ele), np.sin(ele)), 1)
    xyz_d = torch.from_numpy(xyz).to(mono.device)
    foa = torch.empty((N, 4, L), dtype=torch.float32, device=mono.device)
    lib, st = _lib.lib(), _lib.stream_ptr()
    _lib.check(lib.pseld_spatialize_mono(_lib.ptr(mono), mono.stride(0), _lib.ptr(xyz_d), _lib.ptr(foa), N, L, st), "pseld_spatialize_mono")

    def coef_label(lab, outer, inner, first):
        lab = lab.contiguous().float()
        coef = torch.from_numpy(np.concatenate((np.full((N, 1), first), xyz), 1)).to(mono.device)
        out = torch.empty_like(lab)
        _lib.check(lib.pseld_spatial_label(_lib.ptr(lab), _lib.ptr(out), _lib.ptr(coef), N, outer, inner, st), "pseld_spatial_label")
        return out

    if method == 'einv2':
        sed = labels['sed_label'].contiguous().float()
        doa = torch.empty(sed.shape[:3] + (3,), dtype=torch.float32, device=mono.device)
        _lib.check(lib.pseld_spatial_doa_label(_lib.ptr(sed), _lib.ptr(xyz_d), _lib.ptr(doa), N, sed.shape[1], sed.shape[2], sed.shape[3], st),
                   "pseld_spatial_doa_label")
        return foa, labels['sed_label'], doa
    if method == 'accdoa':
        lab = labels['accdoa_label']
        C = lab.shape[-1] // 4
        return foa, coef_label(lab, lab.shape[1], C, 0.0)
    if method == 'multi_accdoa':
        lab = labels['adpit_label']
        return foa, coef_label(lab, lab.shape[1] * lab.shape[2], lab.shape[4], 1.0)
    raise NotImplementedError(method)


class DeviceSELDDataset:
    """The training split as the reference's datasets see it (data/components/data.py:12-116 + data/data.py:62-252), resident in
    HBM: recordings as 16-bit PCM (`DeviceClipStore`), labels as the compact (se, azimuth, elevation) arrays of the label
    files (or, for EINV2, the track-wise arrays), the index rows of `extract_index`. `batch(indices)` returns what the
    reference's DataLoader would collate for those rows — {'filename', 'data' f32 [B,4,L], '<method> label' [B,100,...], 'ov'} —
    with ONE chunk-cutting launch and one label-synthesis launch for the whole batch.
    method: 'multi_accdoa' | 'accdoa' | 'einv2'; metas: {recording name: path of its DCASE metadata CSV} - or label_h5: the label
    file the reference's preprocessing wrote for the method (`.../adpit/<type>/<dataset>.h5`, `accdoa/...`, `track/...`:
    preprocess.py:63-65), read through data/hdf5_lite.py by recording stem exactly as data/data.py:92-96, 159-161, 220-224 do."""

    def __init__(self, store, metas, method, num_classes, sample_rate=24000, chunklen_sec=10, hoplen_sec=10, label_res=0.1, max_ov=3,
                 mono_adapter=False, rng=None, index_csv=None, label_h5=None):
        from .. import inference
        self.mono_adapter, self.rng = mono_adapter, rng          # cfg.adapt.method == 'mono_adapter' (data.py:109-111,169-171,223-225)
        from . import labels as L
        self.store, self.method, self.C = store.finalize(), method, num_classes
        self.chunk_len = int(chunklen_sec * sample_rate)
        self.ppp = int(sample_rate * label_res)                       # points_per_predictions
        self.frames = int(chunklen_sec / label_res)
        self.max_ov = max_ov
        if index_csv is not None:
            # the reference's own index file (read_index_csv): its rows name recordings by path; the store knows them by that path or by
            # its file name (an index written on another machine carries that machine's directories)
            import os
            by_base = {os.path.basename(n): n for n in store.names}
            by_stem = {os.path.splitext(os.path.basename(n))[0]: n for n in store.names}      # (.wav rows, .flac recordings: data/components/data.py:81)
            self.rows = []
            for rec, b, e, pb, pa in read_index_csv(index_csv):
                name = rec if rec in store.names else by_base.get(os.path.basename(rec), by_stem.get(os.path.splitext(os.path.basename(rec))[0]))
                if name is None:
                    raise KeyError(f'{index_csv}: recording {rec!r} is not in the clip store')
                self.rows.append((name, b, e, pb, pa))
        else:
            self.rows = store.index_rows(self.chunk_len, int(hoplen_sec * sample_rate))
        dev = store.device
        self.labels = {}
        if label_h5 is not None:
            import os
            from . import hdf5_lite
            with hdf5_lite.File(label_h5) as hf:
                for name in store.names:
                    fn = os.path.splitext(os.path.basename(name))[0]
                    if method == 'einv2':
                        arrs = (hf[f'{fn}/sed_label'][...], hf[f'{fn}/doa_label'][...].astype('float32'))
                    else:
                        grp = 'adpit' if method == 'multi_accdoa' else 'accdoa'
                        arrs = tuple(hf[f'{fn}/{grp}/{k}'][...] for k in ('se', 'azi', 'ele'))
                    self.labels[name] = tuple(torch.from_numpy(a).to(dev) for a in arrs)
        for name in (store.names if label_h5 is None else ()):
            if method == 'einv2':
                sed, doa = L.track_labels(L.read_meta_rows(metas[name]), num_classes, max_ov)
                self.labels[name] = (torch.from_numpy(sed).to(dev), torch.from_numpy(doa).to(dev))
            else:
                meta = inference.load_output_format_file(metas[name])
                if method == 'multi_accdoa':
                    arrs = L.adpit_labels(meta, num_classes)
                else:
                    arrs = L.accdoa_labels(meta, int(L.read_meta_rows(metas[name])[-1, 0]) + 1, num_classes)
                self.labels[name] = tuple(torch.from_numpy(a).to(dev) for a in arrs)

    def __len__(self):
        return len(self.rows)

    def _label_slices(self, rows, k):
        """Label frames [begin / ppp, end / ppp) of array k for every row, zero-padded to the chunk's frame count (data.py:197-222)."""
        out = []
        for r in rows:
            a = self.labels[r[0]][k]
            sl = a[int(r[1] / self.ppp):int(r[2] / self.ppp)]
            if sl.shape[0] < self.frames:
                sl = torch.cat((sl, torch.zeros((self.frames - sl.shape[0],) + tuple(sl.shape[1:]), dtype=sl.dtype, device=sl.device)), 0)
            out.append(sl[:self.frames])
        return torch.stack(out, 0)

    def batch(self, indices):
        rows = [self.rows[int(i)] for i in indices]
        sample = {'filename': [r[0] for r in rows], 'data': self.store.chunks(rows, self.chunk_len)}
        B = len(rows)
        if self.method == 'einv2':
            sed = self._label_slices(rows, 0)[:, :, :self.max_ov].float()
            sample['sed_label'], sample['doa_label'] = sed, self._label_slices(rows, 1)[:, :, :self.max_ov].float()
            act = sed.sum(dim=(2, 3))
        else:
            se, azi, ele = (self._label_slices(rows, k) for k in range(3))
            lab = polar_labels(se.reshape(B * self.frames, *se.shape[2:]), azi.reshape(B * self.frames, *azi.shape[2:]),
                               ele.reshape(B * self.frames, *ele.shape[2:]))
            if self.method == 'multi_accdoa':
                sample['adpit_label'] = lab.view(B, self.frames, 6, 4, self.C)
                act = se.float().sum(dim=(2, 3))
            else:
                sample['accdoa_label'] = lab.view(B, self.frames, 4 * self.C)[:, :, self.C:]      # data.py:95: the xyz blocks
                act = se.float().sum(dim=2)
        if self.mono_adapter:      # every chunk becomes the FOA encoding of its first channel at a random direction, labels rewritten
            keys = [k for k in sample if k.endswith('_label')]
            if self.method == 'accdoa':                           # the recipe works on the full (se | x | y | z) label, then drops se again
                full = generate_spatial_samples(sample['data'], 'accdoa', self.rng, accdoa_label=lab.view(B, self.frames, 4 * self.C))
                sample['data'], sample['accdoa_label'] = full[0], full[1][:, :, self.C:]
            else:
                res = generate_spatial_samples(sample['data'], self.method, self.rng, **{k: sample[k] for k in keys})
                sample['data'] = res[0]
                for k, v in zip(keys, res[1:]):
                    sample[k] = v
        sample['ov'] = [str(max(int(v), 1)) for v in act.max(dim=1).values.tolist()]             # data.py:229: one host sync per batch
        return sample
