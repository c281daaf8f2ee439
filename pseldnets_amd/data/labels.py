"""Label extraction from DCASE metadata — host-side mirror of the reference's offline preprocessing
(`preproc/preprocess.py`: extract_track_label :80-133, extract_accdoa_label :176-212, extract_adpit_label :346-461): the functions
return the arrays the reference stores as `{fn}/accdoa/{se,azi,ele}`, `{fn}/adpit/{se,azi,ele}` and `{fn}/{sed_label,doa_label}` in its
HDF5 label files (those files themselves are read by `data/hdf5_lite.py`, `DeviceSELDDataset(label_h5=...)`; h5py is absent from this
image, so nothing here writes the container); `data/ingest.py:polar_labels` turns the
(se, azi, ele) triples into the float training labels on the device. Plain numpy: this is preprocessing, not the hot path."""
import numpy as np


def read_meta_rows(path):
    """The rows of a DCASE metadata CSV as a float array [n, columns] (pd.read_csv(header=None).values in the reference)."""
    rows = []
    with open(path) as f:
        for line in f:
            item = [v for v in line.strip().split(',') if v != '']
            if item:
                rows.append([float(v) for v in item])
    return np.array(rows, np.float64)


def accdoa_labels(meta, num_frames, num_classes):
    """preprocess.py:180-192. meta: {frame: [[class, azimuth, elevation], ...]} (inference.load_output_format_file);
    returns (se bool [T, C], azi int16 [T, C], ele int8 [T, C]); a later event of the same class in a frame overwrites."""
    se = np.zeros((num_frames, num_classes), dtype=np.bool_)
    azi = np.zeros((num_frames, num_classes), dtype=np.int16)
    ele = np.zeros((num_frames, num_classes), dtype=np.int8)
    for frame, events in meta.items():
        if frame < num_frames:
            for ev in events:
                c = int(ev[0])
                se[frame, c], azi[frame, c], ele[frame, c] = 1, ev[1], ev[2]
    return se, azi, ele


def adpit_labels(meta, num_classes):
    """preprocess.py:352-444. The six ADPIT slots (A0 | B0 B1 | C0 C1 C2): per frame the events are sorted by class (stable);
    a class with one event fills A0, with two B0 / B1, with three or more the first three fill C0 / C1 / C2.
    The number of label frames is the LAST key of `meta` + 1, as in the reference. Returns (se bool, azi int16, ele int8) [T, 6, C]."""
    num_frames = list(meta.keys())[-1] + 1
    se = np.zeros((num_frames, 6, num_classes), dtype=np.bool_)
    azi = np.zeros((num_frames, 6, num_classes), dtype=np.int16)
    ele = np.zeros((num_frames, 6, num_classes), dtype=np.int8)
    first_slot = {1: 0, 2: 1}
    for frame, events in meta.items():
        if frame >= num_frames:
            continue
        events = sorted(events, key=lambda e: e[0])
        i = 0
        while i < len(events):
            j = i
            while j + 1 < len(events) and events[j + 1][0] == events[i][0]:
                j += 1
            group = events[i:j + 1]
            slot0 = first_slot.get(len(group), 3)
            for k, ev in enumerate(group[:3]):
                c = int(ev[0])
                se[frame, slot0 + k, c], azi[frame, slot0 + k, c], ele[frame, slot0 + k, c] = 1, ev[1], ev[2]
            i = j + 1
    return se, azi, ele


def track_labels(rows, num_classes, max_polyphony=3):
    """preprocess.py:94-127. rows: metadata rows (frame, class, track number, azimuth, elevation[, ...]) in file order; every event
    takes the first free track of its frame (events beyond max_polyphony are dropped). Returns (sed bool [T, P, C], doa f32 [T, P, 3])."""
    num_frames = int(rows[-1, 0]) + 1
    sed = np.zeros((num_frames, max_polyphony, num_classes), dtype=np.bool_)
    doa = np.zeros((num_frames, max_polyphony, 3))
    used = np.zeros(num_frames, dtype=np.int64)
    for row in rows:
        frame, cls = int(row[0]), int(row[1])
        t = used[frame]
        if t >= max_polyphony:
            continue
        azi, elev = row[3] * np.pi / 180, row[4] * np.pi / 180
        sed[frame, t, cls] = 1.0
        doa[frame, t, :] = np.cos(elev) * np.cos(azi), np.cos(elev) * np.sin(azi), np.sin(elev)
        used[frame] += 1
    return sed, doa.astype(np.float32)
