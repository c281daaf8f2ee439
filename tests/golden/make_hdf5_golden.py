"""HDF5 fixtures for pseldnets_amd/data/hdf5_lite.py, written by the HDF5 C LIBRARY (libhdf5 1.10 of this image, /opt/conda/lib, through
ctypes - h5py itself is absent) with the calls h5py makes for the reference's `hf.create_dataset(name=f'{fn}/adpit/se', data=..., dtype=...)`
(preproc/preprocess.py:128-129, 207-209, 457-459, 560): default file creation (libver earliest: superblock 0, symbol-table groups),
intermediate groups created with the link, contiguous little-endian datasets, object time tracking off, np.bool_ as h5py's boolean ENUM
over int8 {FALSE = 0, TRUE = 1}. The arrays are the reference's own label arrays of tests/golden/labels.npz (made by make_golden.py from
/root/reference's extract_*_label functions) plus a float32 `feature` array; tests/test_hdf5_lite.py reads the files back.
Runs in the build container only:   python tests/golden/make_hdf5_golden.py"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'hdf5')
L = C.CDLL('/opt/conda/lib/libhdf5.so.103')
hid = C.c_int64
for fn, res, args in (('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Pcreate', hid, [hid]), ('H5Screate_simple', hid, [C.c_int, C.c_void_p, C.c_void_p]),
                      ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ('H5Dwrite', C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                      ('H5Tenum_create', hid, [hid]), ('H5Tenum_insert', C.c_int, [hid, C.c_char_p, C.c_void_p]), ('H5Tcopy', hid, [hid]),
                      ('H5Pset_create_intermediate_group', C.c_int, [hid, C.c_uint]), ('H5Pset_obj_track_times', C.c_int, [hid, C.c_int]),
                      ('H5Pset_libver_bounds', C.c_int, [hid, C.c_int, C.c_int]), ('H5Pset_chunk', C.c_int, [hid, C.c_int, C.c_void_p]),
                      ('H5Fclose', C.c_int, [hid]), ('H5Dclose', C.c_int, [hid]), ('H5Sclose', C.c_int, [hid]), ('H5Tclose', C.c_int, [hid]), ('H5Pclose', C.c_int, [hid])):
    f = getattr(L, fn); f.restype = res; f.argtypes = args
assert L.H5open() >= 0
g = lambda name: hid.in_dll(L, name).value
NATIVE = {np.dtype('int8'): g('H5T_NATIVE_INT8_g'), np.dtype('int16'): g('H5T_NATIVE_INT16_g'), np.dtype('int32'): g('H5T_NATIVE_INT32_g'),
          np.dtype('int64'): g('H5T_NATIVE_INT64_g'), np.dtype('uint8'): g('H5T_NATIVE_UINT8_g'), np.dtype('float32'): g('H5T_NATIVE_FLOAT_g'),
          np.dtype('float64'): g('H5T_NATIVE_DOUBLE_g')}


def bool_type():
    t = L.H5Tenum_create(NATIVE[np.dtype('int8')])
    for name, v in ((b'FALSE', 0), (b'TRUE', 1)):
        val = C.c_int8(v)
        assert L.H5Tenum_insert(t, name, C.byref(val)) >= 0
    return t


def write(path, datasets, latest=False, chunked=()):
    """datasets: {name: array}; what `h5py.File(path, 'w')` + `create_dataset(name, data=array, dtype=array.dtype)` do."""
    fapl = 0
    if latest:
        fapl = L.H5Pcreate(g('H5P_CLS_FILE_ACCESS_ID_g'))
        assert L.H5Pset_libver_bounds(fapl, 2, 2) >= 0            # H5F_LIBVER_V110 .. latest of this library
    f = L.H5Fcreate(path.encode(), 2, 0, fapl)                    # H5F_ACC_TRUNC
    assert f >= 0, path
    lcpl = L.H5Pcreate(g('H5P_CLS_LINK_CREATE_ID_g'))
    assert L.H5Pset_create_intermediate_group(lcpl, 1) >= 0
    for name, arr in datasets.items():
        arr = np.ascontiguousarray(arr)
        is_bool = arr.dtype == np.bool_
        mem = arr.view(np.int8) if is_bool else arr
        t = bool_type() if is_bool else L.H5Tcopy(NATIVE[arr.dtype])
        dims = (C.c_uint64 * max(arr.ndim, 1))(*arr.shape)
        s = L.H5Screate_simple(arr.ndim, dims, None)
        dcpl = L.H5Pcreate(g('H5P_CLS_DATASET_CREATE_ID_g'))
        assert L.H5Pset_obj_track_times(dcpl, 0) >= 0
        if name in chunked:
            assert L.H5Pset_chunk(dcpl, arr.ndim, (C.c_uint64 * arr.ndim)(*[max(1, d // 2) for d in arr.shape])) >= 0
        d = L.H5Dcreate2(f, name.encode(), t, s, lcpl, dcpl, 0)
        assert d >= 0, name
        if arr.size:
            assert L.H5Dwrite(d, t, 0, 0, 0, mem.ctypes.data_as(C.c_void_p)) >= 0
        L.H5Dclose(d); L.H5Pclose(dcpl); L.H5Sclose(s); L.H5Tclose(t)
    L.H5Pclose(lcpl)
    assert L.H5Fclose(f) >= 0


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    lab = np.load(os.path.join(HERE, 'labels.npz'))
    print(sorted(lab.files))
    fns = ['mix0', 'mix1']                                         # the two recordings of labels.npz (make_golden.py: the reference's functions)
    adpit, accdoa, track = {}, {}, {}
    for fn in fns:
        for k in ('se', 'azi', 'ele'):
            adpit[f'{fn}/adpit/{k}'] = lab[f'{fn}__adpit__{k}']
            accdoa[f'{fn}/accdoa/{k}'] = lab[f'{fn}__accdoa__{k}']
        track[f'{fn}/sed_label'] = lab[f'{fn}__sed_label']
        track[f'{fn}/doa_label'] = lab[f'{fn}__doa_label']
    # 40 more recordings: enough links that the root group's B-tree gets more than one symbol node
    for j in range(40):
        adpit[f'fold3_room{j:02d}_mix/adpit/se'] = lab['mix0__adpit__se'][j:j + 3]
    write(os.path.join(OUT, 'adpit.h5'), adpit)
    write(os.path.join(OUT, 'accdoa.h5'), accdoa)
    write(os.path.join(OUT, 'track.h5'), track)
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((7, 25, 64)).astype(np.float32)
    write(os.path.join(OUT, 'feature.h5'), {'feature': feat, 'scalar': np.float64(3.5) * np.ones((), np.float64), 'empty': np.zeros((0, 4), np.int16)})
    write(os.path.join(OUT, 'latest.h5'), {f'{fns[0]}/adpit/se': adpit[f'{fns[0]}/adpit/se'], f'{fns[0]}/adpit/azi': adpit[f'{fns[0]}/adpit/azi']}, latest=True)
    write(os.path.join(OUT, 'chunked.h5'), {'feature': feat}, chunked=('feature',))
    np.savez_compressed(os.path.join(OUT, 'expected.npz'), feature=feat, **{k.replace('/', '|'): v for d in (adpit, accdoa, track) for k, v in d.items() if not k.startswith('fold3')})
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
