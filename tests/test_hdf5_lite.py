"""pseldnets_amd/data/hdf5_lite.py (the reader for the reference's HDF5 label / feature files: preproc/preprocess.py:88-129, 197-209,
449-459, 559-560 write them, data/data.py:82-96, 150-161, 210-224 read slices back) against files written by the HDF5 C library itself
with the calls h5py makes (tests/golden/hdf5/*.h5, tests/golden/make_hdf5_golden.py) from the reference's own label arrays
(tests/golden/labels.npz): every dataset bit for bit, the slicing forms the reference uses, groups with more links than one symbol
node holds, the 1.10 "latest" encoding of the same data, and a clear refusal for what the subset does not cover."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), 'golden')
H5 = os.path.join(G, 'hdf5')


def _lib():
    from pseldnets_amd.data import hdf5_lite
    return hdf5_lite


@pytest.mark.parametrize("fname,group", [('adpit.h5', 'adpit'), ('accdoa.h5', 'accdoa')])
def test_label_files_read_back_bit_for_bit(fname, group):
    H = _lib()
    lab = np.load(os.path.join(G, 'labels.npz'))
    with H.File(os.path.join(H5, fname)) as hf:
        assert {'mix0', 'mix1'} <= set(hf.keys()) and 'mix0' in hf and 'nope' not in hf
        for fn in ('mix0', 'mix1'):
            assert sorted(hf[fn][group].keys()) == ['azi', 'ele', 'se']
            for k, dt in (('se', np.bool_), ('azi', np.int16), ('ele', np.int8)):
                want = lab[f'{fn}__{group}__{k}']
                ds = hf[f'{fn}/{group}/{k}']
                assert ds.shape == want.shape and ds.dtype == np.dtype(dt) and len(ds) == want.shape[0]
                assert np.array_equal(ds[...], want) and ds[...].dtype == want.dtype
                # the reference's slicing: label frames [b, e) with the trailing axes whole (data.py:94-96, 222-224)
                for b, e in ((0, 60), (7, 19), (50, 80), (30, 30)):
                    assert np.array_equal(ds[b:e, ...], want[b:e]) and np.array_equal(ds[b:e], want[b:e])
                assert np.array_equal(ds[5], want[5]) and np.array_equal(ds[-1], want[-1]) and np.array_equal(np.asarray(ds), want)
                assert np.array_equal(ds[::2], want[::2])
        with pytest.raises(KeyError):
            hf['mix0/' + group + '/nothing']


def test_track_file_and_a_group_spread_over_several_symbol_nodes():
    H = _lib()
    lab = np.load(os.path.join(G, 'labels.npz'))
    with H.File(os.path.join(H5, 'track.h5')) as hf:
        for fn in ('mix0', 'mix1'):
            sed, doa = hf[f'{fn}/sed_label'], hf[f'{fn}/doa_label']
            assert sed.dtype == np.bool_ and doa.dtype == np.float32
            # data.py:160-161: frames [b, e), the first max_ov tracks
            assert np.array_equal(sed[10:40, :2], lab[f'{fn}__sed_label'][10:40, :2]) and np.array_equal(doa[10:40, :2], lab[f'{fn}__doa_label'][10:40, :2])
            assert np.array_equal(doa[...], lab[f'{fn}__doa_label'])
    with H.File(os.path.join(H5, 'adpit.h5')) as hf:                 # 42 recordings in the root group (a symbol node holds at most 8 by default)
        names = hf.keys()
        assert len(names) == 42 and names == sorted(names) and len(hf) == 42 and list(iter(hf)) == names
        for j in (0, 17, 39):
            assert np.array_equal(hf[f'fold3_room{j:02d}_mix/adpit/se'][...], lab['mix0__adpit__se'][j:j + 3])


def test_feature_file_second_axis_slices_and_the_latest_encoding():
    H = _lib()
    exp = np.load(os.path.join(H5, 'expected.npz'))
    with H.File(os.path.join(H5, 'feature.h5')) as hf:
        f = hf['feature']
        assert f.shape == (7, 25, 64) and f.dtype == np.float32
        assert np.array_equal(f[:, 3:9], exp['feature'][:, 3:9])        # data.py:83: hf['feature'][:, index_begin: index_end]
        assert np.array_equal(f[2:5, 1], exp['feature'][2:5, 1]) and np.array_equal(f[...], exp['feature'])
        assert hf['empty'].shape == (0, 4) and hf['empty'][...].shape == (0, 4) and hf['empty'].dtype == np.int16
        assert float(hf['scalar'][...][0]) == 3.5
    with H.File(os.path.join(H5, 'latest.h5')) as hf:                # version-3 superblock, version-2 object headers, link messages
        assert hf.keys() == ['mix0'] and sorted(hf['mix0/adpit'].keys()) == ['azi', 'se']
        assert np.array_equal(hf['mix0/adpit/se'][...], exp['mix0|adpit|se']) and np.array_equal(hf['mix0/adpit/azi'][4:44], exp['mix0|adpit|azi'][4:44])


def test_what_the_subset_does_not_cover_is_refused_not_misread(tmp_path):
    H = _lib()
    with H.File(os.path.join(H5, 'chunked.h5')) as hf:
        with pytest.raises(NotImplementedError, match='chunked'):
            hf['feature']
    p = tmp_path / 'not.h5'
    p.write_bytes(b'RIFF' + bytes(4096))
    with pytest.raises(H.Hdf5FormatError):
        H.File(str(p))
    with pytest.raises(NotImplementedError):
        H.File(os.path.join(H5, 'adpit.h5'), 'w')
    # a truncated file: the error is a format error, not an array of garbage
    raw = open(os.path.join(H5, 'accdoa.h5'), 'rb').read()
    q = tmp_path / 'cut.h5'
    q.write_bytes(raw[:len(raw) // 2])
    with pytest.raises((H.Hdf5FormatError, KeyError)):
        with H.File(str(q)) as hf:
            for fn in hf.keys():
                for k in ('se', 'azi', 'ele'):
                    hf[f'{fn}/accdoa/{k}'][...]


def test_damaged_files_raise_instead_of_hanging(tmp_path):
    """Label files are external input to a training run: a truncated file, a block address / length outside the file and a group
    B-tree whose child pointer leads back into the tree must raise Hdf5FormatError - not recurse for ever, not allocate the bogus length."""
    H = _lib()
    raw = bytearray(open(os.path.join(H5, 'adpit.h5'), 'rb').read())
    # (1) truncated behind the superblock: every later block lies outside the file
    p = tmp_path / 'cut.h5'
    p.write_bytes(bytes(raw[:600]))
    with pytest.raises((H.Hdf5FormatError, NotImplementedError)):
        with H.File(str(p)) as hf:
            list(hf.keys()); hf['mix0/adpit/se'][:]
    # (2) a B-tree node that names itself as its first child
    i = raw.find(b'TREE')
    assert i > 0
    bad = bytearray(raw)
    O = 8
    bad[i + 8 + 2 * O + 8: i + 8 + 2 * O + 8 + O] = i.to_bytes(O, 'little')       # child 0 (behind key 0) <- the node's own address
    p2 = tmp_path / 'loop.h5'
    p2.write_bytes(bytes(bad))
    with pytest.raises(H.Hdf5FormatError):
        with H.File(str(p2)) as hf:
            list(hf.keys()); [list(hf[k].keys()) for k in hf.keys()]
    # (3) the reader refuses lengths that leave the file
    with H.File(os.path.join(H5, 'adpit.h5')) as hf:
        with pytest.raises(H.Hdf5FormatError):
            hf._rd.at(10, 1 << 40)
        with pytest.raises(H.Hdf5FormatError):
            hf._rd.at(len(raw) - 4, 16)
