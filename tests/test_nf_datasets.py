"""nf_load_hdf5 (the dataset format on the input side of the path) against the reference's own function run with real
h5py on synthetic Neurofinder directories (tests/golden/make_nf_dataset_golden.py)."""
import json
import os

import numpy as np
import pytest

from deep_calcium_amd import hdf5_min, nf_datasets

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'nf_dataset.npz'))
CASES = ['neurofinder.00.00', 'neurofinder.01.00.test']


def _make_dirs(root):
    from PIL import Image
    for name in CASES:
        os.makedirs('%s/%s/images' % (root, name))
        for i, fr in enumerate(G['frames_' + name]):
            Image.fromarray(fr).save('%s/%s/images/image%05d.tiff' % (root, name, i))
        key = 'regions_json_' + name
        if key in G.files:
            os.makedirs('%s/%s/regions' % (root, name))
            with open('%s/%s/regions/regions.json' % (root, name), 'wb') as fp:
                fp.write(bytes(G[key]))


def test_dataset_files_equal_the_reference_writers(tmp_path):
    root = str(tmp_path)
    _make_dirs(root)
    paths = nf_datasets.nf_load_hdf5(','.join(CASES), datasets_dir=root)
    assert paths == ['%s/%s/dataset.hdf5' % (root, n) for n in CASES]
    for name, p in zip(CASES, paths):
        with hdf5_min.File(p) as f:
            got_name = f.attrs['name']
            got_name = got_name.decode() if isinstance(got_name, bytes) else str(got_name)
            assert got_name == bytes(G['attr_name_' + name]).decode()
            keys = bytes(G['keys_' + name]).decode().split(',')
            for k in keys:
                want = G['%s:%s' % (name, k)]
                got = f[k].read()
                assert got.dtype == want.dtype and got.shape == want.shape, k
                assert np.array_equal(got, want, equal_nan=True), (name, k)
            if '.test' in name:
                assert 'masks' not in [c for c in f.keys()]
    # quirks the fixture pins: saturation beyond int16, float16 accumulation (inf above 65504, not the rounded f64 mean)
    raw = G['neurofinder.00.00:series/raw']
    assert raw[0, 0, 0] == 32767 and raw[0, 0, 1] == 32767 and raw[0, 0, 3] == 32767
    mean = G['neurofinder.00.00:series/mean']
    assert np.isinf(mean[0, 0])
    fr = G['frames_neurofinder.00.00'].astype(np.float64)
    with np.errstate(over='ignore'):
        assert (mean[2:] != fr.mean(0).astype(np.float16)[2:]).any()
    # second call: files exist, nothing is rebuilt
    stamp = [os.path.getmtime(p) for p in paths]
    assert nf_datasets.nf_load_hdf5(CASES, datasets_dir=root) == paths
    assert [os.path.getmtime(p) for p in paths] == stamp


def test_dataset_file_feeds_the_summaries(tmp_path):
    """The file nf_load_hdf5 writes is what fit()/predict() read: summaries come out as from the arrays themselves."""
    from deep_calcium_amd import unet2ds
    root = str(tmp_path)
    _make_dirs(root)
    p = nf_datasets.nf_load_hdf5(CASES[0], datasets_dir=root)[0]
    with np.errstate(invalid='ignore'):
        s = unet2ds._summarize_series(p)
    m = unet2ds._summarize_mask(p)
    assert s.shape == G['frames_' + CASES[0]].shape[1:] and m.shape == s.shape
    assert np.array_equal(m, unet2ds._flatten_mask_stack(G[CASES[0] + ':masks/raw']))


def test_names_and_urls():
    assert ','.join(nf_datasets.NEUROFINDER_NAMES) == bytes(G['neurofinder_names']).decode()
    assert nf_datasets.NAME_TO_URL['neurofinder.00.00'] == bytes(G['url_00_00']).decode()
    assert nf_datasets._expand_names('all') == nf_datasets.NEUROFINDER_NAMES
    assert all('.test' in n for n in nf_datasets._expand_names('all_test')) and len(nf_datasets._expand_names('all_test')) == 9
    assert len(nf_datasets._expand_names('all_train')) == 19
    with pytest.raises(KeyError):
        nf_datasets.nf_load_hdf5('not.a.dataset', datasets_dir='/tmp/nf_none')


def test_missing_directory_offline_is_a_clear_error(tmp_path, monkeypatch):
    import sys
    import types
    fake = types.ModuleType('requests')
    def get(url):
        raise OSError('no network')
    fake.get = get
    monkeypatch.setitem(sys.modules, 'requests', fake)
    with pytest.raises(IOError, match='unpack the zip'):
        nf_datasets.nf_load_hdf5('neurofinder.02.00', datasets_dir=str(tmp_path))


def test_waiting_rank_fails_fast_when_rank0_is_gone(tmp_path, monkeypatch):
    """A rank that waits for rank 0's dataset build gives up after `stale_s` without a heartbeat (rank 0 killed, or the directory
    is not shared) instead of polling for hours; a failure marker counts only when it carries THIS launch's token (a stale
    marker of an earlier run, whatever its mtime, is ignored)."""
    import time
    from deep_calcium_amd import parallel
    monkeypatch.setattr(parallel, 'rank', lambda: 1)
    monkeypatch.setattr(parallel, 'world_size', lambda: 2)
    monkeypatch.setenv('MASTER_PORT', '12345')
    root = str(tmp_path)
    with open('%s/.nf_build_failed' % root, 'w') as fp:
        fp.write('other-run@-:999\nValueError: an earlier launch')
    t = time.time()
    with pytest.raises(IOError, match='no sign of life from rank 0'):
        nf_datasets.nf_load_hdf5('neurofinder.02.00', datasets_dir=root, stale_s=0.6)
    assert time.time() - t < 5
    with open('%s/.nf_build_failed' % root, 'w') as fp:
        fp.write('%s\nIOError: disk full' % nf_datasets._run_token())
    with pytest.raises(IOError, match='disk full'):
        nf_datasets.nf_load_hdf5('neurofinder.02.00', datasets_dir=root, stale_s=30)


def test_deferred_datasets_roundtrip(tmp_path):
    w = hdf5_min.Writer()
    a = w.create_dataset('g/big', shape=(3, 4, 5), dtype='int16')
    w.create_dataset('g/small', np.arange(6, dtype=np.float32))
    b = w.create_dataset('empty', shape=(0, 4), dtype='int8')
    p = str(tmp_path / 'd.h5')
    w.save(p)
    m = w.open_deferred(a)
    m[...] = np.arange(60).reshape(3, 4, 5)
    m.flush()
    del m
    with hdf5_min.File(p) as f:
        assert np.array_equal(f['g/big'].read(), np.arange(60, dtype=np.int16).reshape(3, 4, 5))
        assert np.array_equal(f['g/small'].read(), np.arange(6, dtype=np.float32))
        assert f['empty'].read().shape == (0, 4)
    assert b.file_offset is None
