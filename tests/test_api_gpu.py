"""The reference's API surface on the GPU: unet_hip as net_builder_func, Model duck-type, UNet2DSummary.fit /
predict (incl. batched 8x TTA), checkpoint round trip at a new window size."""
import os

import numpy as np
import pytest

from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _make_datasets(tmp_path, n=2, hw=(72, 80)):
    rs = np.random.RandomState(3)
    paths = []
    for k in range(n):
        mean = (rs.random_sample(hw) * 500 + 50).astype(np.float16)
        masks = np.zeros((10,) + hw, np.int8)
        for z in range(10):
            cy, cx = rs.randint(6, hw[0] - 6), rs.randint(6, hw[1] - 6)
            masks[z, cy - 3:cy + 4, cx - 3:cx + 4] = 1
            mean[cy - 3:cy + 4, cx - 3:cx + 4] += 400
        p = str(tmp_path / ('ds%d.npz' % k))
        np.savez(p, series_mean=mean, masks_raw=masks, name=np.array('neurofinder.0%d.00' % k))
        paths.append(p)
    return paths


def test_model_duck_type_and_checkpoint_roundtrip(tmp_path):
    from deep_calcium_amd import unet_hip, Adam, load_model_with_new_input_shape
    m = unet_hip((32, 32), nb_filters_base=8)
    assert m.input_shape == (None, 32, 32)
    Wt = on.init_weights(8, randomize_bn=True)
    m.set_weights(Wt)
    x, y = on.synthetic_batch(3, 32, 32)
    p_ref = on.UNetOracle(Wt, 8).forward(x)
    p = m.predict(x)
    assert p.dtype == np.float32 and p.shape == (3, 32, 32) and np.abs(p - p_ref).max() < 1e-4
    m.compile(optimizer=Adam(0.002), loss='binary_crossentropy', metrics=[])
    vals = m.train_on_batch(x, y)
    assert len(vals) == 8 and m.metrics_names == ['loss', 'F1', 'prec', 'reca', 'dice', 'dicesq', 'posyt', 'posyp']
    assert np.isfinite(vals).all() and abs(vals[6] - y.mean()) < 1e-6          # posyt
    path = str(tmp_path / 'ck_01_0.500.hdf5')
    m.save(path)
    # same weights at a NEW window size (keras_helpers.load_model_with_new_input_shape seam)
    m2 = load_model_with_new_input_shape(path, (64, 64), compile=True)
    assert m2.input_shape == (None, 64, 64) and m2.engine.iterations == 1 and m2.optimizer.lr == 0.002
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert np.array_equal(a, b)
    x64, _ = on.synthetic_batch(1, 64, 64)
    assert np.abs(m2.predict(x64) - on.UNetOracle(m.get_weights(), 8).forward(x64)).max() < 1e-4
    with pytest.raises(ValueError):
        m.compile(Adam(), loss='hinge')
    # the reference's alternate losses train too (unet_2d_summary.py:372-377)
    for name in ('weighted_binary_crossentropy', 'dice_loss', 'dicesq_loss'):
        m.compile(Adam(0.002), loss=name)
        v0 = m.train_on_batch(x, y)
        for _ in range(4):
            v1 = m.train_on_batch(x, y)
        assert np.isfinite(v1).all() and v1[0] < v0[0], (name, v0[0], v1[0])


def test_fit_and_predict_end_to_end(tmp_path):
    from deep_calcium_amd import UNet2DSummary, unet_hip, INVERTIBLE_2D_AUGMENTATIONS
    from deep_calcium_amd import load_model_with_new_input_shape
    paths = _make_datasets(tmp_path)
    cp = str(tmp_path / 'cp')
    np.random.seed(865)
    model = UNet2DSummary(cpdir=cp, net_builder_func=lambda shape: unet_hip(shape, nb_filters_base=8))
    hist, _ = model.fit(paths, shape_trn=(32, 32), shape_val=(96, 96), batch_size_trn=4, nb_steps_trn=6, nb_epochs=2)
    for k in ('loss', 'F1', 'prec', 'reca', 'dice', 'dicesq', 'posyt', 'posyp', 'val_nf_f1_mean', 'val_nf_f1_median',
              'val_nf_f1_min', 'val_nf_f1_adj', 'val_nf_prec', 'val_nf_reca', 'lr'):
        assert k in hist and len(hist[k]) == 2, k
    assert hist['loss'][1] < hist['loss'][0] + 0.2 and hist['lr'] == [0.002, 0.002]
    cks = sorted(f for f in os.listdir(cp) if f.endswith('.hdf5'))
    assert len(cks) == 2 and '_model_00_' in cks[0] and '_model_01_' in cks[1]
    assert any(f.endswith('_metrics.csv') for f in os.listdir(cp))

    ck = os.path.join(cp, cks[-1])
    Mp, names = model.predict(paths, ck, window_shape=(512, 512), augmentation=False)
    assert names == ['neurofinder.00.00', 'neurofinder.01.00']
    assert all(m.dtype == np.uint8 and m.shape == (72, 80) and set(np.unique(m)) <= {0, 1} for m in Mp)
    MpT, _ = model.predict(paths, ck, window_shape=(512, 512), augmentation=True, print_scores=True)
    # batched 8x TTA == the reference's 8 sequential batch-1 forwards, inverse-mapped and averaged
    net = load_model_with_new_input_shape(ck, (512, 512), compile=False)
    s = model.series_summary_func(paths[0])
    sb = np.pad(s, ((0, 512 - s.shape[0]), (0, 512 - s.shape[1])), mode='reflect')[np.newaxis]
    mp = np.zeros(s.shape)
    for _, aug, inv in INVERTIBLE_2D_AUGMENTATIONS:
        mp += inv(net.predict(np.ascontiguousarray(aug(sb))))[0, :s.shape[0], :s.shape[1]] / 8
    assert np.array_equal((mp > 0.5).astype(np.uint8), MpT[0])
    with pytest.raises(AssertionError):
        model.predict(paths, ck, window_shape=(256, 256))


def test_keras_model_file_written_by_libhdf5_loads_and_predicts(golden_dir):
    """A Keras-2.0.6-layout model file written by real h5py (tests/golden/make_keras_fixture.py) handed straight to
    load_model_with_new_input_shape -- what predict(model_path='unet2ds_model.hdf5') does with the released weights
    (unet_2d_summary.py:560-561): weights, window size, Adam moments and iteration count arrive in the engine."""
    from deep_calcium_amd import load_model_with_new_input_shape
    path = os.path.join(golden_dir, 'keras_unet_nfb4.hdf5')
    z = np.load(os.path.join(golden_dir, 'keras_unet_nfb4.npz'))
    W = [z['w_%03d' % i] for i in range(134)]
    m = load_model_with_new_input_shape(path, (64, 64), compile=True)
    assert m.input_shape == (None, 64, 64) and m.engine.nfb == 4 and m.engine.iterations == 2000
    assert abs(m.optimizer.lr - 0.001) < 1e-9 and m.loss == 'binary_crossentropy'
    assert all(np.array_equal(a, b) for a, b in zip(m.get_weights(), W))
    x, _ = on.synthetic_batch(2, 64, 64)
    assert np.abs(m.predict(x) - on.UNetOracle(W, 4).forward(x)).max() < 1e-4
    eng = m.engine
    mflat, vflat = eng.mflat.cpu().numpy(), eng.vflat.cpu().numpy()
    k = 0
    for l in eng.layers:
        for key in ('k', 'b', 'gamma', 'beta'):
            if key in l.off:
                o, shp = l.off[key]
                n = int(np.prod(shp))
                assert np.array_equal(mflat[o:o + n].reshape(shp), z['opt_m_%03d' % k])
                assert np.array_equal(vflat[o:o + n].reshape(shp), z['opt_v_%03d' % k])
                k += 1
    assert k == 90
    # compile=False (predict / the validation model): weights only
    m2 = load_model_with_new_input_shape(path, (32, 32), compile=False)
    assert m2.optimizer is None and m2.engine.iterations == 0


def test_hdf5_checkpoint_resumes_training_bit_for_bit(tmp_path):
    """fit(model_path=..., proceed=True) semantics: a '*.hdf5' checkpoint (Keras layout, written in-process) restores
    weights + Adam state so that the next optimizer step equals the uninterrupted run's, bit for bit."""
    from deep_calcium_amd import unet_hip, Adam, load_model_with_new_input_shape
    x, y = on.synthetic_batch(2, 32, 32)
    masks = on.make_drop_masks(8, 2, 32, 32)
    a = unet_hip((32, 32), nb_filters_base=8)
    a.compile(Adam(0.002), 'binary_crossentropy')
    for _ in range(3):
        a.train_on_batch(x, y, drop_masks=masks)
    path = str(tmp_path / '7_model_02_0.123.hdf5')
    a.save(path)
    from deep_calcium_amd import hdf5_min
    assert hdf5_min.is_hdf5(path)
    b = load_model_with_new_input_shape(path, (32, 32), compile=True)
    va = a.train_on_batch(x, y, drop_masks=masks)
    vb = b.train_on_batch(x, y, drop_masks=masks)
    assert va == vb
    assert all(np.array_equal(p, q) for p, q in zip(a.get_weights(), b.get_weights()))
    # the build's own container still works and is sniffed by content, whatever the file is called
    npz = str(tmp_path / 'own_format.ckpt')
    a.save(npz)
    c = load_model_with_new_input_shape(npz, (32, 32), compile=True)
    assert c.engine.iterations == 4 and all(np.array_equal(p, q) for p, q in zip(a.get_weights(), c.get_weights()))


def test_example_script_evaluate_and_predict(tmp_path):
    """examples/neurons/unet2ds_nf.py (the reference's example, same CLI): `evaluate` and `predict` on dataset files in
    the reference's HDF5 schema with a Keras-layout model file; `predict` leaves the Neurofinder submission files."""
    import json
    import subprocess
    import sys
    from deep_calcium_amd import hdf5_min, unet_hip
    rs = np.random.RandomState(3)
    paths = []
    for k in range(2):
        hw = (72, 80)
        mean = (rs.random_sample(hw) * 500 + 50).astype(np.float16)
        masks = np.zeros((6,) + hw, np.int8)
        for zz in range(6):
            cy, cx = rs.randint(6, hw[0] - 6), rs.randint(6, hw[1] - 6)
            masks[zz, cy - 3:cy + 4, cx - 3:cx + 4] = 1
        w = hdf5_min.Writer()
        w.attrs['name'] = 'neurofinder.0%d.00' % k
        w.create_dataset('series/mean', mean)
        w.create_dataset('masks/raw', masks)
        p = str(tmp_path / ('ds%d.hdf5' % k))
        w.save(p)
        paths.append(p)
    model_path = str(tmp_path / 'model.hdf5')
    unet_hip((512, 512), nb_filters_base=4).save(model_path)
    cp = str(tmp_path / 'cp')
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'neurons', 'unet2ds_nf.py')
    for action in ('evaluate', 'predict'):
        r = subprocess.run([sys.executable, script, action, ','.join(paths), '-m', model_path, '-c', cp],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        if action == 'evaluate':
            assert 'Evaluation with TTA.' in r.stderr and 'Evaluation without TTA.' in r.stderr
            assert r.stderr.count('Mean prec=') == 2 and 'neurofinder.00.00: prec=' in r.stderr
    subs = sorted(f for f in os.listdir(cp) if f.startswith('submission_'))
    assert 'submission_latest.json' in subs and 'submission_latest_TTA.json' in subs and len(subs) == 4
    sub = json.load(open(os.path.join(cp, 'submission_latest.json')))
    # (an untrained model predicts next to nothing: an empty mask gives the dummy region, a single component gives [])
    assert [s['dataset'] for s in sub] == ['00.00', '01.00'] and all(isinstance(s['regions'], list) for s in sub)


def test_example_script_trains_by_name_from_a_neurofinder_directory(tmp_path):
    """`train neurofinder.00.00` as the reference's README runs it: the unpacked challenge directory (TIFF frames +
    regions.json) under ~/.deep-calcium/datasets/neurons_nf becomes dataset.hdf5 (nf_load_hdf5), fit() runs the example's
    10 epochs x 100 steps of 128^2 x 20 with validation, checkpoints and history land in the checkpoint directory."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _nf_dirs import make_neurofinder_dir
    home = str(tmp_path / 'home')
    name = 'neurofinder.00.00'
    root = make_neurofinder_dir('%s/.deep-calcium/datasets/neurons_nf' % home, name)
    cp = str(tmp_path / 'cp')
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'neurons', 'unet2ds_nf.py')
    env = dict(os.environ, HOME=home)
    r = subprocess.run([sys.executable, script, 'train', name, '-c', cp], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert os.path.exists(root + '/dataset.hdf5')
    files = os.listdir(cp)
    assert any(f.endswith('.hdf5') and '_model_' in f for f in files), files
    hist = [f for f in files if f.endswith('.csv')]
    assert hist, files
    rows = open(os.path.join(cp, hist[0])).read().strip().splitlines()
    assert len(rows) == 11 and 'val_nf_f1_mean' in rows[0] and 'loss' in rows[0]
    cols = rows[0].split(',')
    first, last = [dict(zip(cols, r_.split(','))) for r_ in (rows[1], rows[-1])]
    assert float(last['loss']) < float(first['loss'])          # bright 7-pixel discs on noise: it learns


@pytest.mark.parametrize('aug', [True, False])
def test_pipelined_predict_equals_one_by_one(aug, monkeypatch):
    """The enqueue-all-then-synchronise path of predict() (engine.tta_begin) gives, for datasets of different sizes, the
    masks the per-dataset path gives -- with the 8x table and with the plain forward (identity map); a dataset whose
    activations leave fp16's range (un-normalised image) is flagged, redone with measured bounds, and the rest are kept."""
    from deep_calcium_amd.net import UNetEngine
    from deep_calcium_amd.unet2ds import INVERTIBLE_2D_AUGMENTATIONS
    from oracle import unet_numpy as on
    monkeypatch.delenv('DC_INFER_GUARD', raising=False)        # the test is about the (default) optimistic mode
    H = W = 64
    nfb = 8
    Wt = on.init_weights(nfb, seed=9, randomize_bn=True)
    rs = np.random.RandomState(4)
    sizes = [(64, 64), (50, 61), (33, 64), (64, 40), (17, 23)]
    imgs = []
    for k, (hs, ws) in enumerate(sizes):
        s = rs.standard_normal((hs, ws)).astype(np.float32)
        if k == 2:
            s *= 3e5                                           # this one overflows the optimistic pass
        imgs.append(np.pad(s, ((0, H - hs), (0, W - ws)), mode='reflect'))
    table = INVERTIBLE_2D_AUGMENTATIONS if aug else None
    eng = UNetEngine((H, W), nb_filters_base=nfb)
    eng.set_weights(Wt)
    thr = 0.5
    job = eng.tta_begin(len(sizes), table)
    for i, ((hs, ws), im) in enumerate(zip(sizes, imgs)):
        job.enqueue(i, im, hs, ws, thr)
    assert not eng.infer_measured
    got = job.finish()
    assert eng.infer_measured                                  # dataset 2 raised the flag and was redone
    ref_eng = UNetEngine((H, W), nb_filters_base=nfb)
    ref_eng.set_weights(Wt)
    one = [('identity', lambda a: a, lambda a: a)]
    for i, ((hs, ws), im) in enumerate(zip(sizes, imgs)):
        e = UNetEngine((H, W), nb_filters_base=nfb)            # fresh engine: optimistic unless it has to switch
        e.set_weights(Wt)
        want = e.predict_tta(im, table if aug else one, hs, ws, thr)
        assert got[i].shape == (hs, ws) and got[i].dtype == np.uint8
        assert np.array_equal(got[i], want), i
        if not aug and i != 2:
            p = ref_eng.forward_infer_checked(torch.from_numpy(im[None]).cuda()).cpu().numpy()[0, :hs, :ws]
            assert np.array_equal(got[i], (p > thr).astype(np.uint8))
    assert eng.tta_begin(0, table).finish() == []


def test_predict_reuses_the_loaded_model_until_the_file_changes(tmp_path):
    """predict() (unet_2d_summary.py:560-561 re-reads the model file on every call) keeps the loaded model per (path, size,
    modification time, window): repeated calls give the same masks from the SAME engine with its forward replayed from a tape, a
    rewritten file is loaded afresh, and taped forwards equal launch-by-launch ones bit for bit."""
    import time
    from deep_calcium_amd import UNet2DSummary
    from deep_calcium_amd.model import Model
    paths = _make_datasets(tmp_path, n=3)
    api = UNet2DSummary(cpdir=str(tmp_path))
    m = Model((512, 512), 8)
    mp = str(tmp_path / 'm.hdf5')
    m.save(mp, include_optimizer=False)
    first, _ = api.predict(paths, mp, augmentation=True)
    eng = api._predict_models[-1][1].engine
    for _ in range(3):
        again, _ = api.predict(paths, mp, augmentation=True)
        assert api._predict_models[-1][1].engine is eng and len(api._predict_models) == 1
        assert all(np.array_equal(a, b) for a, b in zip(first, again))
    assert eng.tape_replays > 0                                    # 12 batch-8 forwards: two recordings, the rest replayed
    eng.use_tapes = False
    untaped, _ = api.predict(paths, mp, augmentation=True)
    assert all(np.array_equal(a, b) for a, b in zip(first, untaped))
    # another set of weights under the same path: picked up (size / mtime differ), and the masks change
    W = m.get_weights()
    m.set_weights([w + np.float32(0.05) * np.sign(w) if w.ndim == 4 else w for w in W])
    time.sleep(0.02)
    m.save(mp, include_optimizer=False)
    os.utime(mp, ns=(time.time_ns(), time.time_ns()))
    changed, _ = api.predict(paths, mp, augmentation=True)
    assert api._predict_models[-1][1].engine is not eng
    fresh, _ = UNet2DSummary(cpdir=str(tmp_path)).predict(paths, mp, augmentation=True)
    assert all(np.array_equal(a, b) for a, b in zip(changed, fresh))
    assert any(not np.array_equal(a, b) for a, b in zip(changed, first))
