"""fit()'s host side on the device (round 5): training batches cut out of resident summaries by dc_crop_augment, the validation
callback's forwards / rounding on the device with native scoring, checkpoints written from a snapshot in the background.

Reference: /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:434-530 (_batch_gen), :62-120 (_ValidationMetricsCB),
:423-424 (ModelCheckpoint).  The generator's golden (tests/golden/batch_gen.npz) was produced by the reference's own
_batch_gen (tests/golden/make_goldens.py): the device batch must equal it bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _engine_for(S, M, hw):
    from deep_calcium_amd.net import UNetEngine
    eng = UNetEngine((hw, hw), 4, device='cuda:0')
    eng.set_crop_sources(S, M)
    return eng


def test_device_batches_equal_the_reference_generated_golden(golden_dir):
    """Seeded like the golden: the items of _device_batch_gen, cut by dc_crop_augment, ARE the reference's batches."""
    from deep_calcium_amd import UNet2DSummary
    g = _load(golden_dir, 'batch_gen.npz')
    S, M = [g['S0'], g['S1']], [g['M0'], g['M1']]
    m = UNet2DSummary.__new__(UNet2DSummary)
    eng = _engine_for(S, M, 32)
    for nb_aug, key in ((15, 'aug15'), (0, 'aug0')):
        np.random.seed(865)
        gen = m._device_batch_gen(S, M, ['a', 'b'], [(0, 48), (0, 48)], 4, 10, (32, 32), nb_aug)
        for i in range(3):
            b = next(gen)
            assert len(b) == 4 and b.items.shape == (4, 4) and b.items.dtype == np.int64
            x, y = eng.crop_batch(b.items)
            torch.cuda.synchronize()
            assert np.array_equal(x.cpu().numpy(), g['%s_s%d' % (key, i)]) and x.dtype == torch.float32
            assert np.array_equal(y.cpu().numpy(), g['%s_m%d' % (key, i)]) and y.dtype == torch.uint8
    # the ragged golden: a training stripe shorter than the window (zero-filled short crops, :520-521)
    np.random.seed(865)
    b = next(m._device_batch_gen(S, M, ['a', 'b'], [(0, 24), (0, 24)], 2, 10, (32, 32), 3))
    x, y = eng.crop_batch(b.items)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy(), g['ragged_s0']) and np.array_equal(y.cpu().numpy(), g['ragged_m0'])


@pytest.mark.parametrize('hw,B,world', [(32, 6, 1), (48, 8, 2), (64, 66, 1), (16, 4, 1)])
def test_device_batches_equal_the_host_generator(hw, B, world):
    """Random datasets of different shapes, windows larger than the training stripe (zero-filled short crops), every
    dihedral composition, more than DC_CROP_MAX_ITEMS items, and the data-parallel slices: device batch == host batch."""
    from deep_calcium_amd import UNet2DSummary
    rs = np.random.RandomState(hw + B)
    shapes = [(70, 90), (55, 64), (96, 60)]
    S = [rs.standard_normal(sh).astype(np.float32) for sh in shapes]
    M = [(rs.random_sample(sh) < 0.05).astype(np.float64) for sh in shapes]
    yc = [(0, int(sh[0] * 0.75)) for sh in shapes]
    m = UNet2DSummary.__new__(UNet2DSummary)
    eng = _engine_for(S, M, hw)
    for r in range(world):
        np.random.seed(7)                # (both draw from numpy's GLOBAL stream: one generator at a time)
        host = m._batch_gen(S, M, ['a', 'b', 'c'], yc, B, 10, (hw, hw), 15, shard=(r, world))
        want = [next(host) for _ in range(4)]
        np.random.seed(7)
        devg = m._device_batch_gen(S, M, ['a', 'b', 'c'], yc, B, 10, (hw, hw), 15, shard=(r, world))
        seen = set()
        for i in range(4):
            xs, ys = want[i]
            b = next(devg)
            seen.update(int(v) for v in b.items[:, 3])
            x, y = eng.crop_batch(b.items)
            torch.cuda.synchronize()
            assert x.shape == (B // world, hw, hw)
            assert np.array_equal(x.cpu().numpy(), xs) and np.array_equal(y.cpu().numpy(), ys)
        if B >= 8:
            assert len(seen) >= 6, seen


def test_crop_augment_rejects_items_outside_the_source():
    from deep_calcium_amd._lib import DcunetError
    S = [np.zeros((40, 40), np.float32)]
    eng = _engine_for(S, [np.zeros((40, 40))], 32)
    ok = np.array([[0, 40, (32 << 32) | 32, 5]], np.int64)
    eng.crop_batch(ok)
    for bad in ([[40 * 40, 40, (32 << 32) | 32, 0]], [[0, 40, (33 << 32) | 32, 0]], [[0, 40, (32 << 32) | 32, 8]], [[-1, 40, 1 << 32 | 1, 0]],
                [[0, 16, (32 << 32) | 32, 0]]):
        with pytest.raises(DcunetError, match='dc_crop_augment'):
            eng.crop_batch(np.array(bad, np.int64))
    torch.cuda.synchronize()


def test_round_window_is_numpy_round_of_the_stripe(dclib):
    rs = np.random.RandomState(2)
    p = rs.random_sample((3, 40, 56)).astype(np.float32)
    p[0, 5, 7] = 0.5                      # half to even: 0
    p[1, 9, 9] = np.float32(0.5) + np.float32(6e-8)
    pd = torch.from_numpy(p).cuda()
    y0, y1, x0, x1 = 4, 33, 3, 50
    out = torch.full((3, y1 - y0, x1 - x0), 7, dtype=torch.uint8, device='cuda')
    dclib.dc_round_window_u8(pd.data_ptr(), 3, 40, 56, y0, y1, x0, x1, out.data_ptr(), None)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), p[:, y0:y1, x0:x1].round().astype(np.uint8))


def _fit_setup(tmp_path, n=3, hw=(72, 80)):
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_api_gpu import _make_datasets
    return _make_datasets(tmp_path, n=n, hw=hw)


def test_validation_on_device_equals_the_references_sequence(tmp_path):
    """_ValidationMetricsCB.on_epoch_end through the device path (resident padded images, batch-8 forwards, dc_round_window_u8,
    native scorer on a thread) writes the logs the reference's own sequence -- predict one by one, .round(), nf_mask_metrics
    (:76-112) -- writes: bit for bit, incl. epoch eps and the per-dataset pickle."""
    import pickle
    from deep_calcium_amd import UNet2DSummary, unet_hip, Adam
    from deep_calcium_amd.unet2ds import _ValidationMetricsCB, _summarize_series, _summarize_mask
    paths = _fit_setup(tmp_path)
    S = [_summarize_series(p) for p in paths]
    M = [_summarize_mask(p) for p in paths]
    yc = [(s.shape[0] - int(s.shape[0] * 0.25), s.shape[0]) for s in S]
    model = unet_hip((32, 32), nb_filters_base=8)
    model.compile(Adam(0.002))
    rs = np.random.RandomState(0)
    # a few steps so that BatchNorm's moving statistics and the head are non-trivial and the masks non-empty
    for _ in range(12):
        k = rs.randint(len(S))
        y0, x0 = rs.randint(0, S[k].shape[0] - 32), rs.randint(0, S[k].shape[1] - 32)
        xb = np.stack([S[k][y0:y0 + 32, x0:x0 + 32]] * 4).astype(np.float32)
        yb = np.stack([M[k][y0:y0 + 32, x0:x0 + 32]] * 4).astype(np.uint8)
        model.train_on_batch(xb, yb)
    model_val = unet_hip((96, 96), nb_filters_base=8)
    sp = str(tmp_path / 'scores.pkl')
    cb = _ValidationMetricsCB(model_val, S, M, ['a', 'b', 'c'], yc, scores_path=sp)
    cb.set_model(model)
    logs_dev, logs_ref = {}, {}
    cb.on_epoch_end(3, logs_dev)
    f1_dev = pickle.load(open(sp, 'rb'))
    again = {}
    cb.on_epoch_end(3, again)                       # resident buffers reused
    assert again == logs_dev
    # the reference's sequence on the same weights
    model_val.set_weights(model.get_weights())
    scores = cb._score_through_predict(len(cb.S_summ))
    ff, pp, rr = scores[:, 2], scores[:, 0], scores[:, 1]
    eps = 1e-4 * 3
    logs_ref = {'val_nf_f1_mean': np.mean(ff) + eps, 'val_nf_f1_median': np.median(ff) + eps, 'val_nf_f1_min': np.min(ff) + eps,
                'val_nf_f1_adj': np.mean(ff) * np.min(ff) + eps, 'val_nf_prec': np.mean(pp), 'val_nf_reca': np.mean(rr)}
    assert set(logs_dev) == set(logs_ref)
    for k in logs_ref:
        assert logs_dev[k] == logs_ref[k], (k, logs_dev[k], logs_ref[k])
    assert max(ff) > 0, 'degenerate case: nothing was detected'
    assert [f1_dev[n] for n in 'abc'] == [list(ff[6 * i:6 * i + 6]) for i in range(3)]


@pytest.mark.parametrize('nfb', [8, 32])
def test_validation_crop_is_bit_identical_to_the_full_forward(nfb):
    """The validation callback forwards only the scored stripe + a 112-pixel halo (_ValidationMetricsCB.HALO: the network's output
    depends on the input within 102 pixels).  At the real validation size (512 x 512 windows, stripes of 128 rows / columns after
    the 6 augmentations) the bytes that come back -- `mp[y0:y1, x0:x1].round()` of every item -- and the logs must equal those of
    the full 512 x 512 forwards, on weights with non-trivial BatchNorm statistics."""
    from deep_calcium_amd import unet_hip
    from deep_calcium_amd.unet2ds import _ValidationMetricsCB
    from oracle import unet_numpy as on
    rs = np.random.RandomState(4)
    S = [rs.standard_normal((512, 512)).astype(np.float32), rs.standard_normal((500, 470)).astype(np.float32)]
    M = []
    for s_ in S:
        m = np.zeros(s_.shape)
        for _ in range(80):
            cy, cx = rs.randint(6, s_.shape[0] - 6), rs.randint(6, s_.shape[1] - 6)
            m[cy - 3:cy + 4, cx - 3:cx + 4] = 1
        M.append(m)
    yc = [(s_.shape[0] - int(s_.shape[0] * 0.25), s_.shape[0]) for s_ in S]
    model = unet_hip((32, 32), nb_filters_base=nfb)
    model.set_weights(on.init_weights(nfb, seed=5, randomize_bn=True))
    model_val = unet_hip((512, 512), nb_filters_base=nfb)
    cb = _ValidationMetricsCB(model_val, S, M, ['a', 'b'], yc)
    cb.set_model(model)
    logs_crop, logs_full = {}, {}
    cb.on_epoch_end(1, logs_crop)
    shapes = sorted(cb._dev['groups'])
    assert (240, 512) in shapes and (512, 240) in shapes, shapes            # 128-row / -column stripes + halo, 16-aligned
    bytes_crop = cb._dev['out_host'].numpy().copy()
    cb.HALO = None                                                         # the reference's own extent: the whole padded image
    cb.on_epoch_end(1, logs_full)
    assert sorted(cb._dev['groups']) == [(512, 512)]
    bytes_full = cb._dev['out_host'].numpy().copy()
    assert bytes_crop.shape == bytes_full.shape and 0.02 < bytes_full.mean() < 0.98
    assert np.array_equal(bytes_crop, bytes_full)
    assert logs_crop == logs_full


def test_background_checkpoint_is_the_snapshot_not_the_later_weights(tmp_path):
    """ModelCheckpoint(background=True): the file written by the background thread holds the weights / Adam state of the moment
    save() was called, byte for byte what a blocking save() of that moment writes -- although training went on meanwhile."""
    from deep_calcium_amd import unet_hip, Adam
    from deep_calcium_amd.model import read_checkpoint
    m = unet_hip((32, 32), nb_filters_base=8)
    m.compile(Adam(0.002))
    x = np.random.RandomState(1).standard_normal((4, 32, 32)).astype(np.float32)
    y = (np.random.RandomState(2).random_sample((4, 32, 32)) < 0.2).astype(np.uint8)
    m.train_on_batch(x, y)
    a, b, c = (str(tmp_path / n) for n in ('sync.hdf5', 'bg.hdf5', 'bg.npz'))
    m.save(a)
    m.save(b, background=True)
    for _ in range(3):
        m.train_on_batch(x, y)                      # the snapshot must not see these
    m.wait_for_saves()
    assert open(a, 'rb').read() == open(b, 'rb').read()
    w_now = m.get_weights()
    m.save(c, background=True)
    m.wait_for_saves()
    st = read_checkpoint(c)
    assert all(np.array_equal(u, v) for u, v in zip(st['weights'], w_now)) and st['optimizer']['iterations'] == 4
    assert not [f for f in os.listdir(str(tmp_path)) if '.partial.' in f]
    with pytest.raises(IOError, match='failed'):
        m.save(str(tmp_path / 'no_such_dir' / 'x.hdf5'), background=True)
        m.wait_for_saves()


def test_fit_uses_device_batches_and_matches_host_batches(tmp_path, monkeypatch):
    """UNet2DSummary.fit() with the generator on the device (default) and with DC_HOST_BATCHES=1 (numpy crops, pinned staging):
    same seed -> the same batches -> the same history (explicit dropout off: nfb-8 net with prop_dropout_base 0)."""
    from deep_calcium_amd import UNet2DSummary, unet_hip
    paths = _fit_setup(tmp_path, n=2)

    def run(tag):
        np.random.seed(865)
        u = UNet2DSummary(cpdir=str(tmp_path / tag), net_builder_func=lambda ws: unet_hip(ws, nb_filters_base=8, prop_dropout_base=0.0))
        hist, _ = u.fit(paths, shape_trn=(32, 32), shape_val=(96, 96), batch_size_trn=6, nb_steps_trn=5, nb_epochs=2)
        files = sorted(os.listdir(str(tmp_path / tag)))
        assert len([f for f in files if f.endswith('.hdf5')]) == 2 and not [f for f in files if '.partial.' in f], files
        return hist

    h_dev = run('dev')
    monkeypatch.setenv('DC_HOST_BATCHES', '1')
    h_host = run('host')
    assert set(h_dev) == set(h_host) and 'val_nf_f1_mean' in h_dev
    for k in h_dev:
        assert h_dev[k] == h_host[k], (k, h_dev[k], h_host[k])
