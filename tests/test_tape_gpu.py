"""Launch tapes (include/dcunet.h dc_tape_*, csrc/tape.cpp, UNetEngine._taped): a train step replayed from C must be THE step --
the same launches with the same arguments on the same streams -- so a taped run and a launch-by-launch run of the same seeded
training are bit-identical, whatever changes between steps (dropout seeds, Adam's lr_t, the learning rate itself, the batch
buffers, a set_weights() in between).  The reference's loop is model.fit_generator -> train_on_batch
(/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:429-430)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _pair(hw, nfb, **kw):
    from deep_calcium_amd.model import Model, Adam
    a, b = Model((hw, hw), nfb, **kw), Model((hw, hw), nfb, **kw)
    for m in (a, b):
        m.compile(Adam(0.002), 'binary_crossentropy')
    b.engine.use_tapes = False
    assert a.engine.use_tapes
    return a, b


def _batches(n, B, hw, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        x = torch.from_numpy(rs.standard_normal((B, hw, hw)).astype(np.float32)).cuda()
        y = torch.from_numpy((rs.random_sample((B, hw, hw)) < 0.15).astype(np.uint8)).cuda()
        out.append((x, y))
    return out


@pytest.mark.parametrize('hw,nfb,B', [(32, 8, 4), (64, 32, 3), (128, 32, 20)])
def test_taped_training_is_bit_identical_to_launch_by_launch(hw, nfb, B):
    a, b = _pair(hw, nfb)
    data = _batches(3, B, hw, 5)                 # three (x, y) buffer pairs in rotation: the batch pointers are tape values
    hist = [[], []]
    for step in range(9):
        if step == 6:                            # ReduceLROnPlateau halves the rate
            a.optimizer.lr = b.optimizer.lr = 0.001
        x, y = data[step % 3]
        for k, m in enumerate((a, b)):
            hist[k].append(m.train_on_device_batch(x, y))
    torch.cuda.synchronize()
    assert a.engine.tape_replays >= 3 * 6 and b.engine.tape_replays == 0      # steps 2.. of (forward, backward, adam) were replayed
    assert hist[0] == hist[1]                    # loss + 7 metrics of every step (RNG dropout active: the seeds were patched)
    assert torch.equal(a.engine.pflat, b.engine.pflat) and torch.equal(a.engine.sflat, b.engine.sflat)
    assert torch.equal(a.engine.mflat, b.engine.mflat) and torch.equal(a.engine.vflat, b.engine.vflat)
    # a set_weights() between two steps dirties the packed weight images: the next forward must re-pack (another tape key)
    W = a.get_weights()
    for m in (a, b):
        m.set_weights([w * np.float32(0.5) if w.ndim == 4 else w for w in W])
        for step in range(3):
            m.train_on_device_batch(*data[step])
    torch.cuda.synchronize()
    assert torch.equal(a.engine.pflat, b.engine.pflat)
    # the un-deferred backward (what grads() / the data-parallel path use) next to the deferred one
    for m in (a, b):
        for _ in range(3):
            m.engine.forward_train(*data[0])
            m.engine.backward()
    ga, gb = a.engine.grads(), b.engine.grads()
    for name in ga:
        for u, v in zip(ga[name], gb[name]):
            assert np.array_equal(u, v), name


@pytest.mark.parametrize('hw,nfb,B', [(32, 8, 6), (128, 32, 20)])
def test_last_weight_gradient_on_the_main_stream_is_the_same_step(hw, nfb, B):
    """DC_TAIL_MAIN (the first layer's weight gradient on the main stream beside the previous block's on the side stream, own slab
    workspace, one Adam launch) against the side-stream tail with the deferred Adam: 12 seeded steps with RNG dropout, parameters and
    Adam state bit for bit after every step, with the weight-gradient stream of one run held back by unrelated work.  (The block on the
    main stream still has to wait for the weight gradient that last read its dz slot: that race, when it existed, needed two ranks
    contending for one GPU to show -- tests/test_dp_gpu.py's bitwise bucket comparison caught it, this single-process run did not.)"""
    a, b = _pair(hw, nfb)
    b.engine.use_tapes = True
    a.engine.tail_main, b.engine.tail_main = True, False
    data = _batches(2, B, hw, 11)
    junk = torch.randn(2048, 2048, device='cuda') * 0.01
    for step in range(12):
        x, y = data[step % 2]
        if step >= 3:
            # the weight-gradient stream of `a` is held back by a few ms of unrelated work: the main stream runs through the whole
            # backward ahead of it, every buffer hand-back (slot_free / g_free events) is now load-bearing
            with torch.cuda.stream(a.engine._side_stream):
                t = junk
                for _ in range(6):
                    t = (t @ junk) * 0.01
        va, vb = a.train_on_device_batch(x, y), b.train_on_device_batch(x, y)
        assert va == vb, step
        assert torch.equal(a.engine.gflat, b.engine.gflat), step
    torch.cuda.synchronize()
    assert torch.equal(a.engine.pflat, b.engine.pflat) and torch.equal(a.engine.mflat, b.engine.mflat) and torch.equal(a.engine.vflat, b.engine.vflat)


def test_tape_failure_names_the_operation(dclib):
    """A replayed operation goes through its entry point's own checks: the failing one is named, its message kept."""
    from deep_calcium_amd._lib import Tape, DcunetError
    t = Tape(dclib, [('dc_fill', (torch.zeros(8, device='cuda').data_ptr(), 8, 1.0, None)),
                     ('dc_conv3x3_dgrad', (None, None, None, 1, 8, 8, 8, 8, None))])
    with pytest.raises(DcunetError, match=r'operation 1 \(dc_conv3x3_dgrad\) failed: dc_conv3x3_dgrad: null pointer'):
        t.replay({})
    with pytest.raises(DcunetError, match='not a tape-able entry point'):
        Tape(dclib, [('dc_crop_augment', (0,) * 9)])


def test_fit_through_tapes_and_device_batches_reaches_the_same_history(tmp_path, monkeypatch):
    """UNet2DSummary.fit(): tapes on (default) and DC_TAPES=0 give the same history (RNG dropout on: prop_dropout_base 0.25)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_api_gpu import _make_datasets
    from deep_calcium_amd import UNet2DSummary, unet_hip
    paths = _make_datasets(tmp_path, n=2)

    def run(tag):
        np.random.seed(865)
        u = UNet2DSummary(cpdir=str(tmp_path / tag), net_builder_func=lambda ws: unet_hip(ws, nb_filters_base=8))
        hist, _ = u.fit(paths, shape_trn=(32, 32), shape_val=(96, 96), batch_size_trn=6, nb_steps_trn=8, nb_epochs=2)
        return hist, u.model.engine.tape_replays

    h1, r1 = run('taped')
    monkeypatch.setenv('DC_TAPES', '0')
    h0, r0 = run('plain')
    assert r1 >= 3 * 12 and r0 == 0
    assert h1 == h0


def test_engine_state_change_drops_the_tapes_and_the_run_stays_the_untaped_one():
    """A tape bakes in the control flow a body took and raw device pointers.  Engine state the bodies branch on that is not in
    the tape keys (per-workgroup BatchNorm partial rows, the tail placement, the fold threshold, ...) and any buffer regrow
    must therefore drop the tapes (UNetEngine._TAPE_STATE / invalidate_tapes): a switch flipped MID-RUN gives the run a
    launch-by-launch engine with the same flips gives, bit for bit, and the tapes are re-recorded afterwards."""
    a, b = _pair(64, 32)
    data = _batches(2, 3, 64, 21)
    flips = {4: ('stats_per_wg', False), 8: ('tail_main', False), 12: ('stats_per_wg', True)}
    for step in range(16):
        if step in flips:
            name, val = flips[step]
            before = a.engine._tape_epoch if hasattr(a.engine, '_tape_epoch') else 0
            for m in (a, b):
                setattr(m.engine, name, val)
            assert a.engine._tape_epoch == before + 1 and not a.engine._tapes       # dropped, not replayed into the old sequence
        x, y = data[step % 2]
        for m in (a, b):
            m.train_on_device_batch(x, y)
    torch.cuda.synchronize()
    assert a.engine.tape_replays >= 3 * 4 and b.engine.tape_replays == 0
    assert torch.equal(a.engine.pflat, b.engine.pflat) and torch.equal(a.engine.sflat, b.engine.sflat)
    assert torch.equal(a.engine.mflat, b.engine.mflat) and torch.equal(a.engine.vflat, b.engine.vflat)
    # assigning the value a switch already has keeps the tapes; a regrown scratch buffer drops them
    n = len(a.engine._tapes)
    assert n > 0
    a.engine.stats_per_wg = True
    assert len(a.engine._tapes) == n
    a.engine.buf('tape_test_scratch', 16)
    assert len(a.engine._tapes) == n             # a NEW buffer moves nothing
    a.engine.buf('tape_test_scratch', 1 << 20)
    assert not a.engine._tapes


def test_tape_verify_rerecords_and_catches_a_stale_tape():
    """DC_TAPE_VERIFY=K (UNetEngine.tape_verify): every K-th replay is issued launch by launch under a fresh recording and compared
    with the tape -- the run stays the untaped one bit for bit, and a tape that no longer matches what the engine issues raises."""
    from deep_calcium_amd._lib import DcunetError
    a, b = _pair(32, 8)
    a.engine.tape_verify = 2
    data = _batches(2, 4, 32, 3)
    for step in range(8):
        x, y = data[step % 2]
        for m in (a, b):
            m.train_on_device_batch(x, y)
    torch.cuda.synchronize()
    assert a.engine.tape_replays >= 12
    assert torch.equal(a.engine.pflat, b.engine.pflat) and torch.equal(a.engine.mflat, b.engine.mflat)
    # a tape whose recording no longer equals the live sequence (simulated: one recorded op dropped) is reported, not replayed
    key = [k for k in a.engine._tapes if k[0] == 'fwd'][0]
    a.engine._tapes[key]['ops'] = a.engine._tapes[key]['ops'][:-1]
    a.engine.tape_verify = 1
    with pytest.raises(DcunetError, match='no longer matches'):
        a.train_on_device_batch(*data[0])
