"""Neurofinder directories -> the HDF5 dataset files `UNet2DSummary.fit()/predict()` read.

Restates `nf_load_hdf5` of /root/reference/deepcalcium/datasets/nf.py:37-150 (the data format on the INPUT side of the
hot path): per dataset `<datasets_dir>/<name>/dataset.hdf5` with
    attrs['name'], series/raw (n,h,w) int16, series/mean (h,w) float16, series/max (h,w) int16,
    masks/raw (neurons,h,w) int8, masks/max (h,w) int8           (no masks for '.test' datasets)
from `<name>/images/*.tiff` (sorted) and `<name>/regions/regions.json`.  What the reference's h5py calls do to the
values is reproduced, quirks included:
  * 16-bit frames are stored as int16 through libhdf5's SATURATING conversion (65535 -> 32767), nf.py:113,:121;
  * series/mean is accumulated IN its float16 storage, one rounding per frame (`ds_mean[...] += img * 1. / n`,
    nf.py:122), so it is not the float64 mean rounded once -- and overflows to inf above 65504;
  * series/max starts at zero (nf.py:119), so an all-negative pixel reads 0.
Pinned by tests/golden/nf_dataset.npz: the reference's own function run on synthetic directories with real h5py.
The file is written by `hdf5_min.Writer` (h5py is not a dependency); series/raw is streamed frame by frame into a
reserved extent, as the reference streams it into its dataset.
Host-side numpy; not part of the GPU path.
"""
import json
import logging
import os
import threading
import time
import shutil
from glob import glob

import numpy as np

from . import hdf5_min

NEUROFINDER_NAMES = sorted(
    ['neurofinder.00.%02d' % i for i in range(12)] +
    ['neurofinder.01.00', 'neurofinder.01.01', 'neurofinder.02.00', 'neurofinder.02.01', 'neurofinder.03.00',
     'neurofinder.04.00', 'neurofinder.04.01'] +
    ['neurofinder.%s.test' % s for s in ('00.00', '00.01', '01.00', '01.01', '02.00', '02.01', '03.00', '04.00',
                                         '04.01')])
NAME_TO_URL = dict((n, 'https://s3.amazonaws.com/neuro.datasets/challenges/neurofinder/%s.zip' % n)
                   for n in NEUROFINDER_NAMES)


def default_dirs():
    """(datasets_dir, checkpoints_dir) as /root/reference/deepcalcium/utils/config.py resolves them: the values of
    ~/.deep-calcium/deep-calcium.json when that file exists, else ~/.deep-calcium/{datasets,checkpoints}.  Read-only:
    unlike the reference, importing this module creates nothing."""
    base = '%s/.deep-calcium' % os.path.expanduser('~')
    cfg = {'datasets_dir': '%s/datasets' % base, 'checkpoints_dir': '%s/checkpoints' % base}
    try:
        with open('%s/deep-calcium.json' % base) as fp:
            cfg.update(json.load(fp))
    except (IOError, OSError, ValueError):
        pass
    return cfg['datasets_dir'], cfg['checkpoints_dir']


def _imread(p):
    from PIL import Image            # what scipy.misc.imread (nf.py:6) was: PIL's reader -> ndarray
    return np.array(Image.open(p))


def _to_int(a, dtype):
    """libhdf5's integer / float -> integer conversion on store: saturate at the target's limits."""
    info = np.iinfo(dtype)
    a = np.asarray(a)
    if a.dtype.kind == 'f':
        a = np.trunc(a)
    return np.clip(a, info.min, info.max).astype(dtype)


def _expand_names(names):
    if isinstance(names, str):
        low = names.lower()
        if low == 'all':
            return list(NEUROFINDER_NAMES)
        if low == 'all_train':
            return sorted(n for n in NEUROFINDER_NAMES if '.test' not in n)
        if low == 'all_test':
            return sorted(n for n in NEUROFINDER_NAMES if '.test' in n)
        return names.split(',')
    return list(names)


def _download(name, datasets_dir, logger):
    """nf.py:70-97: fetch and unpack the challenge zip when the directory is missing."""
    from zipfile import ZipFile
    try:
        import requests
    except ImportError:
        raise IOError('%s/%s is missing and `requests` is not importable: unpack %s there' %
                      (datasets_dir, name, NAME_TO_URL[name]))
    zip_path = '%s/%s.zip.%d' % (datasets_dir, name, os.getpid())
    logger.info('Downloading %s.zip.' % name)
    try:
        download = requests.get(NAME_TO_URL[name])
        download.raise_for_status()
    except Exception as e:
        raise IOError('%s/%s is missing and %s could not be fetched (%s): unpack the zip there by hand' %
                      (datasets_dir, name, NAME_TO_URL[name], e))
    with open(zip_path, 'wb') as fp:
        fp.write(download.content)
    # unpack next to the target and rename: nobody ever sees a half-extracted <datasets_dir>/<name>
    stage = '%s/.%s.extract.%d' % (datasets_dir, name, os.getpid())
    with ZipFile(zip_path, 'r') as z:
        z.extractall(stage)
    os.remove(zip_path)
    inner = '%s/%s' % (stage, name)
    try:
        os.rename(inner if os.path.isdir(inner) else stage, '%s/%s' % (datasets_dir, name))
    except OSError:
        if not os.path.isdir('%s/%s' % (datasets_dir, name)):
            raise
    shutil.rmtree(stage, ignore_errors=True)


def _run_token():
    """Identifies THIS launch to the ranks that wait on rank 0's dataset build: the launcher's run id / rendezvous endpoint
    (the same on every rank of a job, different for a relaunch)."""
    return (os.environ.get('TORCHELASTIC_RUN_ID') or 'run') + '@' + os.environ.get('MASTER_ADDR', '-') + ':' + os.environ.get('MASTER_PORT', '-')


def nf_load_hdf5(names, datasets_dir=None, wait_s=6 * 3600.0, stale_s=120.0):
    """Returns the list of `dataset.hdf5` paths for `names` ('all' | 'all_train' | 'all_test' | 'a,b' | list), building
    the files that do not exist yet.  nf.py:37-150.  Under data parallelism EVERY rank calls it (rank 0 builds, the others
    poll for the finished files for at most wait_s seconds, and give up after stale_s without a heartbeat from rank 0)."""
    logger = logging.getLogger('nf_load_hdf5')
    if datasets_dir is None:
        datasets_dir = '%s/neurons_nf' % default_dirs()[0]         # nf.py:37
    dataset_names = _expand_names(names)
    os.makedirs(datasets_dir, exist_ok=True)
    for name in dataset_names:
        url = NAME_TO_URL[name]          # KeyError for an unknown name, as in the reference (:72)
        del url
    # Under data-parallel fit() EVERY rank calls this.  Rank 0 alone downloads / unpacks / builds; the others never touch a
    # directory rank 0 may still be extracting: they poll (no collective: a first 'all' build takes far longer than any
    # process-group timeout) until every dataset.hdf5 exists -- os.replace() publishes a finished file atomically -- or rank
    # 0 leaves a failure marker.  Calling it from rank-0-only code is fine too (nobody waits on anybody).
    from . import parallel
    dataset_paths = ['%s/%s/dataset.hdf5' % (datasets_dir, name) for name in dataset_names]
    marker = '%s/.nf_build_failed' % datasets_dir
    beat = '%s/.nf_build_heartbeat' % datasets_dir
    token = _run_token()
    if parallel.rank() == 0:
        todo = [q for q in dataset_paths if not os.path.exists(q)]
        stop = threading.Event()
        th = None
        if todo and parallel.world_size() > 1:
            # liveness for the waiting ranks: a counter rank 0 bumps every second while it builds (they watch it CHANGE, by
            # their own monotonic clock: no cross-host clock comparison)
            def pulse():
                k = 0
                while not stop.is_set():
                    k += 1
                    try:
                        with open(beat + '.tmp', 'w') as fp:
                            fp.write('%s %d' % (token, k))
                        os.replace(beat + '.tmp', beat)
                    except OSError:
                        pass
                    stop.wait(1.0)
            th = threading.Thread(target=pulse, daemon=True)
            th.start()
        try:
            if os.path.exists(marker):
                os.remove(marker)
            for name, ds_path in zip(dataset_names, dataset_paths):
                if os.path.exists('%s/%s' % (datasets_dir, name)):
                    logger.info('%s already downloaded.' % name)
                else:
                    _download(name, datasets_dir, logger)
                if not os.path.exists(ds_path):
                    logger.info('Populating %s.' % ds_path)
                    _populate(name, '%s/%s' % (datasets_dir, name), ds_path)
        except Exception as e:           # tell the waiting ranks (this run's token, not a timestamp), then fail here
            try:
                with open(marker, 'w') as fp:
                    fp.write('%s\n%s: %s' % (token, type(e).__name__, e))
            except OSError:
                pass
            raise
        finally:
            stop.set()
            if th is not None:
                th.join(timeout=5)
                try:
                    os.remove(beat)
                except OSError:
                    pass
    elif parallel.world_size() > 1:
        t_start = time.monotonic()
        last_beat, last_change = None, time.monotonic()
        while not all(os.path.exists(q) for q in dataset_paths):
            if os.path.exists(marker):
                try:
                    head, _, msg = open(marker).read().partition('\n')
                except OSError:
                    head, msg = '', ''
                if head == token:
                    raise IOError('rank %d: rank 0 failed to build the datasets: %s' % (parallel.rank(), msg))
            try:
                cur = open(beat).read()
            except OSError:
                cur = None
            now = time.monotonic()
            if cur is not None and cur.split(' ')[0] == token and cur != last_beat:
                last_beat, last_change = cur, now
            # rank 0 gone (SIGKILL, OOM) or the directory is not shared: its heartbeat stopped / never showed up
            if now - last_change > stale_s:
                raise IOError('rank %d: no sign of life from rank 0 in %s for %.0f s (%s) -- is it running, and is the directory '
                              'shared between the ranks?' % (parallel.rank(), datasets_dir, stale_s,
                                                             'heartbeat stopped' if last_beat else 'no heartbeat seen'))
            if now - t_start > float(wait_s):
                raise IOError('rank %d: %s still missing after %.0f s' % (
                    parallel.rank(), ', '.join(q for q in dataset_paths if not os.path.exists(q)), float(wait_s)))
            time.sleep(0.2)
    missing = [q for q in dataset_paths if not os.path.exists(q)]
    if missing:
        raise IOError('rank %d: %s missing' % (parallel.rank(), ', '.join(missing)))
    return dataset_paths


def _populate(name, root, ds_path):
    s_paths = sorted(glob('%s/images/*.tiff' % root))
    if not s_paths:
        raise IOError('no TIFF frames under %s/images' % root)
    i_shape = _imread(s_paths[0]).shape
    n = len(s_paths)
    regions = None
    if '.test' not in name:
        with open('%s/regions/regions.json' % root) as fp:
            regions = json.load(fp)

    w = hdf5_min.Writer()
    w.attrs['name'] = name
    d_raw = w.create_dataset('series/raw', shape=(n,) + i_shape, dtype='int16')
    d_mean = w.create_dataset('series/mean', shape=i_shape, dtype='float16')
    d_max = w.create_dataset('series/max', shape=i_shape, dtype='int16')
    if regions is not None:
        d_mraw = w.create_dataset('masks/raw', shape=(len(regions),) + i_shape, dtype='int8')
        d_mmax = w.create_dataset('masks/max', shape=i_shape, dtype='int8')
    tmp = '%s.partial.%d' % (ds_path, os.getpid())      # concurrent builders (ranks) never share a half-written file
    w.save(tmp)

    raw = w.open_deferred(d_raw)
    mean = np.zeros(i_shape, np.float16)
    mx = np.zeros(i_shape, np.int16)
    with np.errstate(over='ignore'):
        for idx, p in enumerate(s_paths):
            img = _imread(p)
            raw[idx] = _to_int(img, np.int16)
            mean = (mean + (img * 1. / n)).astype(np.float16)          # one float16 rounding per frame
            mx = _to_int(np.maximum(mx, img), np.int16)
    raw.flush()
    del raw
    for d, a in ((d_mean, mean), (d_max, mx)):
        m = w.open_deferred(d)
        m[...] = a
        m.flush()
        del m
    if regions is not None:
        mraw = w.open_deferred(d_mraw)
        mmax = np.zeros(i_shape, np.int8)
        for idx, r in enumerate(regions):
            msk = np.zeros(i_shape)
            yy, xx = [c[0] for c in r['coordinates']], [c[1] for c in r['coordinates']]
            msk[yy, xx] = 1
            mraw[idx] = _to_int(msk, np.int8)
            mmax = _to_int(np.maximum(mmax, msk), np.int8)
        mraw.flush()
        del mraw
        m = w.open_deferred(d_mmax)
        m[...] = mmax
        m.flush()
        del m
    os.replace(tmp, ds_path)         # a crash mid-way leaves no half-written dataset.hdf5 behind
