#!/usr/bin/env python
"""Neurofinder training, evaluation and prediction with the HIP UNet2DS path: the reference's example
(/root/reference/examples/neurons/unet2ds_nf.py) with the same three actions, arguments, seeds and call sequence, on
`deep_calcium_amd.UNet2DSummary`.

    python examples/neurons/unet2ds_nf.py evaluate neurofinder.00.00 --model unet2ds_model.hdf5
    python examples/neurons/unet2ds_nf.py train all_train [-m model.hdf5] [-c checkpoints_dir]
    python examples/neurons/unet2ds_nf.py predict all_test --model unet2ds_model.hdf5

`--model` takes the reference's own files: the released Keras `unet2ds_model.hdf5`
(unet_2d_summary.py:28), any Keras ModelCheckpoint file, or a checkpoint written by this build.  With the released
weights and the Neurofinder data in place, `evaluate neurofinder.00.00` is the one command whose output the reference's
README quotes (prec=0.976, reca=1.000, comb=0.988 with TTA; 0.919 / 1.000 / 0.958 without: /root/reference/README.md:30-37)
-- neither is available offline, so that comparison cannot be run inside the build container.

Datasets: names (`neurofinder.00.00`, comma lists, `all`, `all_train`, `all_test`) resolve to the HDF5 files the
reference's `nf_load_hdf5` leaves at <datasets_dir>/neurons_nf/<name>/dataset.hdf5 (datasets/nf.py:37-150); paths to
.hdf5 / .npz dataset files are taken as they are.  Downloading and TIFF preprocessing are out of scope.
Multi-GPU training: launch with `python -m torch.distributed.run --nproc-per-node N examples/neurons/unet2ds_nf.py train ...`.
"""
from time import time
import argparse
import logging
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

from deep_calcium_amd import UNet2DSummary, parallel             # noqa: E402
from deep_calcium_amd.nf_metrics import nf_submit                # noqa: E402

BASE_DIR = os.path.join(os.path.expanduser('~'), '.deep-calcium')
DATASETS_DIR = os.path.join(BASE_DIR, 'datasets')
CHECKPOINTS_DIR = os.path.join(BASE_DIR, 'checkpoints', 'neurons_unet2ds_nf')

NEUROFINDER_NAMES = sorted(
    ['neurofinder.00.%02d' % i for i in range(12)] +
    ['neurofinder.01.00', 'neurofinder.01.01', 'neurofinder.02.00', 'neurofinder.02.01', 'neurofinder.03.00',
     'neurofinder.04.00', 'neurofinder.04.01'] +
    ['neurofinder.%s.test' % s for s in ('00.00', '00.01', '01.00', '01.01', '02.00', '02.01', '03.00', '04.00', '04.01')])

np.random.seed(865)                 # examples/neurons/unet2ds_nf.py:18: the batch generator's stream
# :19 tf.set_random_seed(7535) seeds TF's dropout / initialisers; here 7535 is the default `seed` of unet_hip's engine
# (initial weights + the counter-hash dropout stream) -- TF's Philox stream itself is not reproducible outside TF
logging.basicConfig(level=logging.INFO)


def nf_find_hdf5(names, datasets_dir=os.path.join(DATASETS_DIR, 'neurons_nf')):
    """The path-resolution half of the reference's nf_load_hdf5 (datasets/nf.py:56-66, :111): names -> dataset files."""
    if isinstance(names, str) and names.lower() == 'all':
        names = NEUROFINDER_NAMES
    elif isinstance(names, str) and names.lower() == 'all_train':
        names = [n for n in NEUROFINDER_NAMES if '.test' not in n]
    elif isinstance(names, str) and names.lower() == 'all_test':
        names = [n for n in NEUROFINDER_NAMES if '.test' in n]
    elif isinstance(names, str):
        names = names.split(',')
    paths = []
    for n in names:
        p = n if os.path.exists(n) else '%s/%s/dataset.hdf5' % (datasets_dir, n)
        if not os.path.exists(p):
            raise SystemExit('dataset %r: %s not found (downloading / TIFF preprocessing is not part of this build: create '
                             'it with the reference\'s nf_load_hdf5, or pass the path of a dataset .hdf5 / .npz file)' % (n, p))
        paths.append(p)
    return paths


def training(dataset_name, model_path, checkpoints_dir):
    """Train on neurofinder datasets (reference :23-44, same hyper-parameters)."""
    parallel.init_from_env()                       # no-op unless launched with torch.distributed.run
    dspaths = nf_find_hdf5(dataset_name)
    model = UNet2DSummary(cpdir=checkpoints_dir)
    return model.fit(
        dspaths,
        model_path=model_path,
        shape_trn=(128, 128),
        shape_val=(512, 512),
        batch_size_trn=20,
        nb_steps_trn=100,
        nb_epochs=10,
        keras_callbacks=[],
        prop_trn=0.75,
        prop_val=0.25,
    )


def evaluation(dataset_name, model_path, checkpoints_dir):
    """Evaluate datasets -- once with test-time augmentation and once without (reference :47-64)."""
    logger = logging.getLogger('evaluation')
    ds_trn = nf_find_hdf5(dataset_name)
    model = UNet2DSummary(cpdir=checkpoints_dir)
    for aug in [True, False]:
        logger.info('Evaluation with%s.' % (' TTA' if aug else 'out TTA'))
        model.predict(ds_trn, model_path=model_path, window_shape=(512, 512), save=True, print_scores=True, augmentation=aug)


def prediction(dataset_name, model_path, checkpoints_dir):
    """Predictions + Neurofinder submission files with and without test-time augmentation (reference :67-96)."""
    logger = logging.getLogger('prediction')
    ds_tst = nf_find_hdf5(dataset_name)
    model = UNet2DSummary(cpdir=checkpoints_dir)
    tic = int(time())
    for aug in [True, False]:
        logger.info('Prediction with%s.' % (' TTA' if aug else 'out TTA'))
        Mp, names = model.predict(ds_tst, model_path=model_path, window_shape=(512, 512), save=False, augmentation=aug)
        Mp = [m.round() for m in Mp]
        nf_submit(Mp, names, '%s/submission_%d%s.json' % (model.cpdir, tic, ('_TTA' if aug else '')))
        nf_submit(Mp, names, '%s/submission_latest%s.json' % (model.cpdir, ('_TTA' if aug else '')))


if __name__ == '__main__':
    ap = argparse.ArgumentParser(description='CLI for UNet2DS model.')
    sp = ap.add_subparsers(title='actions', description='Choose an action.')
    sp_trn = sp.add_parser('train', help='CLI for training.')
    sp_trn.set_defaults(which='train')
    sp_trn.add_argument('dataset_name', help='dataset name', default='all_train', type=str)
    sp_trn.add_argument('-m', '--model_path', help='path to model')
    sp_trn.add_argument('-c', '--checkpoints_dir', help='checkpoint directory', default=CHECKPOINTS_DIR)
    sp_eva = sp.add_parser('evaluate', help='CLI for evaluation.')
    sp_eva.set_defaults(which='evaluate')
    sp_eva.add_argument('dataset_name', help='dataset name', default='all_train', type=str)
    sp_eva.add_argument('-m', '--model_path', help='path to model', required=True)
    sp_eva.add_argument('-c', '--checkpoints_dir', help='checkpoint directory', default=CHECKPOINTS_DIR)
    sp_prd = sp.add_parser('predict', help='CLI for prediction.')
    sp_prd.set_defaults(which='predict')
    sp_prd.add_argument('dataset_name', help='dataset name', default='all', type=str)
    sp_prd.add_argument('-m', '--model_path', help='path to model', required=True)
    sp_prd.add_argument('-c', '--checkpoints_dir', help='checkpoint directory', default=CHECKPOINTS_DIR)
    args = vars(ap.parse_args())
    if 'which' not in args:
        ap.error('choose an action: train, evaluate or predict')
    f = {'train': training, 'evaluate': evaluation, 'predict': prediction}[args.pop('which')]
    f(**args)
