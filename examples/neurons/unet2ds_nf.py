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

Datasets: names (`neurofinder.00.00`, comma lists, `all`, `all_train`, `all_test`) go through
`deep_calcium_amd.nf_datasets.nf_load_hdf5` (the reference's datasets/nf.py:37-150): <datasets_dir>/neurons_nf/<name>/
dataset.hdf5 is built from the unpacked challenge directory (images/*.tiff + regions/regions.json; the zip is fetched
first when the machine has network access); paths to existing .hdf5 / .npz dataset files are taken as they are.
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
from deep_calcium_amd.nf_datasets import nf_load_hdf5, default_dirs      # noqa: E402

DATASETS_DIR, CHECKPOINTS_DIR = default_dirs()                   # ~/.deep-calcium/deep-calcium.json or the defaults
CHECKPOINTS_DIR = '%s/neurons_unet2ds_nf' % CHECKPOINTS_DIR      # reference :16

np.random.seed(865)                 # examples/neurons/unet2ds_nf.py:18: the batch generator's stream
# :19 tf.set_random_seed(7535) seeds TF's dropout / initialisers; here 7535 is the default `seed` of unet_hip's engine
# (initial weights + the counter-hash dropout stream) -- TF's Philox stream itself is not reproducible outside TF
logging.basicConfig(level=logging.INFO)


def nf_find_hdf5(names, datasets_dir=os.path.join(DATASETS_DIR, 'neurons_nf')):
    """names -> dataset files.  Explicit paths of existing dataset .hdf5 / .npz files pass through; Neurofinder names go
    to nf_load_hdf5 (the reference's datasets/nf.py:37-150), which builds `<datasets_dir>/<name>/dataset.hdf5` from the
    unpacked challenge directory (images/*.tiff, regions/regions.json), fetching the zip first if it can."""
    if isinstance(names, str) and names.lower() not in ('all', 'all_train', 'all_test'):
        names = names.split(',')
    if not isinstance(names, str) and all(os.path.isfile(n) for n in names):
        return list(names)
    try:
        return nf_load_hdf5(names, datasets_dir=datasets_dir)
    except (IOError, KeyError) as e:
        raise SystemExit('dataset %r: %s' % (names, e))


def training(dataset_name, model_path, checkpoints_dir, nb_steps=100, nb_epochs=10):
    """Train on neurofinder datasets (reference :23-44, same hyper-parameters; --nb_steps / --nb_epochs are additions
    for short runs, the defaults are the reference's 100 x 10)."""
    parallel.init_from_env()                       # no-op unless launched with torch.distributed.run
    dspaths = nf_find_hdf5(dataset_name)           # every rank calls it: rank 0 builds missing dataset files, the others wait inside
    model = UNet2DSummary(cpdir=checkpoints_dir)
    return model.fit(
        dspaths,
        model_path=model_path,
        shape_trn=(128, 128),
        shape_val=(512, 512),
        batch_size_trn=20,
        nb_steps_trn=nb_steps,
        nb_epochs=nb_epochs,
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
    sp_trn.add_argument('--nb_steps', help='training batches per epoch', default=100, type=int)
    sp_trn.add_argument('--nb_epochs', help='epochs', default=10, type=int)
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
