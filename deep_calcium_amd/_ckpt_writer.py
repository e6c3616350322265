"""Checkpoint-writer process of Model.save(background=True).

The reference's ModelCheckpoint (unet_2d_summary.py:423-424, save_best_only=False) writes the whole model -- 31 MB of
parameters + 62 MB of Adam state -- at the end of EVERY epoch; at this build's speed an epoch of the example configuration
is 0.3 s.  Serialising the Keras HDF5 file (keras_io / hdf5_min: pure Python + numpy, ~0.3 s of CPU with the GIL held) on a
thread of the training process would take that time out of the step loop's host thread, so it happens HERE: the training
process dumps its pinned snapshot as one raw file (one GIL-free write into /dev/shm) and sends a JSON line; this process
(torch-free: numpy + hdf5_min only) splits the flat buffers, writes `<path>.partial.<pid>`, renames it into place, deletes the
raw file and answers with one status line.

    python -m deep_calcium_amd._ckpt_writer      (started by Model, stdin/stdout are the protocol)
"""
import json
import os
import sys

import numpy as np


def write_checkpoint(filepath, config, p, s, m, v, meta):
    """Shared by the in-process (blocking) save and this process: flat host arrays -> the file."""
    from . import keras_io
    from .layers import assign_offsets, build_layer_table, split_optimizer, split_weights
    layers = build_layer_table(config['nb_filters_base'], config.get('prop_dropout_base', 0.25),
                               config.get('upsampling_or_transpose', 'transpose') != 'transpose')
    assign_offsets(layers)
    weights = split_weights(layers, p, s)
    with_opt = m is not None
    tmp = '%s.partial.%d' % (filepath, os.getpid())      # a reader never sees a half-written checkpoint
    if str(filepath).lower().endswith(('.hdf5', '.h5')):
        opt = None
        if with_opt:
            ms, vs = split_optimizer(layers, m, v)
            opt = dict(config=meta['opt_config'], iterations=int(meta['iterations']), m=ms, v=vs)
        keras_io.write_keras_model(tmp, weights, config, optimizer=opt, loss=meta['loss'], metrics=meta['metrics'])
    else:
        arrays = {'w_%03d' % i: w for i, w in enumerate(weights)}
        head = dict(format='dcunet-npz-1', config=config, compiled=bool(meta['compiled']))
        if with_opt:
            arrays['opt_m'], arrays['opt_v'] = np.array(m, copy=True), np.array(v, copy=True)
            head['optimizer'] = dict(meta['opt_config'], iterations=int(meta['iterations']))
            head['loss'] = meta['loss']
        arrays['meta'] = np.frombuffer(json.dumps(head).encode(), dtype=np.uint8)
        with open(tmp, 'wb') as fp:
            np.savez(fp, **arrays)
    os.replace(tmp, filepath)


def main():
    out = sys.stdout
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        job = json.loads(line)
        raw = job['raw']
        try:
            sizes = job['sizes']                               # floats of p, s, m, v (m, v: 0 without optimizer state)
            flat = np.fromfile(raw, dtype=np.float32)
            if flat.size != sum(sizes):
                raise IOError('%s holds %d floats, expected %d' % (raw, flat.size, sum(sizes)))
            cuts = np.cumsum([0] + sizes)
            p, s, m, v = (flat[cuts[i]:cuts[i + 1]] for i in range(4))
            write_checkpoint(job['path'], job['config'], p, s, m if sizes[2] else None, v if sizes[3] else None, job['meta'])
            out.write('ok %s\n' % job['path'])
        except BaseException as e:
            out.write('error %s: %s\n' % (type(e).__name__, str(e).replace('\n', ' ')))
        finally:
            try:
                os.remove(raw)
            except OSError:
                pass
        out.flush()


if __name__ == '__main__':
    main()
