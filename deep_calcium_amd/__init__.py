"""deep_calcium_amd: the UNet2DS hot path of alexklibisz/deep-calcium, MI355X-native.

    from deep_calcium_amd import UNet2DSummary, unet_hip

`UNet2DSummary` keeps the reference's fit()/predict() surface; `unet_hip` is the drop-in `net_builder_func`.
Every tensor op runs as a hand-written gfx950 HIP kernel behind the C ABI in include/dcunet.h; there is no
CPU fallback (importing is cheap and GPU-free, constructing a model requires the GPU and libdcunet.so).
The public names are resolved on first use (PEP 562): importing a torch-free submodule (`deep_calcium_amd.keras_io`,
`.nf_metrics`, `.layers` -- what the background checkpoint-writer process loads) does not import torch.
"""
import os as _os

__version__ = '0.1.0'

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A data-parallel rank holds more than
# that -- the step's main and weight-gradient streams, the input copy stream, torch.distributed's collective stream, RCCL's own
# per-communicator streams -- and when the weight-gradient stream ends up sharing the main stream's queue the two-stream backward
# serialises (measured in round 6: 17.5 -> 18.2-18.6 ms per step, every launch of the trace on one queue; 17.47 ms with 8 queues, and
# no change for a single-GPU run: 17.41 vs 17.42).  Only a default: an explicit setting wins, and it takes effect only if the HIP
# runtime has not initialised yet (it reads the variable once) -- UNetEngine also makes its side stream first, for that case.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

_EXPORTS = {
    'model': ('Model', 'Adam', 'Callback', 'CSVLogger', 'ModelCheckpoint', 'ReduceLROnPlateau', 'History', 'unet_hip',
              'load_model_with_new_input_shape', 'metrics_from_sums'),
    'unet2ds': ('UNet2DSummary', 'INVERTIBLE_2D_AUGMENTATIONS', '_ValidationMetricsCB'),
    'nf_metrics': ('nf_mask_metrics',),
}
_WHERE = dict((name, mod) for mod, names in _EXPORTS.items() for name in names)
__all__ = sorted(_WHERE)


def __getattr__(name):
    mod = _WHERE.get(name)
    if mod is None:
        raise AttributeError('module %r has no attribute %r' % (__name__, name))
    import importlib
    value = getattr(importlib.import_module('.' + mod, __name__), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + list(_WHERE))
