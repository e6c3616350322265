"""deep_calcium_amd: the UNet2DS hot path of alexklibisz/deep-calcium, MI355X-native.

    from deep_calcium_amd import UNet2DSummary, unet_hip

`UNet2DSummary` keeps the reference's fit()/predict() surface; `unet_hip` is the drop-in `net_builder_func`.
Every tensor op runs as a hand-written gfx950 HIP kernel behind the C ABI in include/dcunet.h; there is no
CPU fallback (importing is cheap and GPU-free, constructing a model requires the GPU and libdcunet.so).
"""
__version__ = '0.1.0'

from .model import (Model, Adam, Callback, CSVLogger, ModelCheckpoint, ReduceLROnPlateau, History,  # noqa: F401
                    unet_hip, load_model_with_new_input_shape, metrics_from_sums)
from .unet2ds import UNet2DSummary, INVERTIBLE_2D_AUGMENTATIONS, _ValidationMetricsCB  # noqa: F401
from .nf_metrics import nf_mask_metrics  # noqa: F401
