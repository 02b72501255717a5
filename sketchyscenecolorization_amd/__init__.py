"""MI355X-native (gfx950) implementation of the SketchySceneColorization
Foreground_Instance_Colorization generator/discriminator hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed
only); all arithmetic runs in hand-written HIP kernels behind the C ABI declared
in ``include/sketchycolor_hip.h`` (``lib/libsketchycolor_hip.so``).
"""
__version__ = '0.1.0'

import os as _os

# before the HIP runtime starts (no effect once it has): kernel arguments in device memory
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
