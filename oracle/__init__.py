"""CPU oracle for the U-Net hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain ``torch.nn`` (CPU, fp32) restatement of the arithmetic
the reference's ``train.py`` drives through MONAI: the residual ``UNet``
(reference/train.py:78-86), ``DiceCELoss`` (train.py:88), ``DiceMetric``
(train.py:92), the ``Activations``/``AsDiscrete`` post transform (train.py:89)
and ``sliding_window_inference`` (train.py:140).

MONAI itself is a third-party dependency that is neither vendored in the
reference tree nor installed in this image (the reference pins no version; its
era implies MONAI 1.3.x), so its published algorithm is restated here from the
MONAI 1.3 sources.  The reference holds no tests, fixtures or golden vectors
for this path, therefore:

    PARITY UNPINNED against the reference's own outputs.

What pins this restatement instead: (1) the widely published parameter count
of MONAI's spleen-tutorial UNet (4 808 917), (2) closed-form loss / metric
values, (3) the state_dict key layout MONAI produces, and (4) the fact that
every module used is the very ``torch.nn`` class MONAI instantiates.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product package ``unet_amd`` never
imports it and has no CPU fallback.
"""

from .unet_ref import UNet, ResidualUnit, Convolution, ADN, SkipConnection  # noqa: F401
from .losses_ref import (  # noqa: F401
    DiceLoss,
    DiceCELoss,
    DiceMetric,
    Activations,
    AsDiscrete,
    Compose,
    decollate_batch,
    calculate_dice_score,
    set_determinism,
)
from .inferer_ref import sliding_window_inference, dense_patch_starts  # noqa: F401
