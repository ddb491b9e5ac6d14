"""unet_amd: the U-Net training / inference hot path of hakkini2/UNet as hand-written gfx950 (MI355X) HIP
kernels behind a C ABI, with the reference's Python surface (MONAI-style ``UNet``, ``DiceCELoss``,
``DiceMetric``, ``sliding_window_inference``) on top.  GPU only: no CPU or eager fallback.
"""
from .nets import UNet  # noqa: F401
from .losses import DiceCELoss, DiceLoss  # noqa: F401
from .metrics import DiceMetric, calculate_dice_score, confusion_counts  # noqa: F401
from .inferers import sliding_window_inference, window_starts, shard_range  # noqa: F401
from .transforms import Activations, AsDiscrete, Compose, decollate_batch  # noqa: F401
from .optim import AdamW  # noqa: F401
from .trainer import Trainer  # noqa: F401
from .evaluator import VolumeEvaluator  # noqa: F401
from .pipeline import SliceCropPipeline  # noqa: F401
from .utils import set_determinism, checkpoint_name, save_checkpoint, per_volume_dice, volume_id  # noqa: F401

__version__ = "0.1.0"
