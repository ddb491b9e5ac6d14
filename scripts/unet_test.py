#!/usr/bin/env python3
"""python scripts/unet_test.py --checkpoint_name unet_task03_liver_box_pseudolabels_2D.pth --images_dir ... --labels_dir ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd.testing import main  # noqa: E402

if __name__ == "__main__":
    raise SystemExit(main())
