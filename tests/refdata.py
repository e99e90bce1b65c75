"""The two images the reference's extractor tests read (test/data/equirectangular_image_00{1,2}.jpg,
committed unchanged under tests/golden/), decoded to 8-bit grey the way cv::imread(...,
IMREAD_GRAYSCALE) does it: the luma plane straight from the JPEG decoder."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def equirectangular(n):
    from PIL import Image
    im = Image.open(GOLDEN / ("equirectangular_image_%03d.jpg" % n))
    im.draft("L", im.size)
    return np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))


def toy_sample(width, height, x0, y0, x1, y1):
    """White image with a filled black rectangle, corners inclusive (cv::rectangle, thickness -1)."""
    img = np.full((height, width), 255, np.uint8)
    img[y0:min(y1, height - 1) + 1, x0:min(x1, width - 1) + 1] = 0
    return img
