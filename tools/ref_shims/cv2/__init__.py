"""Stand-in for the three OpenCV calls of bts_test.py / bts_eval.py (cv2.imread :158-161, cv2.imwrite :185 with
IMWRITE_PNG_COMPRESSION) on top of PIL.  BGR channel order like OpenCV; 16-bit single-channel PNGs round-trip as uint16.
Installed by tools/run_reference.py only when cv2 is absent."""
import numpy as np
from PIL import Image

IMWRITE_PNG_COMPRESSION = 16
IMREAD_UNCHANGED = -1


def imread(path, flags=1):
    try:
        im = Image.open(path)
    except (IOError, OSError):
        return None
    if flags == IMREAD_UNCHANGED or flags < 0:
        a = np.array(im)
        if a.ndim == 3:
            a = a[:, :, ::-1]
        if a.dtype == np.int32:
            a = a.astype(np.uint16)
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(np.array(im.convert("RGB"))[:, :, ::-1])


def imwrite(path, img, params=None):
    a = np.asarray(img)
    if a.ndim == 3:
        Image.fromarray(np.ascontiguousarray(a[:, :, ::-1]).astype(np.uint8)).save(path)
    elif a.dtype == np.uint16:
        Image.fromarray(a.astype(np.uint16)).save(path)          # mode I;16
    else:
        Image.fromarray(a.astype(np.uint8)).save(path)
    return True
