"""Minimal ``cv2`` stand-in for importing the reference's dataset helpers in the build container (TEST INFRASTRUCTURE;
OpenCV is not installed and cannot be).  Only what ``lib/datasets/enerf_utils.py`` touches: ``resize`` with
INTER_AREA / INTER_NEAREST.  Ray generation depends on the resized SHAPE (and, in the training branch, on the
nearest-resized mask), not on interpolated image values: INTER_NEAREST follows OpenCV's rule
(src index = floor(dst * src/dst), clipped), INTER_AREA is a plain block mean for integer factors."""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_AREA = 0, 1, 3


def resize(img, dsize, fx=None, fy=None, interpolation=INTER_LINEAR):
    h, w = img.shape[:2]
    if dsize is None:
        ow, oh = int(round(w * fx)), int(round(h * fy))      # cv::resize: saturate_cast<int>(size * f)
    else:
        ow, oh = dsize
    if interpolation == INTER_NEAREST:
        ys = np.minimum((np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
        xs = np.minimum((np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
        return img[ys][:, xs]
    fy_, fx_ = h // oh, w // ow
    if fy_ * oh == h and fx_ * ow == w:
        return img.reshape(oh, fy_, ow, fx_, *img.shape[2:]).mean(axis=(1, 3)).astype(img.dtype)
    ys = np.minimum((np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def setNumThreads(n):          # lib/datasets/make_dataset.py:12
    return None
