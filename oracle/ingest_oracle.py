"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the image-ingest step that precedes the hot
path.  Only tests/ may import this; the product path never does.

Follows, in the reference:
    read_grayscale      src/utils/data_io.py:34-69    cv2.resize(image, (w_new, h_new)).astype('float32')
    process_resize      src/utils/data_io.py:71-86
    pad_bottom_right    src/utils/data_io.py:88-103
    grayscale2tensor    src/utils/data_io.py:105-106  image / 255.

The resize arithmetic is OpenCV's (opencv-python, pinned in the reference's requirements.txt, NOT
vendored under /root/reference and not installed here): cv::resize, INTER_LINEAR, CV_8UC1 -- fixed-point
coefficients (11 bits), horizontal pass in int, vertical pass
    dst = ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
and the INTER_AREA fast path when both scale factors are exactly 2.  There is no cv2 in this container
and the reference holds no image fixtures, so this restatement could not be checked against the library:
**parity unpinned** for the ingest row (the HIP kernel is tested bit-exactly against THIS file).
"""
import numpy as np


def process_resize(w, h, resize, df=None):
    """data_io.py:71-86"""
    if resize is not None:
        resize = tuple(resize)
        assert 0 < len(resize) <= 2
        if len(resize) == 1 and resize[0] > -1:
            scale = resize[0] / max(h, w)
            w_new, h_new = int(round(w * scale)), int(round(h * scale))
        elif len(resize) == 1 and resize[0] == -1:
            w_new, h_new = w, h
        else:
            w_new, h_new = resize[0], resize[1]
    else:
        w_new, h_new = w, h
    if df is not None:
        w_new, h_new = map(lambda x: int(x // df * df), [w_new, h_new])
    return w_new, h_new


def _taps(n_dst, n_src):
    inv = np.float64(n_dst) / np.float64(n_src)
    scale = np.float64(1.0) / inv
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= n_src - 1
    f[hi] = 0.0
    s[hi] = n_src - 1
    s1 = np.minimum(s + 1, n_src - 1)
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)   # cvRound: half to even
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, a0, a1


def resize_u8_linear(img, w_new, h_new):
    """cv2.resize(img, (w_new, h_new)) for a 2-D uint8 array (default INTER_LINEAR)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    if w == 2 * w_new and h == 2 * h_new:      # exact 2x2 decimation -> INTER_AREA fast path
        s = img.astype(np.int64)
        v = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
        return v.astype(np.uint8)
    sx0, sx1, ax0, ax1 = _taps(w_new, w)
    sy0, sy1, ay0, ay1 = _taps(h_new, h)
    s = img.astype(np.int64)
    r = s[:, sx0] * ax0[None, :] + s[:, sx1] * ax1[None, :]          # horizontal pass [h][w_new]
    r0, r1 = r[sy0], r[sy1]
    v = (((ay0[:, None] * (r0 >> 4)) >> 16) + ((ay1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def read_grayscale_u8(image_u8, resize=None, df=None, pad_to=None):
    """read_grayscale (data_io.py:34-69) from an already decoded 8-bit frame.
    -> (image [1, H, W] float32 in [0, 1], scales [2] float32 (h / h_new, w / w_new), mask [H, W] or None)"""
    h, w = image_u8.shape
    w_new, h_new = process_resize(w, h, resize, df)
    scales = np.array([float(h) / float(h_new), float(w) / float(w_new)], dtype=np.float32)
    img = resize_u8_linear(image_u8, w_new, h_new).astype(np.float32)
    mask = None
    if pad_to is not None:
        assert pad_to >= max(img.shape)
        padded = np.zeros((pad_to, pad_to), dtype=np.float32)
        padded[:h_new, :w_new] = img
        mask = np.zeros((pad_to, pad_to), dtype=np.float32)
        mask[:h_new, :w_new] = 1
        img = padded
    return (img / 255.0).astype(np.float32)[None], scales, mask
