"""Query-image ingest on the device: the decoded 8-bit grayscale frame is uploaded
(pinned, async) and resized / normalised by `opp_image_ingest_u8`, replacing the per-image
`cv2.resize` + `/255.` + fp32 H2D of the reference:

    read_grayscale      src/utils/data_io.py:34-69
    process_resize      src/utils/data_io.py:71-86
    pad_bottom_right    src/utils/data_io.py:88-103
    grayscale2tensor    src/utils/data_io.py:105-106

Decoding the file (cv2.imread) stays with the caller: this module starts from the 8-bit array.
"""
import numpy as np
import torch

from . import _lib


def process_resize(w, h, resize, df=None):
    """Same rule as the reference (data_io.py:71-86): `resize` = None | [long_side] | [-1] | [w, h];
    `df` rounds both sides down to a multiple."""
    if resize is not None:
        resize = tuple(resize)
        if not 0 < len(resize) <= 2:
            raise AssertionError("resize must have 1 or 2 entries")
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


def read_grayscale_u8(image_u8, resize=None, df=None, pad_to=None, ret_scales=False, ret_pad_mask=False,
                      device=None, stream=None):
    """`read_grayscale` of the reference from an already decoded frame.

    image_u8: [h, w] uint8 -- numpy array, CPU tensor (uploaded; pinned memory makes the copy async) or
    device tensor.  Returns the image as a DEVICE tensor [1, H, W] float32 in [0, 1] and, like the
    reference, optionally `scales` ([2] = (h / h_new, w / w_new), CPU) and the pad mask ([pad, pad] float32,
    device, or None without `pad_to`)."""
    lib = _lib.load()
    if isinstance(image_u8, np.ndarray):
        image_u8 = torch.from_numpy(np.ascontiguousarray(image_u8))
    if image_u8.dtype != torch.uint8 or image_u8.dim() != 2:
        raise ValueError("read_grayscale_u8 expects a [h, w] uint8 image")
    if device is None:
        device = image_u8.device if image_u8.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("the ingest kernel runs on the GPU only (no CPU fallback)")
    h, w = image_u8.shape
    w_new, h_new = process_resize(w, h, resize, df)
    scales = torch.tensor([float(h) / float(h_new), float(w) / float(w_new)])
    with torch.cuda.device(device):
        st = torch.cuda.current_stream(device) if stream is None else stream
        with torch.cuda.stream(st):
            src = image_u8.to(device, non_blocking=True).contiguous()
            if pad_to is not None:
                if not (isinstance(pad_to, int) and pad_to >= max(h_new, w_new)):
                    raise AssertionError("pad_to must be an int >= the resized image")
                out = torch.zeros(1, pad_to, pad_to, dtype=torch.float32, device=device)
                stride = pad_to
            else:
                out = torch.empty(1, h_new, w_new, dtype=torch.float32, device=device)
                stride = w_new
            _lib.check(lib.opp_image_ingest_u8(src.data_ptr(), h, w, src.stride(0), h_new, w_new, out.data_ptr(), stride,
                                               None, st.cuda_stream), "opp_image_ingest_u8")
            mask = None
            if pad_to is not None and ret_pad_mask:
                mask = torch.zeros(pad_to, pad_to, dtype=torch.float32, device=device)
                mask[:h_new, :w_new] = 1
            src.record_stream(st)
    ret = [out]
    if ret_scales:
        ret.append(scales)
    if ret_pad_mask:
        ret.append(mask)
    return ret[0] if len(ret) == 1 else ret
