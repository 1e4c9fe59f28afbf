"""Input formatting of the hot path's callers (reference utils/data_loading.py:65-91).

Only `BasicDataset.preprocess` -- the one piece of the reference's dataset class that predict_img() (predict.py:21)
and the training loader share -- is provided: PIL image -> resized float CHW array in [0,1] (BICUBIC), or PIL mask ->
int64 class-index map (NEAREST).  File discovery, the multiprocessing scan for mask values and the Dataset
plumbing are control plane around the path.

`BasicDataset.preprocess_gpu` is the same function on decoded uint8 pixels already in HBM (one frame or a batch):
bit-exact with the PIL path (unetk/preprocess.py, csrc/preprocess.hip), for loaders that ship raw frames to the GPU.
"""
import numpy as np
from PIL import Image


class BasicDataset:
    @staticmethod
    def preprocess(mask_values, pil_img, scale, is_mask):
        w, h = pil_img.size
        new_w, new_h = int(scale * w), int(scale * h)
        assert new_w > 0 and new_h > 0, 'Scale is too small, resized images would have no pixel'
        pil_img = pil_img.resize((new_w, new_h), resample=Image.NEAREST if is_mask else Image.BICUBIC)
        img = np.asarray(pil_img)
        if is_mask:
            mask = np.zeros((new_h, new_w), dtype=np.int64)
            for i, v in enumerate(mask_values):
                hit = (img == v) if img.ndim == 2 else (img == v).all(-1)
                mask[hit] = i
            return mask
        img = img[np.newaxis, ...] if img.ndim == 2 else img.transpose((2, 0, 1))
        if (img > 1).any():
            img = img / 255.0
        return img

    @staticmethod
    def preprocess_gpu(mask_values, pixels, scale, is_mask):
        """preprocess() for a CUDA uint8 tensor `pixels` = np.asarray(pil_img) ([H,W] / [H,W,C], or [N,H,W,C]).
        Returns CUDA tensors: float32 [C,H',W'] (NHWC-backed) or int64 [H',W']; a leading N for batches."""
        from unetk import preprocess as P
        if is_mask:
            return P.preprocess_mask(pixels, mask_values, scale)
        return P.preprocess_image(pixels, scale)
