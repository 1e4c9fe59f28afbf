"""predict_img with the reference's signature (predict.py:15-33) on the MI355X kernels.

    predict_img(net, full_img, device, scale_factor=1, out_threshold=0.5) -> numpy int64 mask [H, W]

Same steps as the reference: eval mode, BasicDataset.preprocess (PIL resize + /255), one forward pass under no_grad;
the reference then copies the logits to the CPU, resizes them to the full image size with F.interpolate(bilinear)
and takes argmax / sigmoid > threshold -- here that tail is one kernel on the GPU (unetk_predict_mask) and only the
int64 mask crosses PCIe.  `mask_to_image` (predict.py:62-76) is host-side numpy, as in the reference.  The CLI
around it (argparse, file names, plotting) is control plane and not provided.
"""
import numpy as np
import torch
from PIL import Image

from unetk import ops
from utils.data_loading import BasicDataset


def predict_img(net, full_img, device, scale_factor=1, out_threshold=0.5):
    net.eval()
    img = torch.from_numpy(BasicDataset.preprocess(None, full_img, scale_factor, is_mask=False))
    img = img.unsqueeze(0)
    img = img.to(device=device, dtype=torch.float32)

    with torch.no_grad():
        output = net(img)
        mask = ops.predict_mask(output, full_img.size[1], full_img.size[0], out_threshold)

    return mask[0].long().squeeze().cpu().numpy()


def mask_to_image(mask: np.ndarray, mask_values):
    """Class-index map (or [C,H,W] scores) -> PIL image in the dataset's own mask encoding (predict.py:62-76):
    RGB when the mask values are colour triples, 1-bit for the [0, 1] Carvana case, 8-bit grey otherwise."""
    height, width = mask.shape[-2], mask.shape[-1]
    if isinstance(mask_values[0], list):
        canvas = np.zeros((height, width, len(mask_values[0])), dtype=np.uint8)
    else:
        canvas = np.zeros((height, width), dtype=bool if mask_values == [0, 1] else np.uint8)
    classes = mask.argmax(axis=0) if mask.ndim == 3 else mask
    for index, value in enumerate(mask_values):
        canvas[classes == index] = value
    return Image.fromarray(canvas)
