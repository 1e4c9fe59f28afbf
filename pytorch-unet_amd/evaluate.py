"""Validation round with the reference's signature (evaluate.py:8-40) on the MI355X kernels.

    evaluate(net, dataloader, device, amp) -> mean Dice over the validation batches (a 0-dim tensor)

Same control flow as the reference: eval mode, inference_mode, autocast (bf16 -- the kernels implement bf16/fp32),
per batch the forward pass and the Dice score, train mode restored at the end.  For n_classes > 1 the reference's
one_hot(argmax) -> one_hot(mask) -> multiclass_dice_coeff(..[:, 1:]) chain (evaluate.py:31-36, five full-size fp32
temporaries) is ONE counting pass over the logits and the int64 mask (unetk_eval_dice).  The n_classes == 1 branch
is kept literally (evaluate.py:26-29), including its size assertion inside dice_coeff (the reference passes the
un-squeezed [N,1,H,W] prediction against the [N,H,W] mask).  The tqdm progress bar is not reproduced.
"""
import torch
import torch.nn.functional as F

from unetk import ops
from utils.dice_score import dice_coeff


@torch.inference_mode()
def evaluate(net, dataloader, device, amp):
    net.eval()
    num_val_batches = len(dataloader)
    dice_score = 0

    # iterate over the validation set
    with torch.autocast(device.type, dtype=torch.bfloat16, enabled=amp):
        for batch in dataloader:
            image, mask_true = batch['image'], batch['mask']

            # move images and labels to correct device and type
            image = image.to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
            mask_true = mask_true.to(device=device, dtype=torch.long)

            # predict the mask
            mask_pred = net(image)

            if net.n_classes == 1:
                assert mask_true.min() >= 0 and mask_true.max() <= 1, 'True mask indices should be in [0, 1]'
                mask_pred = (F.sigmoid(mask_pred) > 0.5).float()
                # compute the Dice score
                dice_score += dice_coeff(mask_pred, mask_true, reduce_batch_first=False)
            else:
                assert mask_true.min() >= 0 and mask_true.max() < net.n_classes, \
                    'True mask indices should be in [0, n_classes['
                # argmax -> one-hot -> Dice ignoring background, fused
                dice_score += ops.eval_dice(mask_pred, mask_true, first_class=1)

    net.train()
    return dice_score / max(num_val_batches, 1)
