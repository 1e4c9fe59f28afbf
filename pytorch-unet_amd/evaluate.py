"""Validation round with the reference's signature (evaluate.py:8-40) on the MI355X kernels.

    evaluate(net, dataloader, device, amp) -> mean Dice over the validation batches (a 0-dim tensor)

Same control flow as the reference: eval mode, inference_mode, autocast (`amp_dtype`: bf16 by default, as
BASELINE.json's configurations; torch.float16 is what the reference's bare `torch.autocast(device.type)` picks on a GPU),
per batch the forward pass and the Dice score, train mode restored at the end.  For n_classes > 1 the reference's
one_hot(argmax) -> one_hot(mask) -> multiclass_dice_coeff(..[:, 1:]) chain (evaluate.py:31-36, five full-size fp32
temporaries) is ONE counting pass over the logits and the int64 mask (unetk_eval_dice).  The n_classes == 1 branch
is kept literally (evaluate.py:26-29), including its size assertion inside dice_coeff (the reference passes the
un-squeezed [N,1,H,W] prediction against the [N,H,W] mask).  The tqdm progress bar is not reproduced.
"""
import torch

from unetk import ops
from utils.dice_score import dice_coeff


def _label_range_ok(labels, upper):
    lo, hi = torch.aminmax(labels)
    return bool(lo >= 0) and bool(hi <= upper)


def _score_binary(logits, labels):
    """Single-logit nets (evaluate.py:26-30): threshold the sigmoid at 0.5 and hand the [N,1,H,W] prediction and the
    [N,H,W] labels to dice_coeff exactly as the reference does -- shapes included, so dice_coeff's own size check
    fires just as it does there."""
    assert _label_range_ok(labels, 1), f'binary segmentation labels must be 0 or 1, got up to {int(labels.max())}'
    hard = torch.sigmoid(logits).gt(0.5).float()
    return dice_coeff(hard, labels, reduce_batch_first=False)


def _score_multiclass(logits, labels, n_classes):
    """evaluate.py:31-37 without its one-hot temporaries: unetk_eval_dice counts, per image and foreground class,
    |pred == k|, |label == k| and their overlap in one pass over logits and labels, then averages the Dice ratios."""
    assert _label_range_ok(labels, n_classes - 1), \
        f'labels must lie in 0..{n_classes - 1} for a {n_classes}-class net, got up to {int(labels.max())}'
    return ops.eval_dice(logits, labels, first_class=1)


@torch.inference_mode()
def evaluate(net, dataloader, device, amp, amp_dtype=torch.bfloat16):
    net.eval()
    total, n_batches = 0, len(dataloader)
    with torch.autocast(device.type, dtype=amp_dtype, enabled=amp):
        for sample in dataloader:
            frames = sample['image'].to(device=device, dtype=torch.float32, memory_format=torch.channels_last)
            labels = sample['mask'].to(device=device, dtype=torch.long)
            logits = net(frames)
            if net.n_classes == 1:
                total = total + _score_binary(logits, labels)
            else:
                total = total + _score_multiclass(logits, labels, net.n_classes)
    # the reference unconditionally returns the net to train mode (evaluate.py:39); so does this
    net.train()
    return total / max(n_batches, 1)
