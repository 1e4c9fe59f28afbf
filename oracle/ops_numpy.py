"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of every op on the U-Net hot path.

Layout is NHWC throughout (the layout the HIP kernels use); weights keep PyTorch's layouts
(`[Cout,Cin,kh,kw]` for conv, `[Cin,Cout,2,2]` for the transposed conv) because that is what
the state_dict holds.  Every function cites the reference line whose ATen
op it restates.  Math runs in the dtype of its inputs (use float64 inputs for a tight check).

Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by
importing the reference's modules (oracle/make_golden.py).
"""
import numpy as np


# --------------------------------------------------------------------------- conv 3x3
def conv3x3_fwd(x, w):
    """unet_parts.py:16,19 -- Conv2d(k=3, padding=1, bias=False).  x[N,H,W,Ci], w[Co,Ci,3,3]."""
    n, h, wd, ci = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros((n, h, wd, w.shape[0]), dtype=np.result_type(x, w))
    for kh in range(3):
        for kw in range(3):
            y += xp[:, kh:kh + h, kw:kw + wd, :] @ w[:, :, kh, kw].T
    return y


def conv3x3_bwd(x, w, dy):
    """Gradients of conv3x3_fwd: dx (dgrad) and dw (wgrad)."""
    n, h, wd, ci = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dxp = np.zeros_like(xp, dtype=np.result_type(dy, w))
    dw = np.zeros_like(w, dtype=np.result_type(dy, x))
    dyf = dy.reshape(-1, dy.shape[-1])
    for kh in range(3):
        for kw in range(3):
            patch = xp[:, kh:kh + h, kw:kw + wd, :]
            dw[:, :, kh, kw] = dyf.T @ patch.reshape(-1, ci)
            dxp[:, kh:kh + h, kw:kw + wd, :] += dy @ w[:, :, kh, kw]
    return dxp[:, 1:-1, 1:-1, :], dw


# --------------------------------------------------------------------------- batch norm + relu
def bn_train_fwd(y, gamma, beta, eps=1e-5):
    """unet_parts.py:17,20 -- BatchNorm2d in train mode: biased batch variance, affine."""
    m = y.reshape(-1, y.shape[-1])
    mean = m.mean(axis=0)
    var = m.var(axis=0)
    invstd = 1.0 / np.sqrt(var + eps)
    return (y - mean) * invstd * gamma + beta, mean, var, invstd


def bn_running_update(run_mean, run_var, mean, var, count, momentum=0.1):
    """BatchNorm2d buffers: momentum 0.1, *unbiased* variance n/(n-1) (torch defaults)."""
    unbiased = var * (count / max(count - 1, 1))
    return (1 - momentum) * run_mean + momentum * mean, (1 - momentum) * run_var + momentum * unbiased


def bn_eval_fwd(y, gamma, beta, run_mean, run_var, eps=1e-5):
    return (y - run_mean) / np.sqrt(run_var + eps) * gamma + beta


def relu(x):
    """unet_parts.py:18,21."""
    return np.maximum(x, 0)


def bn_relu_train_bwd(y, gamma, mean, invstd, act, dact):
    """Backward through ReLU then train-mode BN.  act = relu(bn(y)); dact = dL/dact."""
    dz = dact * (act > 0)
    m = y.reshape(-1, y.shape[-1])
    cnt = m.shape[0]
    xhat = (y - mean) * invstd
    dbeta = dz.reshape(cnt, -1).sum(axis=0)
    dgamma = (dz * xhat).reshape(cnt, -1).sum(axis=0)
    dy = gamma * invstd * (dz - dbeta / cnt - xhat * dgamma / cnt)
    return dy, dgamma, dbeta


# --------------------------------------------------------------------------- max pool
def maxpool2_fwd(x):
    """unet_parts.py:34 -- MaxPool2d(2): k2 s2, floor mode (odd last row/col dropped)."""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :2 * ho, :2 * wo, :].reshape(n, ho, 2, wo, 2, c)
    return v.max(axis=(2, 4))


def maxpool2_bwd(x, dy):
    """Route dy to the first maximal element of each 2x2 window (ATen tie-break: row-major first)."""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :2 * ho, :2 * wo, :].reshape(n, ho, 2, wo, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, ho, wo, c, 4)
    arg = v.argmax(axis=-1)
    d = np.zeros(v.shape, dtype=dy.dtype)
    np.put_along_axis(d, arg[..., None], dy[..., None], axis=-1)
    d = d.reshape(n, ho, wo, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, 2 * ho, 2 * wo, c)
    dx = np.zeros(x.shape, dtype=dy.dtype)
    dx[:, :2 * ho, :2 * wo, :] = d
    return dx


# --------------------------------------------------------------------------- up-sampling paths
def convT2x2_fwd(x, w, b):
    """unet_parts.py:53 -- ConvTranspose2d(k=2, s=2) with bias.  x[N,h,w,Ci], w[Ci,Co,2,2]."""
    n, h, wd, ci = x.shape
    co = w.shape[1]
    y = np.empty((n, 2 * h, 2 * wd, co), dtype=np.result_type(x, w))
    for a in range(2):
        for bb in range(2):
            y[:, a::2, bb::2, :] = x @ w[:, :, a, bb] + b
    return y


def convT2x2_bwd(x, w, dy):
    dx = np.zeros(x.shape, dtype=np.result_type(dy, w))
    dw = np.zeros_like(w, dtype=np.result_type(dy, x))
    ci = x.shape[-1]
    for a in range(2):
        for bb in range(2):
            g = dy[:, a::2, bb::2, :]
            dx += g @ w[:, :, a, bb].T
            dw[:, :, a, bb] = x.reshape(-1, ci).T @ g.reshape(-1, g.shape[-1])
    db = dy.reshape(-1, dy.shape[-1]).sum(axis=0)
    return dx, dw, db


def _ac_taps(n_in, n_out):
    """align_corners=True source taps: src = dst*(in-1)/(out-1) (unet_parts.py:50)."""
    if n_out == 1:
        src = np.zeros(1)
    else:
        src = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1))
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    frac = src - i0
    return i0, i1, frac


def bilinear2x_ac_fwd(x):
    """unet_parts.py:50 -- Upsample(scale_factor=2, bilinear, align_corners=True)."""
    n, h, w, c = x.shape
    r0, r1, fr = _ac_taps(h, 2 * h)
    c0, c1, fc = _ac_taps(w, 2 * w)
    fr = fr.astype(x.dtype)[None, :, None, None]
    fc = fc.astype(x.dtype)[None, None, :, None]
    top = x[:, r0][:, :, c0] * (1 - fc) + x[:, r0][:, :, c1] * fc
    bot = x[:, r1][:, :, c0] * (1 - fc) + x[:, r1][:, :, c1] * fc
    return top * (1 - fr) + bot * fr


def bilinear2x_ac_bwd(x_shape, dy):
    n, h, w, c = x_shape
    r0, r1, fr = _ac_taps(h, 2 * h)
    c0, c1, fc = _ac_taps(w, 2 * w)
    dx = np.zeros(x_shape, dtype=dy.dtype)
    fr = fr.astype(dy.dtype)
    fc = fc.astype(dy.dtype)
    for (ri, wr) in ((r0, 1 - fr), (r1, fr)):
        for (cj, wc) in ((c0, 1 - fc), (c1, fc)):
            contrib = dy * wr[None, :, None, None] * wc[None, None, :, None]
            np.add.at(dx, (slice(None), ri[:, None], cj[None, :]), contrib)
    return dx


def pad_cat(skip, up):
    """unet_parts.py:59-67 -- F.pad(up, [dX//2, dX-dX//2, dY//2, dY-dY//2]) then cat([skip, up], C)."""
    dy = skip.shape[1] - up.shape[1]
    dx = skip.shape[2] - up.shape[2]
    upp = np.pad(up, ((0, 0), (dy // 2, dy - dy // 2), (dx // 2, dx - dx // 2), (0, 0)))
    return np.concatenate([skip, upp], axis=-1)


def pad_cat_bwd(skip_shape, up_shape, dcat):
    cs = skip_shape[-1]
    dy = skip_shape[1] - up_shape[1]
    dx = skip_shape[2] - up_shape[2]
    t, l = dy // 2, dx // 2
    return dcat[..., :cs], dcat[:, t:t + up_shape[1], l:l + up_shape[2], cs:]


# --------------------------------------------------------------------------- 1x1 head
def conv1x1_fwd(x, w, b):
    """unet_parts.py:74 -- Conv2d(k=1) with bias.  w[Co,Ci,1,1]."""
    return x @ w[:, :, 0, 0].T + b


def conv1x1_bwd(x, w, dy):
    dx = dy @ w[:, :, 0, 0]
    dw = (dy.reshape(-1, dy.shape[-1]).T @ x.reshape(-1, x.shape[-1]))[:, :, None, None]
    db = dy.reshape(-1, dy.shape[-1]).sum(axis=0)
    return dx, dw, db


# --------------------------------------------------------------------------- dice + loss
def dice_coeff(inp, tgt, reduce_batch_first=False, epsilon=1e-6):
    """utils/dice_score.py:5-17."""
    assert inp.shape == tgt.shape
    assert inp.ndim == 3 or not reduce_batch_first
    axes = (-1, -2) if inp.ndim == 2 or not reduce_batch_first else (-1, -2, -3)
    inter = 2 * (inp * tgt).sum(axis=axes)
    sets = inp.sum(axis=axes) + tgt.sum(axis=axes)
    sets = np.where(sets == 0, inter, sets)
    return np.mean((inter + epsilon) / (sets + epsilon))


def multiclass_dice_coeff(inp, tgt, reduce_batch_first=False, epsilon=1e-6):
    """utils/dice_score.py:20-22 (flatten(0,1) of [N,C,H,W])."""
    return dice_coeff(inp.reshape((-1,) + inp.shape[2:]), tgt.reshape((-1,) + tgt.shape[2:]),
                      reduce_batch_first, epsilon)


def seg_loss_fwd_bwd(logits, mask, epsilon=1e-6):
    """train.py:102-111.  logits[N,H,W,C] (NHWC), mask[N,H,W] int.  Returns loss, dlogits.

    C>1: mean CE + (1 - global dice over softmax probs vs one-hot)   (dice_score.py:25-28,
    reduce_batch_first=True on the flattened [N*C,H,W] tensor => ONE global ratio).
    C==1: mean BCE-with-logits + (1 - global dice over sigmoid).
    """
    n, h, w, c = logits.shape
    z = logits.astype(np.float64)
    cnt = n * h * w
    if c == 1:
        z = z[..., 0]
        t = mask.astype(np.float64)
        p = 1.0 / (1.0 + np.exp(-z))
        bce = np.mean(np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z))))
        inter, sets = 2 * (p * t).sum(), p.sum() + t.sum()
        if sets == 0:
            sets_eff, dsets = inter, 0.0
        else:
            sets_eff, dsets = sets, 1.0
        dice = (inter + epsilon) / (sets_eff + epsilon)
        # d(dice)/dp = (2t*(S+eps) - (I+eps)*dS/dp) / (S+eps)^2 with S = sets_eff
        if sets == 0:
            ddice_dp = np.zeros_like(p)            # dice == 1 identically around inter==sets==0
        else:
            ddice_dp = (2 * t * (sets_eff + epsilon) - (inter + epsilon) * dsets) / (sets_eff + epsilon) ** 2
        dz = (p - t) / cnt - ddice_dp * p * (1 - p)
        return bce + 1 - dice, dz[..., None]
    zmax = z.max(axis=-1, keepdims=True)
    e = np.exp(z - zmax)
    p = e / e.sum(axis=-1, keepdims=True)
    onehot = np.eye(c)[mask]
    ce = -np.mean(np.log(np.take_along_axis(p, mask[..., None], axis=-1)[..., 0]))
    inter, sets = 2 * (p * onehot).sum(), p.sum() + onehot.sum()
    dice = (inter + epsilon) / (sets + epsilon)       # sets >= cnt > 0 always
    g = -(2 * onehot * (sets + epsilon) - (inter + epsilon)) / (sets + epsilon) ** 2   # d(1-dice)/dp
    dz = (p - onehot) / cnt + p * (g - (p * g).sum(axis=-1, keepdims=True))
    return ce + 1 - dice, dz
