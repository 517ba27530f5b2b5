"""CPU restatement of the NuTufts/ubresnet segmentation path (plain PyTorch-CPU fp32).

TEST INFRASTRUCTURE ONLY: this file is the *oracle* the HIP path is checked
against.  The product package (``ubresnet_amd``) never imports it.

Pinning: ``tests/golden/make_golden.py`` imports the reference's own model code
in memory (SURVEY.md section 8c recipe) in the build container and stores
its outputs as fixtures in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those fixtures (forward <=1e-6, gradients <=1e-5).

Everything is functional: parameters and buffers come in as a ``state_dict``-style
mapping with exactly the reference's keys, activations are NCHW
fp32, the residual add is out-of-place (the reference's in-place ``out += x``,
models/common_layers.py:52,54, is value-identical in forward and is the
documented torch>=1.x autograd defect in backward).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5      # nn.BatchNorm2d default (models/common_layers.py:24)
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------
def _bn(sd: Mapping[str, Tensor], pre: str, x: Tensor, train: bool, new_stats: Optional[dict]) -> Tensor:
    """nn.BatchNorm2d (e.g. models/common_layers.py:24,27,35; models/ub_uresnet.py:42,61).

    train: biased batch variance for normalisation, running stats updated with
    the unbiased variance and momentum 0.1.  eval: running stats.
    Running-stat updates are returned through ``new_stats`` (never in place).
    """
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    rm, rv = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    if train:
        if new_stats is not None:
            with torch.no_grad():
                n = x.numel() // x.shape[1]
                mean = x.mean(dim=(0, 2, 3))
                var_u = x.var(dim=(0, 2, 3), unbiased=True) if n > 1 else torch.zeros_like(mean)
                new_stats[pre + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                new_stats[pre + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u
                nbt = sd.get(pre + ".num_batches_tracked")
                if nbt is not None:
                    new_stats[pre + ".num_batches_tracked"] = nbt + 1
        return F.batch_norm(x, None, None, w, b, True, BN_MOMENTUM, BN_EPS)
    return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def _conv(sd, pre, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[pre + ".weight"], sd.get(pre + ".bias"), stride, padding, dilation)


# --------------------------------------------------------------------------
# models/common_layers.py
# --------------------------------------------------------------------------
def basic_block(sd, pre, x, stride, train, ns):
    """BasicBlock.forward, models/common_layers.py:39-58 (ctor :21-37).

    relu( relu2(bn2(conv2(relu1(bn1(conv1(x)))))) + shortcut ),
    shortcut = bnpass(bypass(x)) iff the block has a bypass (inplanes!=planes or stride>1, :33-35).
    """
    out = F.relu(_bn(sd, pre + ".bn1", _conv(sd, pre + ".conv1", x, stride, 1), train, ns))
    out = F.relu(_bn(sd, pre + ".bn2", _conv(sd, pre + ".conv2", out, 1, 1), train, ns))
    if (pre + ".bypass.weight") in sd:
        sc = _bn(sd, pre + ".bnpass", _conv(sd, pre + ".bypass", x, stride, 0), train, ns)
    else:
        sc = x
    return F.relu(out + sc)


def double_resnet(sd, pre, x, stride, train, ns):
    """DoubleResNet.forward, models/common_layers.py:117-120."""
    return basic_block(sd, pre + ".res2", basic_block(sd, pre + ".res1", x, stride, train, ns), 1, train, ns)


def conv_transpose_layer(sd, pre, x, skip, train, ns):
    """ConvTransposeLayer.forward, models/common_layers.py:127-132.

    ConvTranspose2d(k4,s2,p1,bias=False) with output_size=skip.size() -> cat([up, skip],1) -> DoubleResNet.
    """
    w = sd[pre + ".deconv.weight"]
    oh, ow = skip.shape[2], skip.shape[3]
    # output_size handling of nn.ConvTranspose2d: output_padding = requested - minimal size
    min_h = (x.shape[2] - 1) * 2 - 2 + 4
    min_w = (x.shape[3] - 1) * 2 - 2 + 4
    oph, opw = oh - min_h, ow - min_w
    if not (0 <= oph < 2 and 0 <= opw < 2):
        raise ValueError("requested output size (%d,%d) not reachable from input (%d,%d)" % (oh, ow, x.shape[2], x.shape[3]))
    up = F.conv_transpose2d(x, w, None, 2, 1, (oph, opw))
    return double_resnet(sd, pre + ".res", torch.cat([up, skip], 1), 1, train, ns)


# --------------------------------------------------------------------------
# models/ub_uresnet.py
# --------------------------------------------------------------------------
def uresnet_forward(sd: Mapping[str, Tensor], x: Tensor, train: bool = True,
                    new_stats: Optional[dict] = None, return_logits: bool = False) -> Tensor:
    """UResNet.forward, models/ub_uresnet.py:88-147.  Returns log-softmax [B,C,H,W]."""
    if x.shape[2] % 32 or x.shape[3] % 32:
        raise ValueError("H and W must be multiples of 32 (ConvTranspose output_size contract)")
    ns = new_stats
    x = _conv(sd, "conv1", x, 1, 3)                               # :94
    x0 = F.relu(_bn(sd, "bn1", x, train, ns))                     # :95-96
    x = F.max_pool2d(x0, 3, 2, 1)                                 # :97
    x1 = double_resnet(sd, "enc_layer1", x, 1, train, ns)         # :103
    x2 = double_resnet(sd, "enc_layer2", x1, 2, train, ns)
    x3 = double_resnet(sd, "enc_layer3", x2, 2, train, ns)
    x4 = double_resnet(sd, "enc_layer4", x3, 2, train, ns)
    x5 = double_resnet(sd, "enc_layer5", x4, 2, train, ns)        # :107
    x = conv_transpose_layer(sd, "dec_layer5", x5, x4, train, ns)  # :116
    x = conv_transpose_layer(sd, "dec_layer4", x, x3, train, ns)
    x = conv_transpose_layer(sd, "dec_layer3", x, x2, train, ns)
    x = conv_transpose_layer(sd, "dec_layer2", x, x1, train, ns)
    x = conv_transpose_layer(sd, "dec_layer1", x, x0, train, ns)  # :132
    x = F.relu(_bn(sd, "bn10", _conv(sd, "conv10", x, 1, 3), train, ns))  # :137-139
    x = _conv(sd, "conv11", x, 1, 3)                              # :141
    if return_logits:
        return x
    return F.log_softmax(x, dim=1)                                # :143


# --------------------------------------------------------------------------
# models/ASPP_ResNet.py
# --------------------------------------------------------------------------
def aspp(sd, pre, x, train, ns):
    """ASPP.forward, models/ASPP_ResNet.py:227-263: 1x1, 3x3 d1, 3x3 d3, 3x3 d5 (each conv+bias->BN->ReLU, 16 ch) + MaxPool(3,1,1)."""
    b1 = F.relu(_bn(sd, pre + ".B1_bn", _conv(sd, pre + ".B1_conv", x, 1, 0, 1), train, ns))
    b2 = F.relu(_bn(sd, pre + ".B2_bn", _conv(sd, pre + ".B2_conv", x, 1, 1, 1), train, ns))
    b3 = F.relu(_bn(sd, pre + ".B3_bn", _conv(sd, pre + ".B3_conv", x, 1, 3, 3), train, ns))
    b4 = F.relu(_bn(sd, pre + ".B4_bn", _conv(sd, pre + ".B4_conv", x, 1, 5, 5), train, ns))
    b5 = F.max_pool2d(x, 3, 1, 1)
    return torch.cat((b1, b2, b3, b4, b5), 1)


def aspp_post(sd, pre, x, train, ns):
    """ASPP_post.forward, models/ASPP_ResNet.py:280-286."""
    return F.relu(_bn(sd, pre + ".ASPP_bn", _conv(sd, pre + ".ASPP_conv", x, 1, 0), train, ns))


def aspp_resnet_forward(sd, x, train=True, new_stats=None, return_logits=False):
    """ASPP_ResNet.forward, models/ASPP_ResNet.py:416-523."""
    if x.shape[2] % 32 or x.shape[3] % 32:
        raise ValueError("H and W must be multiples of 32")
    ns = new_stats
    x = _conv(sd, "conv1", x, 1, 3)
    x0 = F.relu(_bn(sd, "bn1", x, train, ns))
    x = F.max_pool2d(x0, 3, 2, 1)
    e1 = double_resnet(sd, "enc_layer1", x, 1, train, ns)
    e2 = double_resnet(sd, "enc_layer2", e1, 2, train, ns)
    e3 = double_resnet(sd, "enc_layer3", e2, 2, train, ns)
    e4 = double_resnet(sd, "enc_layer4", e3, 2, train, ns)
    e5 = double_resnet(sd, "enc_layer5", e4, 2, train, ns)
    skips = {}
    for lvl, e in ((3, e3), (4, e4), (5, e5)):                    # :447-483
        a = aspp(sd, "ASPP_layer_enc%d" % lvl, e, train, ns)
        a = aspp_post(sd, "ASPP_combine_enc%d" % lvl, a, train, ns)
        skips[lvl] = torch.cat((a, e), 1)
    d5 = conv_transpose_layer(sd, "dec_layer5", skips[5], skips[4], train, ns)
    d4 = conv_transpose_layer(sd, "dec_layer4", d5, skips[3], train, ns)
    d3 = conv_transpose_layer(sd, "dec_layer3", d4, e2, train, ns)
    d2 = conv_transpose_layer(sd, "dec_layer2", d3, e1, train, ns)
    d1 = conv_transpose_layer(sd, "dec_layer1", d2, x0, train, ns)
    x = F.relu(_bn(sd, "bn10", _conv(sd, "conv10", d1, 1, 3), train, ns))
    x = _conv(sd, "conv11", x, 1, 3)
    if return_logits:
        return x
    return F.log_softmax(x, dim=1)


# --------------------------------------------------------------------------
# training/pixelwise_nllloss.py
# --------------------------------------------------------------------------
def pixelwise_nll(predict: Tensor, target: Tensor, pixelweights: Tensor,
                  weight: Optional[Tensor] = None, ignore_index: int = -100) -> Tensor:
    """PixelWiseNLLLoss.forward, training/pixelwise_nllloss.py:41-61.

    mean_{b,h,w}( -predict[b,target,h,w] * classw[target] * pixelweights[b,h,w] );
    ignore_index pixels contribute 0 but stay in the denominator (:51 reduce=False, :59 torch.mean).
    """
    pix = F.nll_loss(predict, target, weight, ignore_index=ignore_index, reduction="none")
    return torch.mean(pix * pixelweights)


# --------------------------------------------------------------------------
# training/train_ubresnet2018_wlarcv2.py
# --------------------------------------------------------------------------
def accuracy(output: Tensor, target: Tensor) -> List[float]:
    """accuracy(), training/train_ubresnet2018_wlarcv2.py:509-566: per-class recall (%) + total."""
    pred = output.max(1)[1]
    correct = pred.eq(target)
    res, tot_c, tot_n = [], 0.0, 0.0
    for c in range(output.shape[1]):
        cm = target.eq(c)
        n = float(cm.sum().item())
        k = float((correct & cm).sum().item())
        res.append(100.0 * k / n if n > 0 else 0.0)
        tot_c += k
        tot_n += n
    res.append(100.0 * tot_c / tot_n)
    return res


def confusion_matrix(output: Tensor, target: Tensor) -> Tensor:
    """C x C counts [true, pred] (build-defined metric harness; IoU derives from it)."""
    C = output.shape[1]
    pred = output.max(1)[1].reshape(-1)
    t = target.reshape(-1)
    ok = (t >= 0) & (t < C)
    idx = t[ok] * C + pred[ok]
    return torch.bincount(idx, minlength=C * C).reshape(C, C)


def iou_from_confusion(cm: Tensor) -> List[float]:
    cm = cm.double()
    out = []
    for c in range(cm.shape[0]):
        inter = cm[c, c]
        union = cm[c, :].sum() + cm[:, c].sum() - inter
        out.append(float(inter / union) if union > 0 else float("nan"))
    return out


# --------------------------------------------------------------------------
# state_dict schema + deterministic seeded parameters (fixtures are keyed on this)
# --------------------------------------------------------------------------
def _block_keys(pre, cin, cout, stride):
    ks = OrderedDict()
    ks[pre + ".conv1.weight"] = (cout, cin, 3, 3)
    _bn_keys(ks, pre + ".bn1", cout)
    ks[pre + ".conv2.weight"] = (cout, cout, 3, 3)
    _bn_keys(ks, pre + ".bn2", cout)
    if cin != cout or stride > 1:
        ks[pre + ".bypass.weight"] = (cout, cin, 1, 1)
        _bn_keys(ks, pre + ".bnpass", cout)
    return ks


def _bn_keys(ks, pre, c):
    ks[pre + ".weight"] = (c,)
    ks[pre + ".bias"] = (c,)
    ks[pre + ".running_mean"] = (c,)
    ks[pre + ".running_var"] = (c,)
    ks[pre + ".num_batches_tracked"] = ()


def _double_keys(pre, cin, cout, stride):
    ks = _block_keys(pre + ".res1", cin, cout, stride)
    ks.update(_block_keys(pre + ".res2", cout, cout, 1))
    return ks


def uresnet_schema(num_classes=3, input_channels=3, inplanes=16, final_conv_kernels=16) -> "OrderedDict[str, tuple]":
    """state_dict keys/shapes in PyTorch registration order for UResNet (models/ub_uresnet.py:31-70)."""
    ip = inplanes
    ks = OrderedDict()
    ks["conv1.weight"] = (ip, input_channels, 7, 7)
    ks["conv1.bias"] = (ip,)
    _bn_keys(ks, "bn1", ip)
    chans = [(ip, 2 * ip, 1), (2 * ip, 4 * ip, 2), (4 * ip, 8 * ip, 2), (8 * ip, 16 * ip, 2), (16 * ip, 32 * ip, 2)]
    for i, (ci, co, s) in enumerate(chans):
        ks.update(_double_keys("enc_layer%d" % (i + 1), ci, co, s))
    dec = [(5, 32 * ip, 16 * ip, 16 * ip), (4, 16 * ip, 8 * ip, 8 * ip), (3, 8 * ip, 4 * ip, 4 * ip),
           (2, 4 * ip, 2 * ip, 2 * ip), (1, 2 * ip, ip, ip)]
    for lvl, cin, cd, cr in dec:
        ks["dec_layer%d.deconv.weight" % lvl] = (cin, cd, 4, 4)
        ks.update(_double_keys("dec_layer%d.res" % lvl, cr + cd, cr, 1))
    ks["conv10.weight"] = (final_conv_kernels, ip, 7, 7)
    ks["conv10.bias"] = (final_conv_kernels,)
    _bn_keys(ks, "bn10", final_conv_kernels)
    ks["conv11.weight"] = (num_classes, final_conv_kernels, 7, 7)
    ks["conv11.bias"] = (num_classes,)
    return ks


def aspp_resnet_schema(num_classes=3, in_channels=3, inplanes=16) -> "OrderedDict[str, tuple]":
    """state_dict keys/shapes for ASPP_ResNet (models/ASPP_ResNet.py:291-402)."""
    ip = inplanes
    ks = OrderedDict()
    ks["conv1.weight"] = (ip, in_channels, 7, 7)
    ks["conv1.bias"] = (ip,)
    _bn_keys(ks, "bn1", ip)
    chans = [(ip, 2 * ip, 1), (2 * ip, 4 * ip, 2), (4 * ip, 8 * ip, 2), (8 * ip, 16 * ip, 2), (16 * ip, 32 * ip, 2)]
    for i, (ci, co, s) in enumerate(chans):
        ks.update(_double_keys("enc_layer%d" % (i + 1), ci, co, s))
    for lvl, c, cin_post in ((3, 8 * ip, 12 * ip), (4, 16 * ip, 20 * ip), (5, 32 * ip, 36 * ip)):
        p = "ASPP_layer_enc%d" % lvl
        for b, k in ((1, 1), (2, 3), (3, 3), (4, 3)):
            ks["%s.B%d_conv.weight" % (p, b)] = (16, c, k, k)
            ks["%s.B%d_conv.bias" % (p, b)] = (16,)
            _bn_keys(ks, "%s.B%d_bn" % (p, b), 16)
        q = "ASPP_combine_enc%d" % lvl
        ks[q + ".ASPP_conv.weight"] = (c, cin_post, 1, 1)
        ks[q + ".ASPP_conv.bias"] = (c,)
        _bn_keys(ks, q + ".ASPP_bn", c)
    # registration order in the reference: all three ASPP_layer/ASPP_combine pairs interleaved (:342-352) -- the loop above matches
    dec = [(5, 64 * ip, 16 * ip, 32 * ip), (4, 32 * ip, 8 * ip, 16 * ip), (3, 16 * ip, 4 * ip, 4 * ip),
           (2, 4 * ip, 2 * ip, 2 * ip), (1, 2 * ip, ip, ip)]
    for lvl, cin, cd, cr in dec:
        ks["dec_layer%d.deconv.weight" % lvl] = (cin, cd, 4, 4)
        ks.update(_double_keys("dec_layer%d.res" % lvl, cr + cd, cr, 1))
    ks["conv10.weight"] = (16, ip, 7, 7)
    ks["conv10.bias"] = (16,)
    _bn_keys(ks, "bn10", 16)
    ks["conv11.weight"] = (num_classes, ip, 7, 7)
    ks["conv11.bias"] = (num_classes,)
    return ks


def seeded_state_dict(schema: Mapping[str, tuple], seed: int = 42) -> "OrderedDict[str, Tensor]":
    """Deterministic parameters keyed by state_dict order (SURVEY.md section 8c fixture plan).

    numpy RandomState is bit-stable by NumPy policy, so both sides of a parity test
    regenerate the same fp32 values from (schema, seed) and weights are never committed.
    conv/deconv weights ~ N(0, sqrt(2/(kH*kW*shape[0]))) (the reference's init scale for Conv2d,
    models/ub_uresnet.py:73-79; for deconv the reference uses out_channels, we keep the same
    order of magnitude); biases ~ U(-0.1,0.1); BN gamma in [0.7,1.3], beta in [-0.2,0.2],
    running_mean in [-0.3,0.3], running_var in [0.5,1.5] (perturbed so BN folding is exercised).
    """
    import numpy as np
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for k, shp in schema.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0, dtype=torch.int64)
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rs.uniform(-0.3, 0.3, shp).astype("float32"))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(rs.uniform(0.5, 1.5, shp).astype("float32"))
        elif len(shp) == 4:
            std = math.sqrt(2.0 / (shp[2] * shp[3] * shp[0]))
            sd[k] = torch.from_numpy((rs.standard_normal(shp) * std).astype("float32"))
        elif ".bn" in k or k.startswith("bn") or "_bn." in k:
            if k.endswith(".weight"):
                sd[k] = torch.from_numpy(rs.uniform(0.7, 1.3, shp).astype("float32"))
            else:
                sd[k] = torch.from_numpy(rs.uniform(-0.2, 0.2, shp).astype("float32"))
        else:  # conv bias
            sd[k] = torch.from_numpy(rs.uniform(-0.1, 0.1, shp).astype("float32"))
    return sd


def is_param_key(k: str) -> bool:
    return not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))


def train_step_grads(forward_fn, sd: Mapping[str, Tensor], x: Tensor, target: Tensor, pixelweights: Tensor):
    """One forward + PixelWiseNLLLoss + backward on CPU.  Returns (loss, grads dict, logp, new running stats).

    Mirrors train() in training/train_ubresnet2018_wlarcv2.py:332-343 minus the optimizer.
    """
    p = OrderedDict()
    for k, v in sd.items():
        p[k] = v.clone().requires_grad_(True) if is_param_key(k) else v
    ns: Dict[str, Tensor] = {}
    logp = forward_fn(p, x, True, ns)
    loss = pixelwise_nll(logp, target, pixelweights)
    names = [k for k in p if is_param_key(k)]
    gs = torch.autograd.grad(loss, [p[k] for k in names])
    return loss.detach(), OrderedDict(zip(names, gs)), logp.detach(), ns


def state_dict_with_bn_stats(sd: Mapping[str, Tensor], keys, flat) -> "OrderedDict[str, Tensor]":
    """copy of `sd` whose BatchNorm running statistics are replaced by the fixture's (tests/golden/*norm*.npz store
    them as one flat float32 vector `bn_stats` in the order of `bn_keys`: they are part of the fixture's INPUT, produced
    by the reference model's own train-mode calibration passes, tests/golden/make_golden.py::calibrate_running_stats)"""
    import numpy as np
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    off = 0
    flat = np.asarray(flat, dtype=np.float32)
    for k in [str(k) for k in keys]:
        n = out[k].numel()
        out[k] = torch.from_numpy(flat[off:off + n].copy()).reshape(out[k].shape)
        off += n
    assert off == flat.shape[0]
    return out
