"""The searched architectures shipped with the reference (train/fasterseg/arch_0.pt = teacher, arch_1.pt = student),
stored as .npz next to this file, plus the recipe train/train.py:90-107 uses to turn them into networks."""
import os

import numpy as np
import torch

from .model_seg import Network_Multi_Path_Infer

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fasterseg")
WIDTH_MULT_LIST = [4. / 12, 6. / 12, 8. / 12, 10. / 12, 1.]        # config_train.py width_mult_list
STEM_HEAD_WIDTH = [(1., 1.), (8. / 12, 8. / 12)]                     # config_train.py stem_head_width (teacher, student)


def objective_acc_lat(acc, lat, lat_target=8.3, alpha=-0.07, beta=-0.07):
    """Accuracy/latency objective that picks the branch pair (tools/utils/darts_utils.py:343-348)."""
    w = alpha if lat <= lat_target else beta
    return acc * np.power(lat / lat_target, w)


def load_arch(idx):
    """dict with alpha_i_{0,1,2}, beta_i_{1,2}, ratio_i_{0,1,2} tensors and mIoU02/12, latency02/12 floats."""
    raw = np.load(os.path.join(_DIR, "arch_%d.npz" % idx))
    return {k: (torch.tensor(raw[k]) if raw[k].ndim else float(raw[k])) for k in raw.files}


def build_derived(idx, training=False, lasts=None, num_classes=19, layers=16, Fch=12):
    """Network_Multi_Path_Infer for arch `idx` (0 teacher, 1 student) as train/train.py:92-107 builds it: the last
    layers are chosen by objective_acc_lat unless `lasts` is given; teacher ignores 'skip' (ignore_skip=True)."""
    a = load_arch(idx)
    net = Network_Multi_Path_Infer(
        [a["alpha_%d_0" % idx], a["alpha_%d_1" % idx], a["alpha_%d_2" % idx]],
        [None, a["beta_%d_1" % idx], a["beta_%d_2" % idx]],
        [a["ratio_%d_0" % idx], a["ratio_%d_1" % idx], a["ratio_%d_2" % idx]],
        num_classes=num_classes, layers=layers, Fch=Fch, width_mult_list=WIDTH_MULT_LIST,
        stem_head_width=STEM_HEAD_WIDTH[idx], ignore_skip=(idx == 0))
    if lasts is None:
        o02 = objective_acc_lat(a["mIoU02"], a["latency02"])
        o12 = objective_acc_lat(a["mIoU12"], a["latency12"])
        lasts = [2, 0] if o02 > o12 else [2, 1]
    net.train(training)
    net.build_structure(list(lasts))
    net.train(training)
    return net


def init_weight(module, seed=12345, bn_eps=1e-5, bn_momentum=0.1):
    """kaiming_normal_(fan_in, relu) on every conv, BN gamma=1 / beta=0 / eps / momentum — what train/train.py:122 does
    through tools/utils/init_func.py:5-29 (random init: trained weights are not available offline)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d):
            fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.zero_()
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.eps = bn_eps
            m.momentum = bn_momentum
            if m.weight is not None:
                torch.nn.init.constant_(m.weight, 1)
                torch.nn.init.constant_(m.bias, 0)
    return module
