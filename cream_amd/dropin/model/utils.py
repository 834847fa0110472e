"""`model.utils` symbols used by the reference's supernet_transformer.py / embedding_super.py
(AutoFormer/model/utils.py): trunc_normal_, DropPath, to_2tuple."""
import collections.abc
from itertools import repeat

import torch.nn as nn

from cream_amd.autoformer.supernet import DropPath  # noqa: F401


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _ntuple(n):
    def parse(x):
        return x if isinstance(x, collections.abc.Iterable) else tuple(repeat(x, n))
    return parse


to_1tuple, to_2tuple, to_3tuple, to_4tuple = _ntuple(1), _ntuple(2), _ntuple(3), _ntuple(4)
to_ntuple = _ntuple
