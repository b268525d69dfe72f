"""Parameter containers mirroring the reference layer classes (kandinsky/models/nn.py).

These modules own NO arithmetic: they exist so that `state_dict()` / `load_state_dict(assign=True)`
see exactly the checkpoint layout of the reference and so that code poking at
attribute names (`visual_transformer_blocks[i].self_attention.num_heads`, ...) keeps working.  All math
runs inside libk5.so (see dit.py); calling a container directly raises.
"""
import math

import torch
from torch import nn


class _NoMath(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the computation runs in the HIP engine "
            "(DiffusionTransformer3D.forward). There is no eager fallback.")


class Linear(_NoMath):
    """nn.Linear-shaped parameter holder (weight [out,in], bias [out])."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features), requires_grad=False) if bias else None


class NormWeight(_NoMath):
    def __init__(self, dim, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(dim), requires_grad=False)


class TimeEmbeddings(_NoMath):  # reference nn.py:43-61
    def __init__(self, model_dim, time_dim, max_period=10000.0):
        super().__init__()
        assert model_dim % 2 == 0
        self.model_dim, self.max_period = model_dim, max_period
        self.in_layer = Linear(model_dim, time_dim)
        self.out_layer = Linear(time_dim, time_dim)


class TextEmbeddings(_NoMath):  # reference nn.py:64-72
    def __init__(self, text_dim, model_dim):
        super().__init__()
        self.in_layer = Linear(text_dim, model_dim)
        self.norm = NormWeight(model_dim, bias=True)


class VisualEmbeddings(_NoMath):  # reference nn.py:75-96
    def __init__(self, visual_dim, model_dim, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.in_layer = Linear(math.prod(patch_size) * visual_dim, model_dim)


class Modulation(_NoMath):  # reference nn.py:153-164
    def __init__(self, time_dim, model_dim, num_params):
        super().__init__()
        self.out_layer = Linear(time_dim, num_params * model_dim)
        self.out_layer.weight.data.zero_()
        self.out_layer.bias.data.zero_()


class MultiheadAttention(_NoMath):  # reference nn.py:166-349 (Enc / Dec / Cross share the layout)
    def __init__(self, num_channels, head_dim):
        super().__init__()
        assert num_channels % head_dim == 0
        self.num_heads = num_channels // head_dim
        self.to_query = Linear(num_channels, num_channels)
        self.to_key = Linear(num_channels, num_channels)
        self.to_value = Linear(num_channels, num_channels)
        self.query_norm = NormWeight(head_dim)
        self.key_norm = NormWeight(head_dim)
        self.out_layer = Linear(num_channels, num_channels)


MultiheadSelfAttentionEnc = MultiheadSelfAttentionDec = MultiheadCrossAttention = MultiheadAttention


class FeedForward(_NoMath):  # reference nn.py:352-361
    def __init__(self, dim, ff_dim):
        super().__init__()
        self.in_layer = Linear(dim, ff_dim, bias=False)
        self.out_layer = Linear(ff_dim, dim, bias=False)


class OutLayer(_NoMath):  # reference nn.py:364-400
    def __init__(self, model_dim, time_dim, visual_dim, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.modulation = Modulation(time_dim, model_dim, 2)
        self.out_layer = Linear(model_dim, math.prod(patch_size) * visual_dim)
