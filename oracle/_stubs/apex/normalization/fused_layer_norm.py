"""Stub of apex FusedLayerNorm for running the reference on CPU: the real module's CPU path is
literally F.layer_norm (apex/apex/normalization/fused_layer_norm.py:153-156)."""
import torch


class FusedLayerNorm(torch.nn.LayerNorm):
    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__(normalized_shape, eps=eps, elementwise_affine=elementwise_affine)
