"""UNETR on the MI355X kernels: drop-in mirror of
``adell_mri.modules.segmentation.unetr.UNETR`` (reference:
adell_mri/modules/segmentation/unetr.py:21-432).

Same constructor signature (unetr.py:30-66), attributes, module tree /
``state_dict`` keys (``vit``, ``first_encoder``, ``reconstruction_ops``,
``bottleneck_reconstruction``, ``rescalers``, ``upscale_ops``, ``link_ops``,
``decoding_operations``, ``final_layer``) and ``forward`` returns
(unetr.py:332-432). ViT encoder: token kernels + Linear-as-1x1x1-conv; the
reconstruction / decoder path reuses the U-Net conv, transposed-conv and fused
ADN kernels (virtual concat, stats in the conv epilogue).
"""
from typing import Callable, Dict, List

import numpy as np
import torch

from ..._lib import AdellHipError
from ..layers.adn_fn import get_adn_fn
from ..layers.conv import ConvTranspose3d
from ..layers.vit import LinearEmbedding, ViT
from .unet import UNet


class UNETR(UNet, torch.nn.Module):
    def __init__(
        self,
        # linear embedding and transformer
        image_size,
        patch_size,
        number_of_blocks: int,
        return_at: List[int],
        attention_dim: int = None,
        hidden_dim: int = None,
        embedding_size: int = None,
        n_heads: int = 4,
        dropout_rate: float = 0.0,
        embed_method: str = "linear",
        mlp_structure: List[int] = [256, 256],
        adn_fn_mlp: Callable = get_adn_fn(1, "identity", "gelu"),
        # regular u-net parametrization
        spatial_dimensions: int = 2,
        conv_type: str = "regular",
        link_type: str = "identity",
        upscale_type: str = "upsample",
        interpolation: str = "bilinear",
        norm_type: str = "batch",
        dropout_type: str = "dropout",
        padding: int = 0,
        dropout_param: float = 0.0,
        activation_fn: torch.nn.Module = torch.nn.PReLU,
        in_channels: int = 1,
        n_classes: int = 2,
        depth: list = [16, 32, 64],
        kernel_sizes: list = [3, 3, 3],
        bottleneck_classification: bool = False,
        skip_conditioning: int = None,
        feature_conditioning: int = None,
        feature_conditioning_params: Dict[str, torch.Tensor] = None,
        deep_supervision: bool = False,
        encoder_only: bool = False,
    ):
        super().__init__(parent_class=True)
        self.image_size = image_size
        self.patch_size = patch_size
        self.number_of_blocks = number_of_blocks
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.embedding_size = embedding_size
        self.return_at = return_at
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate
        self.embed_method = embed_method
        self.mlp_structure = mlp_structure
        self.adn_fn_mlp = adn_fn_mlp
        self.spatial_dimensions = spatial_dimensions
        self.conv_type = conv_type
        self.link_type = link_type
        self.upscale_type = upscale_type
        self.interpolation = interpolation
        self.norm_type = norm_type
        self.dropout_type = dropout_type
        self.padding = padding
        self.dropout_param = dropout_param
        self.activation_fn = activation_fn
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.depth = depth
        self.kernel_sizes = kernel_sizes
        self.bottleneck_classification = bottleneck_classification
        self.skip_conditioning = skip_conditioning
        self.feature_conditioning = feature_conditioning
        self.feature_conditioning_params = feature_conditioning_params
        self.deep_supervision = deep_supervision
        self.encoder_only = encoder_only

        self.strides = [2 for _ in self.depth]
        self.scale = int(2 ** len(self.return_at))
        self.in_channels_rec = int(np.prod([self.scale ** self.spatial_dimensions,
                                            self.in_channels]))
        self.assertions()
        if self.spatial_dimensions != 3:
            raise NotImplementedError("HIP UNETR is 3-D (the BASELINE configuration)")
        if self.feature_conditioning is not None:
            raise NotImplementedError("feature conditioning is outside the HIP path built so far")

        self.get_norm_op()
        self.get_drop_op()
        self.get_conv_op()
        self.init_vit()
        if self.encoder_only is False:
            self.init_first_encoder()
            self.init_reconstruction_ops()
            self.init_upscale_ops()
            self.init_link_ops()
            self.init_decoder()
            self.init_final_layer()
            self.init_rescalers()
            if self.bottleneck_classification is True:
                self.init_bottleneck_classifier()

    def assertions(self):
        assert (len(self.depth) - 1) == len(self.return_at), \
            "(len(depth)-1) must be the same as len(return_at)"
        assert max(self.return_at) <= self.number_of_blocks, \
            "len(depth) must be smaller than number_of_blocks"
        assert len(self.depth) == len(self.kernel_sizes), \
            "len(depth) must be the same as len(kernel_sizes)"

    def init_vit(self):
        self.vit = ViT(image_size=self.image_size, patch_size=self.patch_size,
                       in_channels=self.in_channels, number_of_blocks=self.number_of_blocks,
                       attention_dim=self.attention_dim, hidden_dim=self.hidden_dim,
                       embedding_size=self.embedding_size, n_heads=self.n_heads,
                       dropout_rate=self.dropout_rate, embed_method=self.embed_method,
                       mlp_structure=self.mlp_structure, adn_fn=self.adn_fn_mlp)
        self.rearrange_rescale = self.vit.embedding.rearrange_rescale

    def init_rescalers(self):
        self.rescalers = torch.nn.ModuleList([
            LinearEmbedding(image_size=self.image_size, patch_size=self.patch_size,
                            in_channels=self.in_channels, out_dim=self.embedding_size,
                            dropout_rate=0.0, embed_method="linear", use_pos_embed=False,
                            use_class_token=False)
            for _ in self.depth[1:]])

    def init_first_encoder(self):
        self.first_encoder = torch.nn.Sequential(
            self.adn_fn(self.in_channels),
            self.conv_op_enc(self.in_channels, self.depth[0], 3, padding="same"),
            self.adn_fn(self.depth[0]))

    def unetr_transp_op(self, in_d: int, out_d: int, kernel_size: int = 3) -> torch.nn.Module:
        return torch.nn.Sequential(
            ConvTranspose3d(in_d, out_d, 2, 2), self.adn_fn(out_d),
            self.conv_op_enc(out_d, out_d, kernel_size, padding="same"), self.adn_fn(out_d))

    def unetr_transp_block(self, in_d: int, out_d: int, n_ops: int,
                           kernel_size: int = 3) -> torch.nn.Module:
        out = [self.unetr_transp_op(in_d, out_d, kernel_size)]
        for _ in range(n_ops):
            out.append(self.unetr_transp_op(out_d, out_d, 3))
        return torch.nn.Sequential(*out)

    def init_reconstruction_ops(self):
        self.reconstructed_dim = [self.in_channels_rec,
                                  *[x // self.scale for x in self.image_size]]
        self.reconstruction_ops = torch.nn.ModuleList([])
        self.n_skip_connections = len(self.depth) - 1
        for i, d in enumerate(self.depth[1:-1]):
            i = i + 1
            n_ops = self.n_skip_connections - i
            self.reconstruction_ops.append(
                self.unetr_transp_block(self.in_channels_rec, d, n_ops - 1, 3))
        self.bottleneck_reconstruction = self.conv_op_enc(self.in_channels_rec, self.depth[-1],
                                                          1, 1)

    def forward(self, X: torch.Tensor, X_skip_layer: torch.Tensor = None,
                X_feature_conditioning: torch.Tensor = None, return_features=False,
                return_bottleneck=False, return_logits=False):
        if not X.is_cuda:
            raise AdellHipError("adell_mri_amd.UNETR runs on MI355X only (no CPU fallback)")
        if X_feature_conditioning is not None:
            raise NotImplementedError("feature conditioning is outside the HIP path built so far")
        if X_skip_layer is not None and len(X_skip_layer.shape) < len(X.shape):
            X_skip_layer = X_skip_layer.unsqueeze(1)

        curr, encoding_out = self.vit(X, return_at=self.return_at)
        X_encoded_first = self.first_encoder(X)
        curr = self.rearrange_rescale(curr, self.scale)
        encoding_out = [rescaler.rearrange_rescale(x, self.scale)
                        for x, rescaler in zip(encoding_out, self.rescalers)]
        curr = self.bottleneck_reconstruction(curr)
        encoding_out = [X_encoded_first,
                        *[rec_op(x) for x, rec_op in zip(encoding_out, self.reconstruction_ops)]]
        encoding_out.append(curr)
        bottleneck = curr
        if return_bottleneck is True:
            return None, None, bottleneck
        elif self.encoder_only is True:
            return bottleneck

        deep_outputs = []
        for i in range(len(self.decoding_operations)):
            op = self.decoding_operations[i]
            link_in = encoding_out[-i - 2]
            if X_skip_layer is not None:
                xfl = torch.nn.functional.interpolate(X_skip_layer, link_in.shape[2:],
                                                      mode="nearest")
                link_in = torch.cat([link_in, xfl], axis=1)
            encoded = self.link_ops[i](link_in)
            curr = self.upscale_ops[i](curr)
            curr = op(curr, X_cat=encoded)
            deep_outputs.append(curr)

        final_features = curr
        curr = self._final(self.final_layer, curr, return_logits)
        if return_features is True:
            return curr, final_features, bottleneck
        if self.bottleneck_classification is True:
            pooled = bottleneck.flatten(start_dim=2).max(-1).values
            bn_out = self.bottleneck_classifier(pooled)
        else:
            bn_out = None
        if self.deep_supervision is True:
            for i in range(len(deep_outputs)):
                deep_outputs[i] = self._final(self.deep_supervision_ops[i], deep_outputs[i], False)
            return curr, bn_out, deep_outputs
        return curr, bn_out
