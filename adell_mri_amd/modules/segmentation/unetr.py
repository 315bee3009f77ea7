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
from ..layers.conv import ConvTranspose2d, ConvTranspose3d
from ... import functional as HF
from ..layers.vit import LinearEmbedding, SWINTransformerBlockStack, ViT
from .unet import UNet, _keep_arguments


class UNETR(UNet, torch.nn.Module):
    # signature: adell_mri/modules/segmentation/unetr.py:30-66 (ViT arguments, then the U-Net's)
    def __init__(self, image_size, patch_size, number_of_blocks: int, return_at: List[int],
                 attention_dim: int = None, hidden_dim: int = None, embedding_size: int = None,
                 n_heads: int = 4, dropout_rate: float = 0.0, embed_method: str = "linear",
                 mlp_structure: List[int] = [256, 256],
                 adn_fn_mlp: Callable = get_adn_fn(1, "identity", "gelu"),
                 spatial_dimensions: int = 2, conv_type: str = "regular",
                 link_type: str = "identity", upscale_type: str = "upsample",
                 interpolation: str = "bilinear", norm_type: str = "batch",
                 dropout_type: str = "dropout", padding: int = 0, dropout_param: float = 0.0,
                 activation_fn: torch.nn.Module = torch.nn.PReLU, in_channels: int = 1,
                 n_classes: int = 2, depth: list = [16, 32, 64], kernel_sizes: list = [3, 3, 3],
                 bottleneck_classification: bool = False, skip_conditioning: int = None,
                 feature_conditioning: int = None,
                 feature_conditioning_params: Dict[str, torch.Tensor] = None,
                 deep_supervision: bool = False, encoder_only: bool = False):
        arguments = dict(locals())
        super().__init__(parent_class=True)
        _keep_arguments(self, arguments)

        self.strides = [2 for _ in self.depth]
        self.scale = int(2 ** len(self.return_at))
        self.in_channels_rec = int(np.prod([self.scale ** self.spatial_dimensions,
                                            self.in_channels]))
        self.assertions()
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
            if self.feature_conditioning is not None:
                # get_segmentation_network passes feature_conditioning=len([]) == 0 and the
                # reference then still builds its (never used) Linear(0, d) gate stacks
                # (unetr.py:217-218): kept so that state_dict keys interchange
                self.init_feature_conditioning_operations()

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
        transp_conv = ConvTranspose2d if self.spatial_dimensions == 2 else ConvTranspose3d
        return torch.nn.Sequential(
            transp_conv(in_d, out_d, 2, 2), self.adn_fn(out_d),
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
        if X_feature_conditioning is not None:   # tiny [B, F] tensor: normalise the features
            X_feature_conditioning = (X_feature_conditioning - self.f_mean) / self.f_std
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

        return self._decode(encoding_out, bottleneck, X_skip_layer, X_feature_conditioning,
                            return_features, return_logits)   # the U-Net's decoder and head


class SWINUNet(UNet, torch.nn.Module):
    """SWIN-UNet: drop-in mirror of ``adell_mri.modules.segmentation.unetr.SWINUNet``
    (unetr.py:635-1033). Same constructor, module tree / ``state_dict`` keys
    (``first_encoder``, ``first_swin_block``, ``swin_blocks``, ``first_rec_op``,
    ``reconstruction_ops``, ``upscale_ops``, ``link_ops``, ``decoding_operations``,
    ``final_layer``) and ``forward`` returns. Encoder: stacks of shifted-window transformer
    blocks at image resolution, each closed by a space-to-depth rescale; channels-first
    LayerNorm + conv reconstruction; standard U-Net decoder; the head reads the virtual
    concatenation [first_encoder(X), decoder output]."""

    # signature: adell_mri/modules/segmentation/unetr.py:644-678
    def __init__(self, image_size, patch_size, window_size, shift_sizes,
                 embedding_size: int = None, n_heads: int = 4, dropout_rate: float = 0.0,
                 embed_method: str = "linear", mlp_structure: List[int] = [256, 256],
                 adn_fn_mlp: Callable = get_adn_fn(1, "identity", "gelu"),
                 spatial_dimensions: int = 2, conv_type: str = "regular",
                 link_type: str = "identity", upscale_type: str = "upsample",
                 interpolation: str = "bilinear", norm_type: str = "batch",
                 dropout_type: str = "dropout", padding: int = 0, dropout_param: float = 0.0,
                 activation_fn: torch.nn.Module = torch.nn.PReLU, in_channels: int = 1,
                 n_classes: int = 2, depth: list = [16, 32, 64], kernel_sizes: list = [3, 3, 3],
                 strides: list = None, bottleneck_classification: bool = False,
                 skip_conditioning: int = None, feature_conditioning: int = None,
                 feature_conditioning_params: Dict[str, torch.Tensor] = None,
                 deep_supervision: bool = False):
        arguments = dict(locals())
        super().__init__(parent_class=True)
        _keep_arguments(self, arguments)
        self.encoder_only = False
        self.number_of_blocks = len(self.depth)
        if self.spatial_dimensions not in (2, 3):
            raise NotImplementedError("SWINUNet: 2 or 3 spatial dimensions")
        self.arg_compliance()
        self.get_norm_op()
        self.get_drop_op()
        self.get_conv_op()
        self.init_first_encoder()
        self.init_swin_blocks()
        self.init_reconstruction_ops()
        self.init_upscale_ops()
        self.init_link_ops()
        self.init_decoder()
        self.init_final_layer()
        if self.bottleneck_classification is True:
            self.init_bottleneck_classifier()
        if self.feature_conditioning is not None:   # == 0 from the factory, unetr.py:818-819
            self.init_feature_conditioning_operations()

    def arg_compliance(self):
        msg = "shift_sizes must be list of ints or list of list of ints"
        assert len(self.depth) == self.number_of_blocks
        assert isinstance(self.shift_sizes, list), msg
        if isinstance(self.shift_sizes[0], int):
            self.shift_sizes = [self.shift_sizes for _ in self.depth]
        if isinstance(self.n_heads, int):
            self.n_heads = [self.n_heads for _ in self.depth]
        if isinstance(self.embedding_size, int) or self.embedding_size is None:
            self.embedding_size = [self.embedding_size for _ in self.depth]
        elif isinstance(self.shift_sizes[0], list):
            assert isinstance(self.shift_sizes[0][0], int), msg
        else:
            raise AssertionError(msg)
        if self.strides is None:
            self.strides = [2 for _ in range(len(self.depth))]
        self.strides = list(self.strides)
        for i in range(len(self.strides)):
            if isinstance(self.strides[i], int):
                self.strides[i] = [self.strides[i] for _ in range(self.spatial_dimensions)]

    def init_first_encoder(self):
        self.first_encoder = torch.nn.Sequential(
            self.adn_fn(self.in_channels),
            self.conv_op_enc(self.in_channels, self.depth[0], 3, padding="same"),
            self.adn_fn(self.depth[0]))

    def init_final_layer(self):
        self.final_layer = self.get_final_layer(self.depth[0] * 2)

    def init_swin_blocks(self):
        self.in_channels_rec = []
        common = dict(patch_size=self.patch_size, window_size=self.window_size,
                      dropout_rate=self.dropout_rate, mlp_structure=self.mlp_structure,
                      adn_fn=self.adn_fn_mlp)
        self.first_swin_block = SWINTransformerBlockStack(
            image_size=self.image_size, in_channels=self.in_channels,
            shift_sizes=self.shift_sizes[0], attention_dim=self.embedding_size[0],
            hidden_dim=self.embedding_size[0], embedding_size=self.embedding_size[0],
            n_heads=self.n_heads[0], dropout_rate_embedding=self.dropout_rate,
            embed_method=self.embed_method, use_pos_embed=True, **common)
        self.swin_blocks = torch.nn.ModuleList([])
        image_size = self.image_size
        for i in range(self.number_of_blocks - 1):
            in_channels = self.in_channels
            if i > 0:
                in_channels *= int(np.prod([np.prod(s) for s in self.strides[:i]]))
                self.in_channels_rec.append(in_channels)
            self.swin_blocks.append(SWINTransformerBlockStack(
                image_size=image_size, in_channels=in_channels,
                shift_sizes=self.shift_sizes[i + 1], attention_dim=self.embedding_size[i + 1],
                hidden_dim=self.embedding_size[i + 1], embedding_size=self.embedding_size[i + 1],
                n_heads=self.n_heads[i + 1], dropout_rate_embedding=0.0, embed_method="linear",
                use_pos_embed=False, **common))
            image_size = [x // s for x, s in zip(image_size, self.strides[i])]
        self.in_channels_rec.append(
            self.in_channels * int(np.prod([np.prod(s) for s in self.strides[:-1]])))

    def init_reconstruction_ops(self):
        layer_norm = get_adn_fn(self.spatial_dimensions, "layer", None, 0.0)
        self.first_rec_op = torch.nn.Sequential(
            layer_norm(self.in_channels),
            self.conv_op_enc(self.in_channels, self.depth[0], 3, padding="same"),
            self.adn_fn(self.depth[0]))
        self.reconstruction_ops = torch.nn.ModuleList([])
        for i, d in enumerate(self.depth[1:]):
            self.reconstruction_ops.append(torch.nn.Sequential(
                layer_norm(self.in_channels_rec[i]),
                self.conv_op_enc(self.in_channels_rec[i], d, 1, padding="same"),
                self.conv_op_enc(d, d, 3, padding="same"),
                self.adn_fn(d)))

    def forward(self, X: torch.Tensor, X_skip_layer: torch.Tensor = None,
                X_feature_conditioning: torch.Tensor = None, return_features=False,
                return_bottleneck=False, return_logits=False):
        if not X.is_cuda:
            raise AdellHipError("adell_mri_amd.SWINUNet runs on MI355X only (no CPU fallback)")
        if X_feature_conditioning is not None:
            X_feature_conditioning = (X_feature_conditioning - self.f_mean) / self.f_std
        if X_skip_layer is not None and len(X_skip_layer.shape) < len(X.shape):
            X_skip_layer = X_skip_layer.unsqueeze(1)

        X_encoded_first = self.first_encoder(X)
        curr = self.first_swin_block(X)
        encoding_out = [self.first_rec_op(curr)]
        for i in range(len(self.swin_blocks)):
            curr = self.swin_blocks[i](curr, scale=self.strides[i])
            encoding_out.append(self.reconstruction_ops[i](curr))
        curr = encoding_out[-1]
        bottleneck = curr
        if return_bottleneck is True:
            return None, None, bottleneck
        elif self.encoder_only is True:
            return bottleneck

        curr, deep_outputs = self._run_decoder(encoding_out, curr, X_skip_layer,
                                               X_feature_conditioning)
        # final_layer(cat[X_encoded_first, curr]) without materialising the concat
        mods = list(self.final_layer)
        head = mods[0](X_encoded_first, X_cat=curr)
        for mod in mods[1:-1]:
            head = mod(head)
        if return_logits is not True:
            if isinstance(mods[-1], torch.nn.Sigmoid):
                head = (HF.norm_drop_act(head.unsqueeze(2), act="sigmoid").squeeze(2)
                        if head.dim() == 4 else HF.norm_drop_act(head, act="sigmoid"))
            else:
                head = mods[-1](head)
        return self._outputs(head, curr, bottleneck, deep_outputs, return_features)
