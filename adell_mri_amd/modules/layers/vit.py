"""Vision-transformer encoder pieces on the HIP kernels (mirror of
adell_mri/modules/layers/vit.py: LinearEmbedding :389-881, TransformerBlock
:884-1002, TransformerBlockStack :1259-1435, ViT :1619-1802) -- the subset UNETR
uses: "linear" embedding without windows, class token or registers.

The patch <-> token rearrangements are pure index permutations (einops in the
reference); here they are reshape/permute views plus one copy. Everything with
arithmetic -- LayerNorm, Linear (MFMA conv kernel), QK-norm attention, GELU,
residual adds (fused into the Linear epilogues) -- runs in libadellhip.
"""
from typing import Callable, List, Union

import numpy as np
import torch

from ... import functional as HF
from .adn_fn import get_adn_fn
from .linear_blocks import MLP, LayerNorm, Linear, MultiHeadSelfAttention


class LinearEmbedding(torch.nn.Module):
    def __init__(self, image_size, patch_size, in_channels: int, out_dim: int = None,
                 window_size=None, dropout_rate: float = 0.0, embed_method: str = "linear",
                 use_pos_embed: bool = True, use_class_token: bool = False,
                 n_registers: int = 0, learnable_embedding: bool = True,
                 channel_to_token: bool = False, channels_last: bool = False):
        super().__init__()
        self.image_size = list(image_size)
        self.patch_size = list(patch_size)
        self.in_channels = in_channels
        self.out_dim = out_dim
        self.window_size = window_size
        self.dropout_rate = dropout_rate
        self.embed_method = embed_method
        self.use_pos_embed = use_pos_embed
        self.use_class_token = use_class_token
        self.n_registers = n_registers
        self.learnable_embedding = learnable_embedding
        self.channel_to_token = channel_to_token
        self.channels_last = channels_last
        assert self.embed_method in ["linear", "convolutional"], "embed_method must be linear or convolutional"
        assert len(self.image_size) == len(self.patch_size), "image_size and patch_size must have the same length"
        unsupported = (embed_method != "linear" or window_size is not None or use_class_token
                       or n_registers > 0 or channel_to_token or channels_last
                       or not learnable_embedding)
        if unsupported:
            raise NotImplementedError("HIP LinearEmbedding covers the UNETR configuration only "
                                      "(linear embedding, no windows / class token / registers)")
        self.n_dims = len(self.image_size)
        self.n_patches_split = [x // y for x, y in zip(self.image_size, self.patch_size)]
        self.n_patches = int(np.prod(self.n_patches_split))
        self.n_features = int(np.prod(self.patch_size) * self.in_channels)
        self.map_to_out = torch.nn.Identity()
        self.map_to_in = torch.nn.Identity()
        if self.out_dim is not None and self.out_dim != self.n_features:
            self.map_to_out = torch.nn.Sequential(LayerNorm(self.n_features),
                                                  Linear(self.n_features, self.out_dim))
            self.map_to_in = Linear(self.out_dim, self.n_features)
        self.drop_op = torch.nn.Dropout(self.dropout_rate)
        if self.use_pos_embed:
            self.positional_embedding = torch.nn.Parameter(
                torch.rand(1, self.n_patches, self.true_n_features))
            torch.nn.init.trunc_normal_(self.positional_embedding, std=0.02)
        self.linearized_dim = [-1, self.n_patches, self.n_features]

    @property
    def true_n_features(self):
        return self.out_dim if self.out_dim else self.n_features

    # "b c (h x) (w y) (d z) -> b (h w d) (x y z c)"  (vit.py:646-676)
    def _to_tokens(self, X):
        b, c = X.shape[:2]
        n = self.n_dims
        hs, ps = self.n_patches_split, self.patch_size
        shape = [b, c]
        for h, p in zip(hs, ps):
            shape += [h, p]
        X = X.reshape(shape)
        grid_axes = [2 + 2 * i for i in range(n)]
        patch_axes = [3 + 2 * i for i in range(n)]
        X = X.permute(0, *grid_axes, *patch_axes, 1)
        return X.reshape(b, self.n_patches, self.n_features)

    def rearrange(self, X):
        X = self._to_tokens(X)
        return self.map_to_out(X)

    def _from_tokens(self, X, scale):
        """inverse rearrangement with the per-axis factor ``scale`` moved from the patch
        axes to the channel axis (vit.py:725-750, 812-842):
        "b (h w d) (x s1 y s2 z s3 c) -> b (c s1 s2 s3) (h x) (w y) (d z)"."""
        b = X.shape[0]
        n = self.n_dims
        hs = self.n_patches_split
        ps = [p // s for p, s in zip(self.patch_size, scale)]
        shape = [b, *hs]
        for p, s in zip(ps, scale):
            shape += [p, s]
        shape.append(self.in_channels)
        X = X.reshape(shape)
        # axes: 0 b | 1..n grid | then (x_i, s_i) pairs | last c
        c_ax = 1 + n + 2 * n
        s_axes = [1 + n + 2 * i + 1 for i in range(n)]
        perm = [0, c_ax, *s_axes]
        for i in range(n):
            perm += [1 + i, 1 + n + 2 * i]
        X = X.permute(perm)
        out_c = self.in_channels * int(np.prod(scale))
        return X.reshape(b, out_c, *[h * p for h, p in zip(hs, ps)])

    def rearrange_inverse_basic(self, X):
        return self._from_tokens(self.map_to_in(X), [1] * self.n_dims)

    def rearrange_rescale(self, X, scale):
        X = self.map_to_in(X)
        if isinstance(scale, int):
            scale = [scale] * self.n_dims
        return self._from_tokens(X, list(scale))

    def forward(self, X, no_pos_embed: bool = False):
        X = self.rearrange(X)
        if (no_pos_embed is False) and (self.use_pos_embed is True):
            X = HF.add_bcast(X, self.positional_embedding)
        if self.dropout_rate > 0 and self.training:
            X = HF.elementwise(X, drop_p=self.dropout_rate, training=True)
        return X


class TransformerBlock(torch.nn.Module):
    def __init__(self, input_dim_primary: int, attention_dim: int, hidden_dim: int,
                 n_heads: int = 4, mlp_structure: List[int] = [128, 128],
                 dropout_rate: float = 0.0, window_size=None,
                 adn_fn: Callable = get_adn_fn(1, "identity", "gelu")):
        super().__init__()
        self.input_dim_primary = input_dim_primary
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.n_heads = n_heads
        self.mlp_structure = mlp_structure
        self.dropout_rate = dropout_rate
        self.window_size = window_size
        self.adn_fn = adn_fn
        self.mha = MultiHeadSelfAttention(input_dim_primary, attention_dim, hidden_dim,
                                          input_dim_primary, window_size=window_size,
                                          dropout_rate=dropout_rate, n_heads=n_heads)
        self.drop_op_1 = torch.nn.Dropout(self.dropout_rate)
        self.drop_op_2 = torch.nn.Dropout(self.dropout_rate)
        self.norm_op_1 = LayerNorm(self.input_dim_primary)
        self.norm_op_2 = LayerNorm(self.input_dim_primary)
        structure = mlp_structure if isinstance(mlp_structure, list) else [mlp_structure]
        self.mlp = MLP(input_dim=input_dim_primary, output_dim=input_dim_primary,
                       structure=structure, adn_fn=adn_fn)

    def forward(self, X, mask=None, return_attention: bool = False):
        drop = self.training and self.dropout_rate > 0
        if drop or return_attention:
            attention = self.mha(self.norm_op_1(X), mask=mask)
            a = HF.elementwise(attention, drop_p=self.dropout_rate, training=True) if drop else attention
            X = X + a
            m = self.mlp(self.norm_op_2(X))
            X = X + (HF.elementwise(m, drop_p=self.dropout_rate, training=True) if drop else m)
            return (X, attention) if return_attention else X
        # residual adds fused into the epilogues of the two closing Linear layers
        X = self.mha(self.norm_op_1(X), mask=mask, residual=X)
        return self.mlp(self.norm_op_2(X), residual=X)


class TransformerBlockStack(torch.nn.Module):
    def __init__(self, number_of_blocks: int, input_dim_primary: int, attention_dim: int,
                 hidden_dim: int, n_heads: int = 4, mlp_structure: List[int] = [128],
                 dropout_rate: float = 0.0, adn_fn: Callable = get_adn_fn(1, "identity", "gelu"),
                 window_size=None):
        super().__init__()
        self.number_of_blocks = number_of_blocks
        self.input_dim_primary = input_dim_primary
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.n_heads = n_heads
        self.mlp_structure = mlp_structure
        self.dropout_rate = dropout_rate
        self.adn_fn = adn_fn
        self.window_size = window_size

        def as_list(x):
            x = x if isinstance(x, list) else [x for _ in range(number_of_blocks)]
            assert len(x) == number_of_blocks
            return x

        mlp = mlp_structure
        if not (isinstance(mlp, list) and len(mlp) > 0 and isinstance(mlp[0], list)):
            mlp = [mlp for _ in range(number_of_blocks)]
        assert len(mlp) == number_of_blocks
        self.transformer_blocks = torch.nn.ModuleList([
            TransformerBlock(input_dim_primary=i, attention_dim=a, hidden_dim=h, n_heads=n,
                             mlp_structure=m, dropout_rate=dropout_rate, window_size=window_size,
                             adn_fn=adn_fn)
            for i, a, h, n, m in zip(as_list(input_dim_primary), as_list(attention_dim),
                                     as_list(hidden_dim), as_list(n_heads), mlp)])

    def forward(self, X, return_at: Union[str, List[int]] = "end", return_attention: bool = False):
        if isinstance(return_at, list):
            assert max(return_at) < self.number_of_blocks, \
                "max(return_at) should be smaller than self.number_of_blocks"
        if return_at == "end" or return_at is None:
            return_at = []
        outputs, attentions = [], []
        attention = None
        for i, block in enumerate(self.transformer_blocks):
            X = block(X, return_attention=return_attention)
            if return_attention is True:
                X, attention = X
            if i in return_at:
                outputs.append(X)
                attentions.append(attention)
        if return_attention is True:
            return X, outputs, ([attention] if return_at == [] else attentions)
        return X, outputs


class ViT(torch.nn.Module):
    def __init__(self, image_size, patch_size, in_channels: int, number_of_blocks: int,
                 attention_dim: int, hidden_dim: int = None, embedding_size: int = None,
                 window_size=None, n_heads: int = 4, dropout_rate: float = 0.0,
                 use_pos_embed: bool = True, embed_method: str = "linear",
                 mlp_structure: Union[List[int], float] = [128],
                 adn_fn=get_adn_fn(1, "identity", "gelu"), use_class_token: bool = False,
                 n_registers: int = 0, learnable_embedding: bool = True,
                 channel_to_token: bool = False, patch_erasing: float = None):
        super().__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.in_channels = in_channels
        self.number_of_blocks = number_of_blocks
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.window_size = window_size
        self.n_heads = n_heads
        self.embedding_size = embedding_size
        self.dropout_rate = dropout_rate
        self.use_pos_embed = use_pos_embed
        self.embed_method = embed_method
        self.mlp_structure = mlp_structure
        self.adn_fn = adn_fn
        self.use_class_token = use_class_token
        self.n_registers = n_registers
        self.learnable_embedding = learnable_embedding
        self.channel_to_token = channel_to_token
        self.patch_erasing = patch_erasing
        if patch_erasing is not None:
            raise NotImplementedError("patch erasing is outside the HIP path built so far")
        self.embedding = LinearEmbedding(
            image_size=image_size, patch_size=patch_size, in_channels=in_channels,
            window_size=window_size, out_dim=embedding_size, embed_method=embed_method,
            use_pos_embed=use_pos_embed, dropout_rate=dropout_rate,
            use_class_token=use_class_token, n_registers=n_registers,
            learnable_embedding=learnable_embedding, channel_to_token=channel_to_token)
        self.input_dim_primary = self.embedding.true_n_features
        self.patch_erasing_op = None
        if isinstance(self.mlp_structure, float):
            self.mlp_structure = [int(self.input_dim_primary * self.mlp_structure)]
        idp = embedding_size if embedding_size is not None else self.input_dim_primary
        self.tbs = TransformerBlockStack(
            number_of_blocks=number_of_blocks, input_dim_primary=idp,
            attention_dim=idp if attention_dim is None else attention_dim,
            hidden_dim=idp if hidden_dim is None else hidden_dim, n_heads=n_heads,
            mlp_structure=self.mlp_structure, dropout_rate=dropout_rate, adn_fn=adn_fn,
            window_size=window_size)

    def forward(self, X, return_at: Union[str, List[int]] = "end"):
        if isinstance(return_at, list):
            assert max(return_at) < self.number_of_blocks, \
                "max(return_at) should be smaller than self.number_of_blocks"
        embeded_X = self.embedding(X)
        if return_at == "end" or return_at is None:
            return_at = []
        outputs = []
        for i, block in enumerate(self.tbs.transformer_blocks):
            embeded_X = block(embeded_X)
            if i in return_at:
                outputs.append(embeded_X)
        return embeded_X, outputs
