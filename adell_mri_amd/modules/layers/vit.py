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
from .regularization import ChannelDropout


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
        self.n_dims = len(self.image_size)
        self.windowed = window_size is not None
        if channel_to_token:
            # every channel of a patch is its own token (vit.py:484-487, 566-571)
            assert self.embed_method == "linear", \
                "embed_method must be 'linear' if channel_to_token == True"
            if window_size is not None:
                raise NotImplementedError("HIP windowed LinearEmbedding: no channel tokens")
        if (use_class_token or n_registers > 0) and window_size is not None:
            raise NotImplementedError("HIP windowed LinearEmbedding: no class token / registers")
        if self.windowed:
            # SWIN configuration (vit.py:553-571): tokens are the patches of one window
            if not channels_last or self.n_dims not in (2, 3):
                raise NotImplementedError("HIP windowed LinearEmbedding: channels_last=True")
            self.n_windows = [x // y for x, y in zip(self.image_size, self.window_size)]
            self.n_patches_split = [x // z // y for x, y, z in
                                    zip(self.image_size, self.patch_size, self.n_windows)]
            # two dimensions run as depth-1 volumes [b, 1, H, W, c] on the same gather kernels: the
            # einops patterns of vit.py:622-676 with a leading singleton factor per group
            lift = [1] * (3 - self.n_dims)
            self._win3 = (lift + self.n_windows, lift + self.n_patches_split,
                          lift + self.patch_size, lift + self.image_size)
        else:
            if embed_method != "linear" or channels_last:
                raise NotImplementedError("HIP LinearEmbedding without windows covers the UNETR "
                                          "configuration only (linear embedding, channels first)")
            self.n_patches_split = [x // y for x, y in zip(self.image_size, self.patch_size)]
        extra_patches, extra_features = ((self.in_channels, 1) if self.channel_to_token
                                         else (1, self.in_channels))
        self.n_patches = int(np.prod(self.n_patches_split) * extra_patches)
        self.n_features = int(np.prod(self.patch_size) * extra_features)
        if self.embed_method == "convolutional":
            # parameter container only: the patch convolution runs as gather + GEMM
            conv_cls = torch.nn.Conv2d if self.n_dims == 2 else torch.nn.Conv3d
            self.conv = conv_cls(self.in_channels, self.true_n_features, self.patch_size,
                                 stride=self.patch_size)
        self.map_to_out = torch.nn.Identity()
        self.map_to_in = torch.nn.Identity()
        if self.out_dim is not None and self.out_dim != self.n_features:
            if self.embed_method == "linear":
                self.map_to_out = torch.nn.Sequential(LayerNorm(self.n_features),
                                                      Linear(self.n_features, self.out_dim))
            self.map_to_in = Linear(self.out_dim, self.n_features)
        self.drop_op = torch.nn.Dropout(self.dropout_rate)
        # (parameter order of vit.py:489-495: class token, registers, positional embedding)
        if self.use_class_token is True:
            self.class_token = torch.nn.Parameter(torch.zeros([1, 1, self.true_n_features]))
        if self.n_registers > 0:
            self.registers = torch.nn.Parameter(
                torch.zeros([1, self.n_registers, self.true_n_features]))
        if self.use_pos_embed:
            if self.learnable_embedding is True:
                self.positional_embedding = torch.nn.Parameter(
                    torch.rand(1, self.n_patches, self.true_n_features))
                torch.nn.init.trunc_normal_(self.positional_embedding, std=0.02)
            else:       # fixed sinusoidal table (vit.py:210-218, 600-607)
                sin_embed = sinusoidal_positional_encoding(
                    self.n_patches, self.true_n_features)[np.newaxis, :, :]
                self.positional_embedding = torch.nn.Parameter(
                    torch.as_tensor(sin_embed, dtype=torch.float32), requires_grad=False)
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
        if self.channel_to_token:
            # "b c (h x) (w y) (d z) -> b (h w c d) (x y z)": the reference's token order puts the
            # channel index after w -- in front of d in three dimensions (vit.py:622-645)
            X = X.permute(0, *grid_axes[:2], 1, *grid_axes[2:], *patch_axes)
        else:
            X = X.permute(0, *grid_axes, *patch_axes, 1)
        return X.reshape(b, self.n_patches, self.n_features)

    def rearrange(self, X):
        X = self._to_tokens(X)
        return self.map_to_out(X)

    # ---- windowed (SWIN) path: channels-last images <-> [b, windows, tokens, features] ----
    def window_tokens(self, X, shift=(0, 0, 0, 0)):
        """Embedding of torch.roll(X, -shift) for X [b, *image_size, c]: window partition
        (one gather), then LayerNorm + Linear ("linear") or the patch convolution as a GEMM
        over (x y z c) features ("convolutional"), + positional embedding, dropout."""
        nwin, ppw, patch, _ = self._win3
        tokens = HF.window_partition(X, nwin, ppw, patch, shift)
        if self.embed_method == "convolutional":
            perm = (0, 2, 3, 1) if self.n_dims == 2 else (0, 2, 3, 4, 1)
            w = self.conv.weight.permute(*perm).reshape(self.true_n_features, -1)
            tokens = HF.linear(tokens, w, self.conv.bias)
        else:
            tokens = self.map_to_out(tokens)
        if self.use_pos_embed is True:
            tokens = HF.add_bcast(tokens, self.positional_embedding)
        if self.dropout_rate > 0 and self.training:
            tokens = HF.elementwise(tokens, drop_p=self.dropout_rate, training=True)
        return tokens

    def window_image(self, tokens):
        """rearrange_inverse / rearrange_inverse_basic of the windowed embedding
        (vit.py:777-811): map_to_in, then tokens back to [b, *image_size, c]."""
        tokens = self.map_to_in(tokens)
        nwin, ppw, patch, image = self._win3
        shape = (tokens.shape[0], *image, self.in_channels)
        return HF.window_merge(tokens, shape, nwin, ppw, patch)

    def _from_tokens(self, X, scale):
        """inverse rearrangement with the per-axis factor ``scale`` moved from the patch
        axes to the channel axis (vit.py:725-750, 812-842):
        "b (h w d) (x s1 y s2 z s3 c) -> b (c s1 s2 s3) (h x) (w y) (d z)"."""
        b = X.shape[0]
        n = self.n_dims
        hs = self.n_patches_split
        ps = [p // s for p, s in zip(self.patch_size, scale)]
        if self.channel_to_token:
            # "b (h w c d) (x s1 y s2 z s3) -> b (c s1 s2 s3) (h x) (w y) (d z)"
            shape = [b, *hs[:2], self.in_channels, *hs[2:]]
            for p, sc in zip(ps, scale):
                shape += [p, sc]
            X = X.reshape(shape)
            g_ax = [1, 2] + ([4] if n == 3 else [])      # grid axes h, w (, d)
            c_ax = 3
            first = 4 + (1 if n == 3 else 0)             # first (x, s1) pair
            perm = [0, c_ax] + [first + 2 * i + 1 for i in range(n)]
            for i in range(n):
                perm += [g_ax[i], first + 2 * i]
            X = X.permute(perm)
            out_c = self.in_channels * int(np.prod(scale))
            return X.reshape(b, out_c, *[h * p for h, p in zip(hs, ps)])
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
        if self.windowed:
            return self.window_tokens(X)
        X = self.rearrange(X)
        if (no_pos_embed is False) and (self.use_pos_embed is True):
            X = HF.add_bcast(X, self.positional_embedding)
        # class token, then registers, in front of the patch tokens (vit.py:871-880)
        if self.use_class_token is True:
            X = self._prepend(self.class_token, X)
        if self.n_registers > 0:
            X = self._prepend(self.registers, X)
        if self.dropout_rate > 0 and self.training:
            X = HF.elementwise(X, drop_p=self.dropout_rate, training=True)
        return X

    @staticmethod
    def _prepend(tokens, X):
        """[1, n, E] parameter rows repeated per batch item in front of X [B, T, E]."""
        rows = HF.add_bcast(X.new_zeros((X.shape[0], tokens.shape[1], tokens.shape[2])), tokens)
        return HF.cat_tokens(rows, X.contiguous())


def sinusoidal_positional_encoding(n_tokens, dim_size):
    """vit.py:210-218."""
    token_range = np.arange(0, n_tokens)[:, np.newaxis]
    dim_range = np.arange(0, dim_size)[np.newaxis, :]
    radians = token_range / (10000 ** (2 * dim_range / dim_size))
    output = np.zeros((n_tokens, dim_size))
    output[:, ::2] = np.sin(radians)[:, ::2]
    output[:, 1::2] = np.cos(radians)[:, 1::2]
    return output


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
        self.embedding = LinearEmbedding(
            image_size=image_size, patch_size=patch_size, in_channels=in_channels,
            window_size=window_size, out_dim=embedding_size, embed_method=embed_method,
            use_pos_embed=use_pos_embed, dropout_rate=dropout_rate,
            use_class_token=use_class_token, n_registers=n_registers,
            learnable_embedding=learnable_embedding, channel_to_token=channel_to_token)
        self.input_dim_primary = self.embedding.true_n_features
        # patch erasing (vit.py:1731-1736): whole tokens zeroed per item, or the caller's own module
        if self.patch_erasing is None:
            self.patch_erasing_op = None
        elif callable(self.patch_erasing):
            self.patch_erasing_op = self.patch_erasing
        else:
            self.patch_erasing_op = ChannelDropout(self.patch_erasing)
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
        if self.patch_erasing_op is not None:
            embeded_X = self.patch_erasing_op(embeded_X)
        if return_at == "end" or return_at is None:
            return_at = []
        outputs = []
        for i, block in enumerate(self.tbs.transformer_blocks):
            embeded_X = block(embeded_X)
            if i in return_at:
                outputs.append(embeded_X)
        return embeded_X, outputs


def move_axis(X: torch.Tensor, axis1: int, axis2: int) -> torch.Tensor:
    axes = list(range(len(X.shape)))
    if axis1 < 0:
        axis1 = axes[axis1]
    if axis2 < 0:
        axis2 = axes[axis2]
    axes.insert(axis2, axes.pop(axis1))
    return X.permute(tuple(axes))


def einops_rescale(X: torch.Tensor, scale) -> torch.Tensor:
    """'b c (h p1) (w p2) (d p3) -> b (c p1 p2 p3) h w d' (vit.py:33-45) as one gather; a 4-D input
    ('b c (h p1) (w p2) -> b (c p1 p2) h w') runs as a depth-1 volume."""
    if isinstance(scale, int):
        scale = [scale] * (X.dim() - 2)
    if all(int(s) == 1 for s in scale):
        return X
    if X.dim() == 4:
        return HF.space_to_depth(X.unsqueeze(2), [1] + [int(s) for s in scale]).squeeze(2)
    return HF.space_to_depth(X, [int(s) for s in scale])


def generate_mask(image_size, window_size, shift_size):
    """Additive attention mask of shifted windows (vit.py:132-207), in patch units:
    [n_windows, tokens, tokens] with -100 between tokens of different shift regions. Host
    logic, evaluated once per block. Like the reference, region labels are read with the
    window index as the FAST factor of each axis ("(w1 h)": position = w1 * n_windows + h)."""
    nd = len(image_size)
    if not isinstance(window_size, list):
        window_size = [window_size for _ in image_size]
    if not isinstance(shift_size, list):
        shift_size = [shift_size for _ in image_size]
    if not any(x > 0 for x in shift_size):
        return None
    label = np.zeros(tuple(image_size), dtype=np.float32)
    regions = [(slice(0, -w), slice(-w, -s), slice(-s, None))
               for w, s in zip(window_size, shift_size)]
    cnt = 0
    for combo in np.ndindex(*[3] * nd):
        label[tuple(regions[i][c] for i, c in enumerate(combo))] = cnt
        cnt += 1
    n_win = [s // w for s, w in zip(image_size, window_size)]
    shape = []
    for w, h in zip(window_size, n_win):
        shape += [w, h]
    lab = label.reshape(shape)                                   # (w1 h w2 w w3 d)
    perm = [2 * i + 1 for i in range(nd)] + [2 * i for i in range(nd)]
    lab = lab.transpose(perm).reshape(int(np.prod(n_win)), int(np.prod(window_size)))
    diff = lab[:, None, :] - lab[:, :, None]
    return torch.from_numpy(np.where(diff != 0, -100.0, 0.0).astype(np.float32))


class SWINTransformerBlock(torch.nn.Module):
    """Shifted-window transformer block (vit.py:1005-1198 ``forward``). Image in, image out:
    [b, c, *image_size] -> cyclic shift + window partition + embedding (one gather + one
    GEMM) -> LayerNorm -> windowed attention -> Linear -> window merge -> + input ->
    per-voxel LayerNorm / MLP over the c channels -> (space-to-depth by ``scale``).

    Reference behaviour kept on purpose (vit.py:1130-1145, 1190-1212): the cyclic shift is by
    ``shift_size`` elements (not patches; the mask is built for ``shift_size`` patches) along
    dims 2, 3, 4 of the channels-last [b, X, Y, Z, c] tensor -- i.e. Y, Z and the CHANNEL
    axis, X stays put -- and the attention output is added to the shortcut in that shifted
    frame (the roll back at vit.py:1207-1210 is overwritten before use)."""

    def __init__(self, image_size, patch_size, window_size, in_channels: int,
                 attention_dim: int = None, hidden_dim: int = None, embedding_size: int = None,
                 shift_size: int = 0, n_heads: int = 4, dropout_rate: float = 0.0,
                 dropout_rate_embedding: float = 0.0, embed_method: str = "linear",
                 mlp_structure: Union[List[int], float] = [32, 32], use_pos_embed: bool = False,
                 adn_fn=get_adn_fn(1, "identity", "gelu")):
        super().__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.window_size = window_size
        self.in_channels = in_channels
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.embedding_size = embedding_size
        self.shift_size = shift_size
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate
        self.dropout_rate_embedding = dropout_rate_embedding
        self.embed_method = embed_method
        self.mlp_structure = mlp_structure
        self.use_pos_embed = use_pos_embed
        self.adn_fn = adn_fn
        self.init_embedding()
        self.init_drop_ops()
        self.init_layers()
        self.init_mask_if_necessary()

    def init_embedding(self):
        self.embedding = LinearEmbedding(
            image_size=self.image_size, patch_size=self.patch_size, in_channels=self.in_channels,
            window_size=self.window_size, dropout_rate=self.dropout_rate_embedding,
            embed_method=self.embed_method, out_dim=self.embedding_size,
            use_pos_embed=self.use_pos_embed, channels_last=True)
        self.input_dim_primary = self.embedding.true_n_features

    def init_mask_if_necessary(self):
        self.attention_mask = generate_mask(
            image_size=[x // y for x, y in zip(self.image_size, self.patch_size)],
            window_size=[x // y for x, y in zip(self.window_size, self.patch_size)],
            shift_size=self.shift_size)
        self._mask_dev = None

    def init_layers(self):
        if isinstance(self.mlp_structure, float):
            self.mlp_structure = [int(self.in_channels * self.mlp_structure)]
        d = self.input_dim_primary
        hidden_dim = d if self.hidden_dim is None else self.hidden_dim
        attention_dim = d if self.attention_dim is None else self.attention_dim
        self.mha = MultiHeadSelfAttention(d, attention_dim, hidden_dim, d,
                                          window_size=self.window_size,
                                          dropout_rate=self.dropout_rate, n_heads=self.n_heads)
        self.norm_op_1 = LayerNorm(d)
        self.norm_op_2 = LayerNorm(self.in_channels)
        self.mlp = MLP(self.in_channels, self.in_channels, self.mlp_structure, self.adn_fn)

    def init_drop_ops(self):
        self.drop_op_1 = torch.nn.Dropout(self.dropout_rate)
        self.drop_op_2 = torch.nn.Dropout(self.dropout_rate)

    def _mask(self, device):
        if self.attention_mask is None:
            return None
        if self._mask_dev is None or self._mask_dev.device != device:
            self._mask_dev = self.attention_mask.to(device)
        return self._mask_dev

    def _drop(self, X):
        if self.training and self.dropout_rate > 0:
            return HF.elementwise(X, drop_p=self.dropout_rate, training=True)
        return X

    def forward(self, X: torch.Tensor, scale=None) -> torch.Tensor:
        from ... import ops

        two_d = X.dim() == 4
        if two_d:                                      # [b, c, H, W] as the volume [b, c, 1, H, W]
            X = X.unsqueeze(2)
        X = ops.ndhwc(X).permute(0, 2, 3, 4, 1)        # move_axis(X, 1, -1): a view of NDHWC
        shortcut = X
        if scale is not None and isinstance(scale, int):
            scale = [scale for _ in self.patch_size]
        ss = self.shift_size
        if isinstance(ss, int):
            ss = [ss] * len(self.patch_size)
        # torch.roll(X, [-s...], dims=[2, 3, ...]) on the channels-last tensor (vit.py:1151-1163: the
        # dims start at 2): Y, Z and the channel axis of [b, X, Y, Z, c]; W and the channel axis of
        # [b, H, W, c] -- in the lifted frame [b, 1, H, W, c] that is (0, 0, s0, s1)
        shift = (0, 0, ss[0], ss[1]) if two_d else (0, ss[0], ss[1], ss[2])
        embedded = self.embedding.window_tokens(X, shift)
        attention = self.mha(self.norm_op_1(embedded), mask=self._mask(X.device))
        shifted = self.embedding.window_image(attention)
        drop = self.training and self.dropout_rate > 0
        X = HF.add(shortcut, self._drop(shifted))
        if drop:
            X = HF.add(X, self._drop(self.mlp(self.norm_op_2(X))))
        else:
            X = self.mlp(self.norm_op_2(X), residual=X)
        X = X.permute(0, 4, 1, 2, 3)                      # move_axis(X, -1, 1)
        if two_d:
            X = X.squeeze(2)
        if scale is not None:
            X = einops_rescale(X, scale)
        return X


class SWINTransformerBlockStack(torch.nn.Module):
    """vit.py:1438-1616: one SWINTransformerBlock per entry of ``shift_sizes``; only the
    first may use the convolutional embedding / positional embedding."""

    def __init__(self, image_size, patch_size, window_size, shift_sizes: List[int],
                 in_channels: int, attention_dim: int = None, hidden_dim: int = None,
                 embedding_size: int = None, n_heads: int = 4, dropout_rate: float = 0.0,
                 dropout_rate_embedding: float = 0.0, embed_method: str = "linear",
                 mlp_structure: Union[List[int], float] = [128], use_pos_embed: bool = False,
                 adn_fn=get_adn_fn(1, "identity", "gelu")):
        super().__init__()
        self.image_size = image_size
        self.patch_size = patch_size
        self.window_size = window_size
        self.shift_sizes = shift_sizes
        self.in_channels = in_channels
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.embedding_size = embedding_size
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate
        self.dropout_rate_embedding = dropout_rate_embedding
        self.embed_method = embed_method
        self.mlp_structure = mlp_structure
        self.use_pos_embed = use_pos_embed
        self.adn_fn = adn_fn
        self.init_swin_transformers()

    def convert_mlp_structure(self, x):
        if isinstance(x, float) is True:
            return [x for _ in self.shift_sizes]
        if isinstance(x[0], list) is False:
            return [x for _ in self.shift_sizes]
        return x

    def convert_to_list_if_necessary(self, x):
        if isinstance(x, list) is False:
            return [x for _ in self.shift_sizes]
        assert len(x) == len(self.shift_sizes)
        return x

    def init_swin_transformers(self):
        attention_dim = self.convert_to_list_if_necessary(self.attention_dim)
        hidden_dim = self.convert_to_list_if_necessary(self.hidden_dim)
        n_heads = self.convert_to_list_if_necessary(self.n_heads)
        dropout_rate = self.convert_to_list_if_necessary(self.dropout_rate)
        mlp_structure = self.convert_mlp_structure(self.mlp_structure)
        self.stbs = torch.nn.ModuleList([])
        first = True
        for ss, ad, hd, nh, dr, mlp_s in zip(self.shift_sizes, attention_dim, hidden_dim, n_heads,
                                             dropout_rate, mlp_structure):
            self.stbs.append(SWINTransformerBlock(
                image_size=self.image_size, patch_size=self.patch_size,
                window_size=self.window_size, in_channels=self.in_channels, attention_dim=ad,
                hidden_dim=hd, embedding_size=self.embedding_size, shift_size=ss, n_heads=nh,
                dropout_rate_embedding=self.dropout_rate_embedding, dropout_rate=dr,
                embed_method=self.embed_method if first else "linear", mlp_structure=mlp_s,
                adn_fn=self.adn_fn, use_pos_embed=first and self.use_pos_embed))
            first = False

    def forward(self, X: torch.Tensor, scale=1) -> torch.Tensor:
        for block in self.stbs[:-1]:
            X = block(X, scale=1)
        return self.stbs[-1](X, scale=scale)
