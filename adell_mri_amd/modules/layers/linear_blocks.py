"""MLP and multi-head self-attention on the HIP kernels (mirror of
adell_mri/modules/layers/linear_blocks.py:54-115, 248-417).

Module trees / state_dict keys are the reference's (``op.N`` Linear / ADN members,
``qkv``, ``q_norm``, ``k_norm``, ``output_layer``). The Linear layers run as
1x1x1 convolutions on the MFMA conv kernel; QK-LayerNorm and the attention core
are the token kernels of ``csrc/tokens.hip``.
"""
from typing import List

import numpy as np
import torch

from ... import functional as HF


class Linear(torch.nn.Linear):
    """torch.nn.Linear whose forward is the MFMA conv kernel (optionally adding a
    residual inside the epilogue)."""

    def forward(self, X, residual=None):
        return HF.linear(X, self.weight, self.bias, residual=residual)


class LayerNorm(torch.nn.LayerNorm):
    def forward(self, X):
        if len(self.normalized_shape) != 1:
            raise NotImplementedError("HIP LayerNorm normalises the last dimension only")
        return HF.layer_norm(X, self.weight, self.bias, self.eps)


def get_relative_position_indices(window_size) -> torch.Tensor:
    """Index into the relative-position bias table for every ordered pair of positions of a
    window (linear_blocks.py:17-51; row-major positions, offsets shifted to start at 0 and
    combined in mixed radix (2*ws-1))."""
    ws = [int(w) for w in window_size]
    coords = np.stack(np.meshgrid(*[np.arange(w) for w in ws], indexing="ij")).reshape(len(ws), -1)
    rel = coords[:, :, None] - coords[:, None, :]          # n, P, P
    index = np.zeros(rel.shape[1:], dtype=np.int64)
    for i, w in enumerate(ws):
        radix = int(np.prod([2 * v - 1 for v in ws[i + 1:]])) if i + 1 < len(ws) else 1
        index += (rel[i] + w - 1) * radix
    return torch.from_numpy(index)


class MLP(torch.nn.Module):
    def __init__(self, input_dim: int, output_dim: int, structure: List[int] = [],
                 adn_fn: torch.nn.Module = torch.nn.Identity):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.structure = structure
        self.adn_fn = adn_fn
        self.init_layers()

    def init_layers(self):
        curr_in = self.input_dim
        ops = torch.nn.ModuleList([])
        if len(self.structure) > 0:
            curr_out = self.structure[0]
            for i in range(1, len(self.structure)):
                ops.append(Linear(curr_in, curr_out))
                ops.append(self.adn_fn(curr_out))
                curr_in = curr_out
                curr_out = self.structure[i]
            ops.append(Linear(curr_in, curr_out))
        else:
            curr_out = curr_in
        ops.append(self.adn_fn(curr_out))
        ops.append(Linear(curr_out, self.output_dim))
        self.op = torch.nn.Sequential(*ops)

    def forward(self, X: torch.Tensor, residual: torch.Tensor = None) -> torch.Tensor:
        mods = list(self.op)
        if (len(mods) == 3 and isinstance(mods[0], Linear) and isinstance(mods[2], Linear)
                and hasattr(mods[1], "pure_activation")):
            # Linear -> activation -> Linear: the activation and its backward inside GEMM epilogues
            spec = mods[1].pure_activation()
            if spec is not None and HF.mlp_ok(X, mods[0].weight, mods[2].weight):
                return HF.mlp(X, mods[0].weight, mods[0].bias, mods[2].weight, mods[2].bias,
                              act=spec[0], act_p=spec[1], residual=residual)
        for mod in mods[:-1]:
            X = mod(X)
        return mods[-1](X, residual=residual) if residual is not None else mods[-1](X)


class MultiHeadSelfAttention(torch.nn.Module):
    def __init__(self, input_dim: int, attention_dim: int, hidden_dim: int, output_dim: int,
                 n_heads: int = 4, dropout_rate: float = 0.0, window_size: bool = False):
        super().__init__()
        self.input_dim = input_dim
        self.attention_dim = attention_dim
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate
        self.window_size = window_size
        assert (attention_dim % n_heads) == 0, "attention_dim must be divisible by n_heads"
        assert (hidden_dim % n_heads) == 0, "hidden_dim must be divisible by n_heads"
        self.init_layers()
        self.init_output_layer()
        self.init_weights()

    def init_layers(self):
        self.real_attention_dim = self.attention_dim // self.n_heads
        self.real_hidden_dim = self.hidden_dim // self.n_heads
        self.qkv_dim = self.attention_dim * 2 + self.hidden_dim
        self.qkv = Linear(self.input_dim, self.qkv_dim, bias=False)
        self.drop_op = torch.nn.Dropout(self.dropout_rate)
        self.q_norm = LayerNorm(self.real_attention_dim)
        self.k_norm = LayerNorm(self.real_attention_dim)
        if self.window_size:
            # linear_blocks.py:331-343: bias table over the (2*ws-1)^n relative offsets
            self.relative_position_bias_table = torch.nn.Parameter(torch.zeros(
                int(np.prod([2 * ws - 1 for ws in self.window_size])), self.n_heads))
            torch.nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
            self.relative_position_index = get_relative_position_indices(self.window_size)

    def init_output_layer(self):
        self.output_layer = Linear(self.hidden_dim, self.output_dim)

    def init_weights(self):
        torch.nn.init.xavier_uniform_(self.qkv.weight)
        torch.nn.init.xavier_uniform_(self.output_layer.weight)

    def relative_position_bias(self, t):
        """[n_heads, t, t] additive bias. As in the reference (linear_blocks.py:389-393) the
        gathered [t*t, n_heads] rows are RESHAPED (not permuted) to [n_heads, t, t]."""
        idx = self.relative_position_index[:t, :t].reshape(-1).to(
            self.relative_position_bias_table.device)
        return self.relative_position_bias_table[idx].reshape(self.n_heads, t, t)

    def forward(self, X: torch.Tensor, mask=None, residual: torch.Tensor = None) -> torch.Tensor:
        sh = X.shape
        b, t = sh[:-2], sh[-2]
        nb = 1
        for i in b:
            nb *= i
        a, h = self.real_attention_dim, self.real_hidden_dim
        if t <= 64 and a <= 32 and h <= 32 and (mask is None or mask.ndim == 3):
            # short sequences / windows: q-norm, k-norm, bias, mask, dropout and attention on
            # the QKV buffer in place (csrc/window.hip)
            rel = self.relative_position_bias(t) if self.window_size else None
            m = None
            if mask is not None:
                m = mask.to(device=X.device, dtype=torch.float32)
            O = HF.window_attention(self.qkv(X).reshape(nb * t, self.qkv_dim), self.q_norm.weight,
                                    self.q_norm.bias, self.k_norm.weight, self.k_norm.bias, nb,
                                    self.n_heads, t, a, h, rel=rel, mask=m,
                                    drop_p=self.dropout_rate, training=self.training,
                                    eps=self.q_norm.eps)
            return self.output_layer(O.view(*b, t, self.hidden_dim), residual=residual)
        if self.window_size:
            raise NotImplementedError("windowed attention beyond 64 tokens per window")
        bias = None
        if mask is not None:
            m = mask.to(device=X.device, dtype=torch.float32)
            if m.ndim == 3:
                m = m.unsqueeze(1)
            bias = m.expand(nb, self.n_heads, t, t).reshape(nb * self.n_heads, t, t)
        if self.q_norm.weight is not None and HF.seq_attention_ok(t, a, h):
            # q-norm, k-norm and attention on the projection output in place: no slices, no
            # permutes (functional._SeqAttnFn)
            O = HF.seq_attention(self.qkv(X).reshape(nb * t, self.qkv_dim), self.q_norm.weight,
                                 self.q_norm.bias, self.k_norm.weight, self.k_norm.bias, nb,
                                 self.n_heads, t, a, h, bias=bias, drop_p=self.dropout_rate,
                                 training=self.training, eps=self.q_norm.eps)
            return self.output_layer(O.view(*b, t, self.hidden_dim), residual=residual)
        QKV = self.qkv(X).reshape(nb, t, self.n_heads, 2 * a + h).permute(0, 2, 1, 3)
        Q = self.q_norm(QKV[..., :a].contiguous())      # per-head interleaved q | k | v
        K = self.k_norm(QKV[..., a:2 * a].contiguous())
        V = QKV[..., 2 * a:].contiguous()
        O = HF.attention(Q.reshape(nb * self.n_heads, t, a), K.reshape(nb * self.n_heads, t, a),
                         V.reshape(nb * self.n_heads, t, h), bias, drop_p=self.dropout_rate,
                         training=self.training)
        O = O.reshape(nb, self.n_heads, t, h).transpose(1, 2).reshape(*b, t, self.hidden_dim)
        return self.output_layer(O, residual=residual)
