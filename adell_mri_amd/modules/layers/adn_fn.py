"""Activation / dropout / normalisation builder (mirror of
adell_mri/modules/layers/adn_fn.py:17-230).

``ActDropNorm`` keeps the reference's module tree (``op.normalization``,
``op.dropout``, ``op.activation``: the stock torch modules own any parameters and
buffers, so ``state_dict`` keys match) but ``forward`` collapses each
Norm -> Dropout -> Activation run of the ordering into ONE fused HIP kernel
(``functional.norm_drop_act``); with the U-Net's "NDA" ordering that is one
launch per ADN site, fed by the statistics the producing conv already reduced.
"""
from collections import OrderedDict
from functools import partial

import torch

from ... import functional as HF
from ..activations import act_spec, activation_factory
from .regularization import LayerNormChannelsFirst, UOut

norm_fn_dict = {
    "batch": {1: torch.nn.BatchNorm1d, 2: torch.nn.BatchNorm2d, 3: torch.nn.BatchNorm3d},
    "instance": {1: torch.nn.InstanceNorm1d, 2: torch.nn.InstanceNorm2d,
                 3: torch.nn.InstanceNorm3d},
    "instance_affine": {
        1: partial(torch.nn.InstanceNorm1d, affine=True),
        2: partial(torch.nn.InstanceNorm2d, affine=True),
        3: partial(torch.nn.InstanceNorm3d, affine=True),
    },
    "layer": {1: torch.nn.LayerNorm, 2: LayerNormChannelsFirst, 3: LayerNormChannelsFirst},
    "identity": {1: torch.nn.Identity, 2: torch.nn.Identity, 3: torch.nn.Identity},
}

_RANK = {"N": 0, "D": 1, "A": 2}


def _as5d(X):
    """[N, C, *spatial] (1-3 spatial dims) -> [N, C, D, H, W] view and the inverse."""
    n = X.dim()
    if n == 5:
        return X, lambda t: t
    if n == 4:
        return X.unsqueeze(2), lambda t: t.squeeze(2)
    if n == 3:
        return X.unsqueeze(2).unsqueeze(2), lambda t: t.squeeze(2).squeeze(2)
    if n == 2:
        return X[:, :, None, None, None], lambda t: t[:, :, 0, 0, 0]
    raise ValueError(f"ActDropNorm expects [N, C, ...] with 1-3 spatial dims, got {tuple(X.shape)}")


class ActDropNorm(torch.nn.Module):
    def __init__(self, in_channels: int = None, ordering: str = "NDA",
                 norm_fn: torch.nn.Module = torch.nn.BatchNorm2d,
                 act_fn: torch.nn.Module = torch.nn.PReLU,
                 dropout_fn: torch.nn.Module = torch.nn.Dropout, dropout_param: float = 0.0,
                 inplace: bool = False):
        super().__init__()
        self.ordering = ordering
        self.norm_fn = norm_fn if norm_fn is not None else torch.nn.Identity
        self.in_channels = in_channels
        self.act_fn = act_fn if act_fn is not None else torch.nn.Identity
        self.dropout_fn = dropout_fn if dropout_fn is not None else torch.nn.Identity
        self.dropout_param = dropout_param
        self.inplace = inplace
        self.name_dict = {"A": "activation", "D": "dropout", "N": "normalization"}
        self.init_layers()

    def _make(self, k):
        if k == "A":
            try:
                return self.act_fn(inplace=self.inplace)
            except Exception:
                return self.act_fn()
        if k == "D":
            return self.dropout_fn(self.dropout_param)
        return self.norm_fn(self.in_channels)

    def init_layers(self):
        self.op_list = OrderedDict()
        for k in self.ordering:
            self.op_list[self.name_dict[k]] = self._make(k)
        self.op = torch.nn.Sequential(self.op_list)
        # fuse maximal N->D->A runs of the ordering into single kernel launches
        self._stages = []
        last = 3
        for k in self.ordering:
            if _RANK[k] <= last:
                self._stages.append({})
            self._stages[-1][k] = self.name_dict[k]
            last = _RANK[k]

    def _run_stage(self, X, stage, reader=None):
        kw = {"training": self.training}
        if "N" in stage:
            m = self.op_list[stage["N"]]
            if isinstance(m, torch.nn.modules.instancenorm._InstanceNorm):
                if m.track_running_stats:
                    raise NotImplementedError("InstanceNorm with running stats: no HIP kernel")
                kw.update(norm="instance", eps=m.eps, gamma=m.weight, beta=m.bias)
            elif isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                kw.update(norm="batch", eps=m.eps, gamma=m.weight, beta=m.bias,
                          momentum=m.momentum,
                          running=(m.running_mean, m.running_var, m.num_batches_tracked))
            elif isinstance(m, LayerNormChannelsFirst):
                X = m(X)
            elif isinstance(m, torch.nn.LayerNorm):
                # feature vectors / tokens ([..., C]): row LayerNorm kernel, then the rest of
                # the stage elementwise
                if X.dim() == 5 or tuple(m.normalized_shape) != (X.shape[-1],):
                    raise NotImplementedError("torch.nn.LayerNorm in an ADN: [..., C] inputs only")
                X = HF.layer_norm(X, m.weight, m.bias, m.eps)
            elif not isinstance(m, torch.nn.Identity):
                raise NotImplementedError(
                    f"normalisation {type(m).__name__} has no HIP kernel on the adell_mri_amd path")
        if "D" in stage:
            m = self.op_list[stage["D"]]
            if isinstance(m, torch.nn.Dropout):
                kw["drop_p"] = m.p
            elif isinstance(m, UOut):
                if self.training and m.beta != 0.0:
                    # not a per-element mask: apply what precedes it (the norm), U-out itself
                    # (one broadcast-scale kernel), then the rest of the stage
                    if kw.get("norm", "none") != "none":
                        X5, back = _as5d(X)
                        if hasattr(X, "_adell_partials") and X5 is not X:
                            X5._adell_partials = X._adell_partials
                        X = back(HF.norm_drop_act(X5, **kw))
                        kw = {"training": self.training}
                    X = m(X)
            elif not isinstance(m, torch.nn.Identity):
                raise NotImplementedError(
                    f"dropout {type(m).__name__} has no HIP kernel on the adell_mri_amd path")
        if "A" in stage:
            name, p, w = act_spec(self.op_list[stage["A"]])
            kw.update(act=name, act_p=p, act_w=w)
        if (kw.get("norm", "none") == "none" and kw.get("act", "identity") == "identity"
                and not (self.training and kw.get("drop_p", 0.0) > 0.0)):
            return X  # nothing left to apply (e.g. a LayerNorm-only ADN)
        if ((X.dim() != 5 or X.is_contiguous()) and kw.get("norm", "none") == "none"
                and kw.get("act_w") is None):
            # purely elementwise on a token / feature tensor: no channel semantics
            return HF.elementwise(X, act=kw.get("act", "identity"), act_p=kw.get("act_p", 0.0),
                                  drop_p=kw.get("drop_p", 0.0), training=self.training)
        X5, back = _as5d(X)
        if hasattr(X, "_adell_partials") and X5 is not X:
            X5._adell_partials = X._adell_partials
        if reader is not None and X5 is X:
            kw["rows_reader"] = reader
        return back(HF.norm_drop_act(X5, **kw))

    def pure_activation(self):
        """(name, parameter) when this module is nothing but a parameter-free element-wise
        activation right now (identity norm, dropout absent / zero / in eval mode) -- what
        ``functional.mlp`` can carry in a GEMM epilogue -- else None."""
        name = None
        for k in self.ordering:
            m = self.op_list[self.name_dict[k]]
            if k == "N" and not isinstance(m, torch.nn.Identity):
                return None
            if k == "D" and not isinstance(m, torch.nn.Identity):
                if not isinstance(m, torch.nn.Dropout) or (self.training and m.p > 0.0):
                    return None
            if k == "A":
                name, p, w = act_spec(m)
                if w is not None:
                    return None
        return None if name is None else (name, p)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        # the conv that module code announced as the only reader of this output
        # (functional.expect_rows): the last stage may then write split rows instead of fp32
        reader = HF.take_rows_reader(self)
        last = len(self._stages) - 1
        for i, stage in enumerate(self._stages):
            X = self._run_stage(X, stage, reader if i == last else None)
        return X


class ActDropNormBuilder:
    def __init__(self, ordering: str = "NDA", norm_fn=torch.nn.BatchNorm2d,
                 act_fn=torch.nn.PReLU, dropout_fn=torch.nn.Dropout, dropout_param: float = 0.0):
        self.ordering = ordering
        self.norm_fn = norm_fn
        self.act_fn = act_fn
        self.dropout_fn = dropout_fn
        self.dropout_param = dropout_param
        self.name_dict = {"A": "activation", "D": "dropout", "N": "normalization"}

    def __call__(self, in_channels: int):
        return ActDropNorm(in_channels=in_channels, ordering=self.ordering, norm_fn=self.norm_fn,
                           act_fn=self.act_fn, dropout_fn=self.dropout_fn,
                           dropout_param=self.dropout_param)


def get_adn_fn(spatial_dim: int, norm_fn: str = "batch", act_fn: str = "swish",
               dropout_param: float = 0.1) -> ActDropNormBuilder:
    if norm_fn not in norm_fn_dict:
        raise NotImplementedError("norm_fn must be one of {}".format(norm_fn_dict.keys()))
    norm_fn = norm_fn_dict[norm_fn][spatial_dim]
    if isinstance(act_fn, str):
        if act_fn not in activation_factory:
            raise NotImplementedError(
                "act_fn must be function or one of {}".format(activation_factory.keys()))
        act_fn = activation_factory[act_fn]
    return ActDropNormBuilder(norm_fn=norm_fn, act_fn=act_fn, dropout_param=dropout_param)
