"""U-Net on the MI355X kernels: drop-in mirror of
``adell_mri.modules.segmentation.unet.UNet`` (reference: adell_mri/modules/
segmentation/unet.py:31-843) for the 3-D training path.

What is kept verbatim from the reference: the constructor signature and
defaults (unet.py:43-68), attribute names, the module tree / ``state_dict``
keys (``encoding_operations.L.{0,1}...``, ``link_ops``, ``upscale_ops``,
``decoding_operations``, ``final_layer``), and ``forward``'s arguments and
return tuples (unet.py:751-843).

What is different underneath: every Conv3d / ConvTranspose3d / ActDropNorm is a
HIP-kernel leaf; ``torch.concat((upsampled, skip))`` (unet.py:817) is never
materialised -- the decoder's first convolution reads both tensors as one
virtual channel range; residual adds live in conv epilogues; activations stay
NDHWC between layers.
"""
from typing import Dict, List

import numpy as np
import torch

from ... import functional as HF
from ... import ops
from ..._lib import AdellHipError
from ..layers.adn_fn import ActDropNorm, get_adn_fn, norm_fn_dict
from ..layers.conv import (Conv2d, Conv3d, ConvTranspose2d, ConvTranspose3d, MaxPool2d,
                           MaxPool3d, Upsample)
from ..layers.linear_blocks import Linear
from ..layers.regularization import UOut
from ..layers.res_blocks import (DepthwiseConv2dStrided, DepthwiseConv3dStrided, ResidualBlock2d,
                                 ResidualBlock3d)
from ..layers.utils import crop_to_size


def _per_dim(v, n):
    return [v for _ in range(n)] if isinstance(v, int) else list(v)


class ConcatConvBlock(torch.nn.Sequential):
    """``Sequential(conv, adn, conv)`` whose first conv can take the two halves
    of a channel concat separately (same children / keys as the Sequential built
    by the reference's ``conv_block_3d``, unet.py:260-273)."""

    def forward(self, X, X_cat=None, carry_cat=None):
        mods = list(self)
        if X_cat is not None and carry_cat is not None and _takes_carry(mods[0]):
            h = mods[0](X, X_cat=X_cat, carry_cat=carry_cat)
        else:
            h = mods[0](X, X_cat=X_cat) if X_cat is not None else mods[0](X)
        return _run_rest(mods[1:], h)


def _run_rest(mods, h):
    """The remaining modules of a Sequential in turn. An ADN output that goes straight into a
    Conv3d has that conv as its only reader: functional.single_use lets the conv's backward-data
    kernel take over half of the site's backward."""
    for i, mod in enumerate(mods):
        only_reader = (isinstance(mod, ActDropNorm) and i + 1 < len(mods)
                       and type(mods[i + 1]) is Conv3d)
        if only_reader:
            HF.expect_rows(mod, mods[i + 1])     # ... which may take its input as split rows
        h = mod(h)
        if only_reader:
            h = HF.single_use(h)
    return h


def _last_adn(module):
    """The ActDropNorm whose output is the output of ``module`` (a decoder op, a residual link
    block), or None."""
    if isinstance(module, ResidualBlock3d):
        adn = getattr(module, "adn_op", None)
        return adn if isinstance(adn, ActDropNorm) and module.skip_activation is not True else None
    if isinstance(module, _DecoderOp) and len(module) > 0 and isinstance(module[-1], ActDropNorm):
        return module[-1]
    return None


def _first_conv(module):
    """The Conv3d that reads the input of ``module`` first (through nested Sequentials), or None."""
    while isinstance(module, torch.nn.Sequential) and len(module) > 0:
        module = module[0]
    return module if type(module) is Conv3d else None


def _takes_carry(module):
    return type(module) is Conv3d and module.takes_carry()


def _head_with_carry(module, X, **carries):
    """``module(X)`` with the functional.GradCarry arguments delivered to the Conv3d that reads X
    (through nested Sequentials); None when X is not read by such a conv first."""
    if _takes_carry(module):
        return module(X, **carries)
    if isinstance(module, torch.nn.Sequential) and len(module) > 0:
        mods = list(module)
        h = _head_with_carry(mods[0], X, **carries)
        if h is not None:
            for mod in mods[1:]:
                h = mod(h)
        return h
    return None


def _cat_channels(a, b):
    """torch.cat((a, b), 1) for 4-D / 5-D activations on the channel-copy kernel."""
    if a.dim() == 4:
        return HF.cat_channels([a.unsqueeze(2), b.unsqueeze(2)]).squeeze(2)
    return HF.cat_channels([a, b])


def _nearest(x, size):
    """F.interpolate(x, size, mode="nearest") for 4-D / 5-D tensors on the resampling kernel."""
    if x.dim() == 4:
        return HF.interpolate_nearest(x.unsqueeze(2), (1, *size)).squeeze(2)
    return HF.interpolate_nearest(x, size)


class SaeBlock(torch.nn.Sequential):
    """``Sequential(conv_block, ConcurrentSqueezeAndExcite)`` of ``conv_type="sae"``
    (unet.py:375-397): same children / keys; the concat operand of a decoder level goes to the conv
    block's first conv as two sources."""

    def forward(self, X, X_cat=None, carry_cat=None):
        h = self[0](X, X_cat=X_cat, carry_cat=carry_cat) if X_cat is not None else self[0](X)
        return self[1](h)


class _DecoderOp(torch.nn.Sequential):
    """``Sequential(conv_block, adn)`` forwarding the concat operand."""

    def forward(self, X, X_cat=None, carry_cat=None):
        mods = list(self)
        if X_cat is not None and isinstance(mods[0], (ConcatConvBlock, SaeBlock)):
            h = mods[0](X, X_cat=X_cat, carry_cat=carry_cat)
        else:
            h = mods[0](X if X_cat is None else _cat_channels(X, X_cat))
        for mod in mods[1:]:
            h = mod(h)
        return h


def _keep_arguments(module, arguments: dict, skip=("self", "parent_class")):
    """Store constructor arguments under their own names (the attribute set the reference's
    classes expose and its wrappers / entry points read)."""
    for name, value in arguments.items():
        if name not in skip and not name.startswith("__"):
            setattr(module, name, value)


class UNet(torch.nn.Module):
    # signature: adell_mri/modules/segmentation/unet.py:43-68
    def __init__(self, spatial_dimensions: int = 2,
                 encoding_operations: List[torch.nn.ModuleList] = None,
                 conv_type: str = "regular", link_type: str = "identity",
                 upscale_type: str = "upsample", interpolation: str = "bilinear",
                 norm_type: str = "batch", dropout_type: str = "dropout", padding: str = "same",
                 dropout_param: float = 0.1, activation_fn: torch.nn.Module = torch.nn.PReLU,
                 in_channels: int = 1, n_classes: int = 2, depth: list = [16, 32, 64],
                 kernel_sizes: list = [3, 3, 3], strides: list = [2, 2, 2],
                 bottleneck_classification: bool = False, skip_conditioning: int = None,
                 feature_conditioning: int = None,
                 feature_conditioning_params: Dict[str, torch.Tensor] = None,
                 deep_supervision: bool = False, parent_class: bool = False,
                 encoder_only: bool = False):
        arguments = dict(locals())
        super().__init__()
        _keep_arguments(self, arguments)

        if self.encoder_only is True or parent_class is False:
            self.get_norm_op()
            self.get_drop_op()
            self.get_conv_op()
            if self.encoding_operations is None:
                self.init_encoder()
            else:
                self.init_encoder_backbone()
        if self.encoder_only is not True and parent_class is False:
            self.init_upscale_ops()
            self.init_link_ops()
            self.init_decoder()
            self.init_final_layer()
            if self.bottleneck_classification is True:
                self.init_bottleneck_classifier()
            if self.feature_conditioning == 0:
                self.feature_conditioning = None
            if self.feature_conditioning is not None:
                self.init_feature_conditioning_operations()

    # ---- operator selection ------------------------------------------------
    def get_norm_op(self):
        if self.norm_type is None:
            self.norm_op = torch.nn.Identity
            return
        self.norm_op = norm_fn_dict[self.norm_type][self.spatial_dimensions]

    def get_drop_op(self):
        if self.dropout_type is None:
            self.drop_op = torch.nn.Identity
        elif self.dropout_type == "dropout":
            self.drop_op = torch.nn.Dropout
        elif self.dropout_type == "uout":
            self.drop_op = UOut

    def get_conv_op(self):
        if self.spatial_dimensions not in (2, 3):
            raise ValueError("spatial_dimensions must be 2 or 3")
        if self.conv_type == "regular":
            self.conv_op_enc = self.conv_block
            self.conv_op_dec = self.conv_block
        elif self.conv_type == "depthwise":
            self.conv_op_enc = self.depthwise_conv_block
            self.conv_op_dec = self.depthwise_conv_block
        elif self.conv_type == "resnet":
            self.conv_op_enc = self.res_block_conv_3d
            self.conv_op_dec = self.conv_block
        elif self.conv_type == "sae":
            self.conv_op_enc = self.sae_block
            self.conv_op_dec = self.sae_block
        elif self.conv_type == "asp":
            self.conv_op_enc = self.asp_block
            self.conv_op_dec = self.sae_block
        else:
            raise NotImplementedError(
                f"conv_type={self.conv_type!r} (spatial_dimensions={self.spatial_dimensions}) "
                "is outside the HIP path built so far")

    @property
    def _conv(self):
        return Conv3d if self.spatial_dimensions == 3 else Conv2d

    def conv_block(self, in_d, out_d, kernel_size, stride=None, padding=None):
        """conv(in->in, k, stride) -> ADN(in) -> conv(in->out, k): unet.py:245-273."""
        padding = 0 if padding is None else padding
        stride = 1 if stride is None else stride
        return ConcatConvBlock(
            self._conv(in_d, in_d, kernel_size, stride, padding),
            self.adn_fn(in_d),
            self._conv(in_d, out_d, kernel_size, 1, padding),
        )

    # the reference names (unet.py:245,260) stay available
    conv_block_2d = conv_block
    conv_block_3d = conv_block

    def depthwise_conv_block(self, in_d, out_d, kernel_size, stride=None, padding=None):
        """conv(in->in, k, stride, groups=in) -> ADN(in) -> conv(in->out, 1): unet.py:276-307. As
        in the reference the 1x1 conv receives the SAME ``padding`` argument: with an integer
        padding p > 0 (every downsampling block: p = k // 2, unet.py:563-567) its output is the
        input grown by p voxels of bias on every side, and the decoder crops the skip tensors."""
        padding = 0 if padding is None else padding
        stride = 1 if stride is None else stride
        dw = DepthwiseConv3dStrided if self.spatial_dimensions == 3 else DepthwiseConv2dStrided
        return ConcatConvBlock(dw(in_d, in_d, kernel_size, stride, padding, groups=in_d),
                               self.adn_fn(in_d), self._conv(in_d, out_d, 1, 1, padding))

    depthwise_conv_block_2d = depthwise_conv_block
    depthwise_conv_block_3d = depthwise_conv_block

    def sae_block(self, in_d, out_d, kernel_size, stride=None, padding=None):
        """conv block -> concurrent (spatial + channel) squeeze-and-excite: unet.py:375-397,
        self_attention.py:40-149. The gate pass is one kernel (``adell_cse_apply``, the BrUNet
        merge kernel with a single branch)."""
        from ..layers.self_attention import (ConcurrentSqueezeAndExcite2d,
                                             ConcurrentSqueezeAndExcite3d)
        cse = ConcurrentSqueezeAndExcite3d if self.spatial_dimensions == 3 else ConcurrentSqueezeAndExcite2d
        return SaeBlock(self.conv_block(in_d, out_d, kernel_size, stride, padding), cse(out_d))

    sae_2d = sae_block
    sae_3d = sae_block

    def asp_block(self, in_d, out_d, kernel_size, stride=None, padding=None):
        """Atrous spatial pyramid with rates (1, 2) and an instance-norm ADN of its own, whatever
        ``norm_type`` says; ``kernel_size`` / ``stride`` / ``padding`` are ignored as in the reference
        (unet.py:399-413) -- an "asp" encoder never downsamples and the decoder crops."""
        from ..layers.adn_fn import get_adn_fn
        from ..layers.multi_resolution import (AtrousSpatialPyramidPooling2d,
                                               AtrousSpatialPyramidPooling3d)
        asp = AtrousSpatialPyramidPooling3d if self.spatial_dimensions == 3 else AtrousSpatialPyramidPooling2d
        return asp(in_d, out_d, [1, 2],
                   get_adn_fn(self.spatial_dimensions, "instance", self.activation_fn, self.dropout_param))

    asp_2d = asp_block
    asp_3d = asp_block

    def res_block_conv_3d(self, in_d, out_d, kernel_size, stride=None, padding=None):
        """ResidualBlock3d / 2d (+ max pooling when strided): unet.py:309-379."""
        nd = self.spatial_dimensions
        res, pool = (ResidualBlock3d, MaxPool3d) if nd == 3 else (ResidualBlock2d, MaxPool2d)
        inter_d = int(in_d) if in_d > 32 else None
        stride = _per_dim(1 if stride is None else stride, nd)
        block = res(in_d, kernel_size, inter_d, out_d, adn_fn=self.adn_fn)
        if any(s > 1 for s in stride):
            padding = _per_dim(0 if padding is None else padding, nd)
            new_padding = [p // 2 if p > s // 2 else p for p, s in zip(padding, stride)]
            return torch.nn.Sequential(block, pool(stride, stride, padding=new_padding))
        return block

    def adn_fn(self, s: int) -> torch.nn.Module:
        return ActDropNorm(in_channels=s, ordering="NDA", norm_fn=self.norm_op,
                           act_fn=self.activation_fn, dropout_fn=self.drop_op,
                           dropout_param=self.dropout_param)

    # ---- builders (same trees as unet.py:415-655) ----------------------------
    def init_upscale_ops(self):
        depths_a = self.depth[:0:-1]
        depths_b = self.depth[-2::-1]
        ops_ = []
        if self.upscale_type == "upsample":   # unet.py:419-443: 1x1 conv, then interpolation
            conv = Conv3d if self.spatial_dimensions == 3 else Conv2d
            for d1, d2, s in zip(depths_a, depths_b, self.strides[::-1][1:]):
                ops_.append(torch.nn.Sequential(conv(d1, d2, 1),
                                                Upsample(scale_factor=s, mode=self.interpolation)))
        elif self.upscale_type == "transpose":
            convt = ConvTranspose3d if self.spatial_dimensions == 3 else ConvTranspose2d
            for d1, d2, s in zip(depths_a, depths_b, self.strides[::-1][1:]):
                s = _per_dim(s, self.spatial_dimensions)
                p = [int(np.maximum(i - 2, 0)) for i in s]
                ops_.append(convt(d1, d2, s, stride=s, padding=p))
        self.upscale_ops = torch.nn.ModuleList(ops_)

    def init_link_ops(self):
        ex = self.skip_conditioning if self.skip_conditioning is not None else 0
        rev_depth = self.depth[-2::-1]
        if self.link_type == "identity":
            self.link_ops = torch.nn.ModuleList([torch.nn.Identity() for _ in self.depth[:-1]])
        elif self.link_type == "conv":
            self.link_ops = torch.nn.ModuleList([
                torch.nn.Sequential(self._conv(d + ex, d, 3, padding=self.padding), self.adn_fn(d))
                for d in rev_depth])
        elif self.link_type == "residual":
            block = ResidualBlock3d if self.spatial_dimensions == 3 else ResidualBlock2d
            self.link_ops = torch.nn.ModuleList([
                block(d + ex, 3, out_channels=d, adn_fn=self.adn_fn) for d in rev_depth])
        elif self.link_type == "attention":     # unet.py:473-481
            from ..layers.self_attention import SelfAttentionBlock
            self.link_ops = torch.nn.ModuleList([
                SelfAttentionBlock(self.spatial_dimensions, d, d, [16, 16, 1])
                for d in self.depth[-2::-1]])
        else:
            raise NotImplementedError(f"link_type={self.link_type!r} is outside the HIP path")

    def init_encoder(self):
        self.encoding_operations = self._build_encoder()

    def _build_encoder(self):
        """One encoder tree (unet.py:531-586; BrUNet builds one per input branch, :999-1041)."""
        encoding_operations = torch.nn.ModuleList([])
        previous_d = self.in_channels
        nd = self.spatial_dimensions
        k = None
        for i in range(len(self.depth) - 1):
            d = self.depth[i]
            k = _per_dim(self.kernel_sizes[i], nd)
            s = _per_dim(self.strides[i], nd)
            p = [int(j // 2) for j in k]
            op = torch.nn.Sequential(
                self.conv_op_enc(previous_d, d, kernel_size=k, stride=1, padding=self.padding),
                self.adn_fn(d))
            op_downsample = torch.nn.Sequential(
                self.conv_op_enc(d, d, kernel_size=k, stride=s, padding=p), self.adn_fn(d))
            encoding_operations.append(torch.nn.ModuleList([op, op_downsample]))
            previous_d = d
        op = torch.nn.Sequential(
            self.conv_op_enc(self.depth[-2], self.depth[-1], kernel_size=k, stride=1,
                             padding=self.padding),
            self.adn_fn(self.depth[-1]))
        encoding_operations.append(torch.nn.ModuleList([op, torch.nn.Identity()]))
        return encoding_operations

    def init_encoder_backbone(self):
        """Every downsampling op becomes MaxPool(kernel=s, stride=s, padding=s//2)
        (unet.py:588-603); the last level keeps Identity."""
        pool = MaxPool3d if self.spatial_dimensions == 3 else MaxPool2d
        for i in range(len(self.encoding_operations)):
            s = np.array(_per_dim(self.strides[i], self.spatial_dimensions))
            self.encoding_operations[i][1] = pool(kernel_size=tuple(int(j) for j in s),
                                                  stride=tuple(int(j) for j in s),
                                                  padding=tuple(int(j) for j in s // 2))
        self.encoding_operations[-1][1] = torch.nn.Identity()

    def init_decoder(self):
        self.decoding_operations = torch.nn.ModuleList([])
        depths = self.depth[-2::-1]
        kernel_sizes = self.kernel_sizes[-2::-1]
        self.deep_supervision_ops = torch.nn.ModuleList([])
        for d, k in zip(depths, kernel_sizes):
            k = _per_dim(k, self.spatial_dimensions)
            op = _DecoderOp(
                self.conv_op_dec(d * 2, d, kernel_size=k, stride=1, padding=self.padding),
                self.adn_fn(d))
            self.decoding_operations.append(op)
            if self.deep_supervision is True:
                self.deep_supervision_ops.append(self.get_ds_final_layer(d))

    def _head(self, d, padding):
        last = torch.nn.Softmax(dim=1) if self.n_classes > 2 else torch.nn.Sigmoid()
        nc = self.n_classes if self.n_classes > 2 else 1
        return torch.nn.Sequential(self._conv(d, d, 3, padding=padding), self.adn_fn(d),
                                   self._conv(d, nc, 1), last)

    def get_final_layer(self, d: int) -> torch.nn.Module:
        return self._head(d, "same")

    def get_ds_final_layer(self, d: int) -> torch.nn.Module:
        return self._head(d, 0)

    def init_final_layer(self):
        self.final_layer = self.get_final_layer(self.depth[0])

    def init_feature_conditioning_operations(self):
        """Tabular features -> per-level channel gates (unet.py:716-740)."""
        depths = self.depth[-2::-1]
        self.feature_conditioning_ops = torch.nn.ModuleList([])
        if self.feature_conditioning_params is not None:
            self.f_mean = torch.nn.parameter.Parameter(
                self.feature_conditioning_params["mean"], requires_grad=False)
            self.f_std = torch.nn.parameter.Parameter(
                self.feature_conditioning_params["std"], requires_grad=False)
        else:
            self.f_mean = torch.nn.parameter.Parameter(
                torch.zeros([self.feature_conditioning]), requires_grad=False)
            self.f_std = torch.nn.parameter.Parameter(
                torch.ones([self.feature_conditioning]), requires_grad=False)
        for d in depths:
            self.feature_conditioning_ops.append(torch.nn.Sequential(
                Linear(self.feature_conditioning, d),
                get_adn_fn(1, "batch", "swish", self.dropout_param)(d),
                Linear(d, d),
                get_adn_fn(1, "batch", "sigmoid", self.dropout_param)(d)))

    def init_bottleneck_classifier(self):
        nc = self.n_classes if self.n_classes > 2 else 1
        self.bottleneck_classifier = Linear(self.depth[-1], nc)   # fp32-MFMA GEMM

    # ---- forward (unet.py:751-843) --------------------------------------------
    def _final(self, layer, X, return_logits):
        mods = list(layer)
        X = _run_rest(mods[:-1], X)
        if return_logits is True:
            return X
        if isinstance(mods[-1], torch.nn.Sigmoid):
            if X.dim() == 4:  # 2-D network: depth-1 volume
                return HF.norm_drop_act(X.unsqueeze(2), act="sigmoid").squeeze(2)
            return HF.norm_drop_act(X, act="sigmoid")
        return HF.channel_softmax(X)   # torch.nn.Softmax(dim=1) of the n_classes > 2 head

    def forward(self, X: torch.Tensor, X_skip_layer: torch.Tensor = None,
                X_feature_conditioning: torch.Tensor = None, return_features=False,
                return_bottleneck=False, return_logits=False):
        if not X.is_cuda:
            raise AdellHipError("adell_mri_amd.UNet runs on MI355X only (no CPU fallback)")
        HF.clear_row_expectations()
        encoding_out, bottleneck, X_skip_layer, X_feature_conditioning = self._encode(
            X, X_skip_layer, X_feature_conditioning)
        if return_bottleneck is True:
            return None, None, bottleneck
        elif self.encoder_only is True:
            return bottleneck
        return self._decode(encoding_out, bottleneck, X_skip_layer, X_feature_conditioning,
                            return_features, return_logits)

    def _encode(self, X, X_skip_layer, X_feature_conditioning):
        """Encoder levels (unet.py:768-788): the per-level outputs, the bottleneck and the two
        conditioning inputs made ready for the decoder (channel axis added / standardised)."""
        if X_skip_layer is not None and X_skip_layer.dim() < X.dim():
            X_skip_layer = X_skip_layer.unsqueeze(1)
        if X_feature_conditioning is not None:   # tiny [B, F] tensor
            X_feature_conditioning = (X_feature_conditioning - self.f_mean) / self.f_std
        encoding_out, curr = [], X
        # skip forks: a level output feeds the downsampling conv and (later) the link op / the
        # decoder's concat. The later reader's gradient is left in a functional.GradCarry and
        # added in the epilogue of the downsampling conv's backward-data kernel.
        for level, downsample in self.encoding_operations:
            curr = level(curr)
            encoding_out.append(curr)
            fork, down = None, None
            if (curr.dim() == 5 and curr.requires_grad and torch.is_grad_enabled()
                    and not ops.FLAGS["no_grad_carry"] and not ops.FLAGS["no_skip_fork"]
                    and not HF.grad_observed(curr)):
                fork = HF.GradCarry()
                down = _head_with_carry(downsample, curr, carry_in=fork)
            if down is None:
                fork, down = None, downsample(curr)
            curr._adell_fork = fork    # for the first later reader of this tensor (one use)
            curr = down
        return encoding_out, curr, X_skip_layer, X_feature_conditioning

    def _run_decoder(self, encoding_out, bottleneck, X_skip_layer, X_feature_conditioning):
        """Decoder levels only (unet.py:790-822): the last feature map and the per-level outputs
        (also semi_supervised_segmentation/unet.py:84-131)."""
        curr = bottleneck
        deep_outputs = []
        for i in range(len(self.decoding_operations)):
            op = self.decoding_operations[i]
            link_op = self.link_ops[i]
            up = self.upscale_ops[i]
            link_in = encoding_out[-i - 2]
            fork = getattr(link_in, "_adell_fork", None)
            link_in._adell_fork = None
            if X_skip_layer is not None:
                S = link_in.shape[2:]
                link_in = _cat_channels(link_in, _nearest(X_skip_layer, S))
                fork = None
            # the link block's output is read by the decoder conv alone (as the second half of its
            # channel concat): its last ADN may write split rows
            link_adn = _last_adn(link_op)
            if (link_adn is not None and X_feature_conditioning is None
                    and isinstance(op, _DecoderOp) and isinstance(op[0], ConcatConvBlock)):
                reader = _first_conv(op[0])
                if reader is not None:
                    HF.expect_rows(link_adn, reader, as_x1=True, producers=(link_op,),
                                   readers=(op,))
            encoded = None
            if fork is not None and isinstance(link_op, ResidualBlock3d):
                encoded, fork = link_op(link_in, fork=fork), None
            elif fork is not None and not isinstance(link_op, torch.nn.Identity):
                encoded, fork = _head_with_carry(link_op, link_in, carry_x0=fork), None
            if encoded is None:
                encoded = link_op(link_in)
            if X_feature_conditioning is not None:   # channel gates, unet.py:803-810
                gates = self.feature_conditioning_ops[i](X_feature_conditioning)
                encoded = HF.scale_per_item_channel(encoded, gates)
            curr = up(curr)
            sh, sh2 = list(curr.shape)[2:], list(encoded.shape)[2:]
            if np.prod(sh) < np.prod(sh2):
                encoded = crop_to_size(encoded, sh)
            if np.prod(sh) > np.prod(sh2):
                curr = crop_to_size(curr, sh2)
            # virtual concat (curr, encoded); identity link: the fork's gradient is the concat's
            # second half
            if fork is not None and encoded is encoding_out[-i - 2] and isinstance(op, _DecoderOp):
                curr = op(curr, X_cat=encoded, carry_cat=fork)
            else:
                # the link op's own output (not an encoder level: those fork) is read by the
                # decoder conv alone
                if not isinstance(link_op, torch.nn.Identity) and encoded is not encoding_out[-i - 2]:
                    encoded = HF.single_use(encoded)
                curr = op(curr, X_cat=encoded)
            deep_outputs.append(curr)
        return curr, deep_outputs

    def _decode(self, encoding_out, bottleneck, X_skip_layer, X_feature_conditioning,
                return_features, return_logits):
        """Decoder, head, bottleneck classifier and deep supervision (unet.py:790-843; the same
        code closes BrUNet.forward, :1209-1253)."""
        head_only = return_features is not True and self.deep_supervision is not True
        if head_only and len(self.decoding_operations) > 0:
            # the last decoder op's ADN output is read by the head's first conv alone
            adn, reader = _last_adn(self.decoding_operations[-1]), _first_conv(self.final_layer)
            if adn is not None and reader is not None:
                HF.expect_rows(adn, reader, producers=(self.decoding_operations[-1],),
                               readers=(self.final_layer,))
        curr, deep_outputs = self._run_decoder(encoding_out, bottleneck, X_skip_layer,
                                               X_feature_conditioning)
        if head_only:
            curr = HF.single_use(curr)      # the head's first conv is its only reader
        head = self._final(self.final_layer, curr, return_logits)
        return self._outputs(head, curr, bottleneck, deep_outputs, return_features)

    def _outputs(self, head, final_features, bottleneck, deep_outputs, return_features):
        """What ``forward`` returns once the head has run (unet.py:824-843): (head, features,
        bottleneck) on request, else (head, bottleneck class logits or None[, auxiliary heads])."""
        if return_features is True:
            return head, final_features, bottleneck
        bn_out = None
        if self.bottleneck_classification is True:   # global max over space, then the GEMM
            bn_out = self.bottleneck_classifier(HF.channel_max(bottleneck))
        if self.deep_supervision is True:
            aux = [self._final(op, o, False) for op, o in zip(self.deep_supervision_ops, deep_outputs)]
            return head, bn_out, aux
        return head, bn_out


class BrUNet(UNet):
    """Multi-branch U-Net (mirror of adell_mri/modules/segmentation/unet.py:846-1253): one encoder
    per input, the branches merged at every skip level and at the bottleneck by concurrent
    squeeze-and-excite gates, summed and divided by the summed branch weights; decoder, head,
    conditioning and deep supervision as in ``UNet``. Same constructor arguments, attribute names
    and ``state_dict`` keys (``encoders.B.L.{0,1}...``, ``merge_ops.L.B.{spatial,channel}...``).

    On the HIP path the merge of one level is ``n_input_branches`` passes of one kernel
    (``acc + x * (s + c) / w_sum``, functional.cse_apply) instead of, per branch, two gated copies,
    an add, a division and the running sum."""

    def __init__(self, spatial_dimensions: int = 2, n_input_branches: int = 1,
                 encoders: List[torch.nn.ModuleList] = None, conv_type: str = "regular",
                 link_type: str = "identity", upscale_type: str = "upsample",
                 interpolation: str = "bilinear", norm_type: str = "batch",
                 dropout_type: str = "dropout", padding: str = "same", dropout_param: float = 0.1,
                 activation_fn: torch.nn.Module = torch.nn.PReLU, in_channels: int = 1,
                 n_classes: int = 2, depth: list = [16, 32, 64], kernel_sizes: list = [3, 3, 3],
                 strides: list = [2, 2, 2], bottleneck_classification: bool = False,
                 skip_conditioning: int = None, feature_conditioning: int = None,
                 feature_conditioning_params: Dict[str, torch.Tensor] = None,
                 deep_supervision: bool = False, encoder_only: bool = False):
        super().__init__(parent_class=True)
        self.spatial_dimensions = spatial_dimensions
        self.n_input_branches = n_input_branches
        self.encoders = encoders
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
        self.strides = strides
        self.bottleneck_classification = bottleneck_classification
        self.skip_conditioning = skip_conditioning
        self.feature_conditioning = feature_conditioning
        self.feature_conditioning_params = feature_conditioning_params
        self.deep_supervision = deep_supervision
        self.encoder_only = encoder_only

        self.get_norm_op()
        self.get_drop_op()
        self.get_conv_op()
        if self.encoders is None:
            self.init_encoders()
        else:
            self.init_backbone_encoders()
        self.init_merge_ops()

        if self.encoder_only is not True:
            self.init_upscale_ops()
            self.init_link_ops()
            self.init_decoder()
            self.init_final_layer()
            if self.bottleneck_classification is True:
                self.init_bottleneck_classifier()
            if self.feature_conditioning is not None:
                self.init_feature_conditioning_operations()

    def init_encoders(self):
        self.encoders = torch.nn.ModuleList(
            [self._build_encoder() for _ in range(self.n_input_branches)])

    def init_backbone_encoders(self):
        """Backbone encoders: max pooling (kernel k, stride 2, padding k // 2) between levels
        (unet.py:1043-1062)."""
        assert len(self.encoders) == self.n_input_branches, \
            "n_input_branches and len(self.encoders) must be the same"
        pool = MaxPool3d if self.spatial_dimensions == 3 else MaxPool2d
        for i in range(self.n_input_branches):
            for j in range(len(self.encoders[i])):
                self.encoders[i][j][1] = pool(self.kernel_sizes[j], 2, self.kernel_sizes[j] // 2)
            self.encoders[i][-1][1] = torch.nn.Identity()

    def init_merge_ops(self):
        from ..layers.self_attention import (ConcurrentSqueezeAndExcite2d,
                                             ConcurrentSqueezeAndExcite3d)
        cse = (ConcurrentSqueezeAndExcite3d if self.spatial_dimensions == 3
               else ConcurrentSqueezeAndExcite2d)
        D = [self.depth[-1]] if self.encoder_only is True else self.depth
        self.merge_ops = torch.nn.ModuleList([
            torch.nn.ModuleList([cse(d) for _ in range(self.n_input_branches)]) for d in D])

    @staticmethod
    def fix_input(X: List[List[torch.Tensor]]):
        """Lists of per-item tensors with ``None`` for a missing input -> stacked batches (zeros
        where missing) and the 0 / 1 branch weights (unet.py:1094-1111)."""
        shapes = []
        for items in X:
            found = {tuple(x.shape) for x in items if x is not None}
            assert len(found) == 1, "all tensors must have the same shape"
            shapes.append(found.pop())
        weights = [torch.ones(len(X[0])) for _ in X]
        batches = []
        for i, items in enumerate(X):
            like = next(x for x in items if x is not None)
            row = []
            for j, x in enumerate(items):
                if x is None:
                    x = torch.zeros(shapes[i], dtype=like.dtype, device=like.device)
                    weights[i][j] = 0.0
                row.append(x)
            batches.append(torch.stack(row, 0))
        return batches, weights

    @staticmethod
    def _weighted(t, w):
        """t * w[n] (per batch item) on the per-(item, channel) scale kernel."""
        if w is None:
            return t
        return HF.scale_per_item_channel(t, w[:, None].expand(t.shape[0], t.shape[1]).contiguous())

    def forward(self, X: List[torch.Tensor], X_weights: List[torch.Tensor] = None,
                X_skip_layer: torch.Tensor = None, X_feature_conditioning: torch.Tensor = None,
                return_features=False, return_bottleneck=False, return_logits=False):
        if not all(x.is_cuda for x in X):
            raise AdellHipError("adell_mri_amd.BrUNet runs on MI355X only (no CPU fallback)")
        dev, B = X[0].device, X[0].shape[0]
        if X_weights is not None:
            assert len(X) == len(X_weights), "X and X_weights should have identical length"
            assert all(x.shape[0] == xw.shape[0] for x, xw in zip(X, X_weights)), \
                "The elements of X and X_weights should have identical batch sizes"
            X_weights = [xw.to(dev, torch.float32) for xw in X_weights]
            inv = 1.0 / sum(X_weights)                      # [B]: tiny host-launched divide
        else:   # the reference multiplies by ones and divides by the branch count
            inv = torch.full((B,), 1.0 / len(X), device=dev)
        if X_skip_layer is not None and len(X_skip_layer.shape) < len(X[0].shape):
            X_skip_layer = X_skip_layer.unsqueeze(1)
        if X_feature_conditioning is not None:
            X_feature_conditioning = (X_feature_conditioning - self.f_mean) / self.f_std

        pre_merge = [[] for _ in self.encoders]
        bottleneck_pre = []
        for i in range(self.n_input_branches):
            curr = X[i]
            w = None if X_weights is None else X_weights[i]
            for op, op_ds in self.encoders[i]:
                curr = op(curr)
                pre_merge[i].append(self._weighted(curr, w))
                curr = op_ds(curr)
            bottleneck_pre.append(self._weighted(curr, w))

        def merge(ops_, tensors):
            acc = None
            for j in range(self.n_input_branches):
                acc = ops_[j](tensors[j], inv, acc)
            return acc

        bottleneck = merge(self.merge_ops[-1], bottleneck_pre)
        if self.encoder_only is True:
            return bottleneck
        if return_bottleneck is True:
            return None, None, bottleneck
        encoding_out = [merge(self.merge_ops[i], tensors)
                        for i, tensors in enumerate(zip(*pre_merge))]
        return self._decode(encoding_out, bottleneck, X_skip_layer, X_feature_conditioning,
                            return_features, return_logits)
