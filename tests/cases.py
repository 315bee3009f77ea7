"""Constructor kwargs of the golden U-Net cases (same as oracle/make_golden.py)."""
import numpy as np
UNET_CASES = {
    "unet3d_cfg2_tiny": dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                             upscale_type="transpose", norm_type="instance", padding=1,
                             dropout_param=0.15, activation_fn="swish", in_channels=2,
                             n_classes=2, depth=[4, 4, 8, 16, 32], kernel_sizes=[3] * 5,
                             strides=[2] * 5),
    "unet3d_cfg2_small": dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                              upscale_type="transpose", norm_type="instance", padding=1,
                              dropout_param=0.15, activation_fn="swish", in_channels=2,
                              n_classes=2, depth=[8, 8, 16], kernel_sizes=[3] * 3,
                              strides=[2] * 3),
    "unet3d_identity_links_relu": dict(spatial_dimensions=3, conv_type="regular",
                                       link_type="identity", upscale_type="transpose",
                                       norm_type="instance", padding=1, dropout_param=0.0,
                                       activation_fn="relu", in_channels=1, n_classes=2,
                                       depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
    "unet3d_conv_links_gelu": dict(spatial_dimensions=3, conv_type="regular", link_type="conv",
                                   upscale_type="transpose", norm_type="instance", padding=1,
                                   dropout_param=0.0, activation_fn="gelu", in_channels=3,
                                   n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3,
                                   strides=[2] * 3),
}

# conv_type="depthwise" (unet.py:276-307); fixtures: `python oracle/make_golden.py depthwise`
DEPTHWISE_CASES = {
    "unet3d_depthwise": dict(spatial_dimensions=3, conv_type="depthwise", link_type="identity",
                             upscale_type="transpose", norm_type="instance", padding="same",
                             dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                             depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
    "unet2d_depthwise": dict(spatial_dimensions=2, conv_type="depthwise", link_type="identity",
                             upscale_type="transpose", norm_type="instance", padding="same",
                             dropout_param=0.0, activation_fn="relu", in_channels=1, n_classes=2,
                             depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
}

# conv_type="sae" (unet.py:375-397: conv block + concurrent squeeze-and-excite); fixtures:
# `python oracle/make_golden.py sae`
SAE_CASES = {
    "unet3d_sae": dict(spatial_dimensions=3, conv_type="sae", link_type="identity",
                       upscale_type="transpose", norm_type="instance", padding=1,
                       dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                       depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
    "unet2d_sae": dict(spatial_dimensions=2, conv_type="sae", link_type="conv",
                       upscale_type="transpose", norm_type="instance", padding=1,
                       dropout_param=0.0, activation_fn="relu", in_channels=1, n_classes=2,
                       depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
}

# conv_type="asp" (unet.py:399-413: atrous pyramid encoder ops that ignore stride, "sae" decoder ops);
# 3-D only -- the reference's 2-D pyramid cannot be constructed (standard_blocks.py:78 reads
# self.paddign); fixtures: `python oracle/make_golden.py asp`
ASP_CASES = {
    "unet3d_asp": dict(spatial_dimensions=3, conv_type="asp", link_type="identity",
                       upscale_type="transpose", norm_type="instance", padding=1,
                       dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                       depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
}

# link_type="attention" (unet.py:473-481); fixtures: `python oracle/make_golden.py attention`
ATTENTION_LINK_CASES = {
    "unet3d_attention_links": dict(spatial_dimensions=3, conv_type="regular", link_type="attention",
                                   upscale_type="transpose", norm_type="instance", padding="same",
                                   dropout_param=0.0, activation_fn="swish", in_channels=1,
                                   n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3,
                                   strides=[2] * 3),
    "unet2d_attention_links": dict(spatial_dimensions=2, conv_type="regular", link_type="attention",
                                   upscale_type="transpose", norm_type="instance", padding="same",
                                   dropout_param=0.0, activation_fn="gelu", in_channels=2,
                                   n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3,
                                   strides=[2] * 3),
}

UNETR_CASES = {
    "unetr3d_small": dict(image_size=[32, 32, 32], patch_size=[8, 8, 8], number_of_blocks=4,
                          return_at=[1, 2], embedding_size=64, attention_dim=64, hidden_dim=64,
                          n_heads=4, mlp_structure=[128], spatial_dimensions=3,
                          link_type="identity", upscale_type="transpose", norm_type="instance",
                          padding=1, dropout_param=0.0, activation_fn="swish", in_channels=1,
                          n_classes=2, depth=[8, 16, 32], kernel_sizes=[3, 3, 3]),
}

UNETR_CASES["unetr3d_feature_cond"] = dict(
    image_size=[16, 16, 16], patch_size=[4, 4, 4], number_of_blocks=2, return_at=[1],
    embedding_size=32, attention_dim=32, hidden_dim=32, n_heads=2, mlp_structure=[64],
    spatial_dimensions=3, link_type="identity", upscale_type="transpose", norm_type="instance",
    padding=1, dropout_param=0.0, activation_fn="swish", in_channels=1, n_classes=2,
    depth=[8, 16], kernel_sizes=[3, 3], feature_conditioning=5)
UNETR_CASES["unetr2d_small"] = dict(
    image_size=[32, 48], patch_size=[8, 8], number_of_blocks=4, return_at=[1, 2],
    embedding_size=64, attention_dim=64, hidden_dim=64, n_heads=4, mlp_structure=[128],
    spatial_dimensions=2, link_type="identity", upscale_type="transpose", norm_type="instance",
    padding=1, dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
    depth=[8, 16, 32], kernel_sizes=[3, 3, 3])

UNETPP_CASES = {
    "unetpp3d_small": dict(spatial_dimensions=3, conv_type="regular", upscale_type="transpose",
                           norm_type="instance", padding=1, dropout_param=0.0,
                           activation_fn="swish", in_channels=2, n_classes=2,
                           depth=[8, 8, 16, 32], kernel_sizes=[3] * 4, strides=[2] * 4),
    "unetpp2d_small": dict(spatial_dimensions=2, conv_type="regular", upscale_type="transpose",
                           norm_type="instance", padding=1, dropout_param=0.0,
                           activation_fn="swish", in_channels=1, n_classes=2, depth=[8, 8, 16],
                           kernel_sizes=[3] * 3, strides=[2] * 3),
}


SWIN_CASES = {
    "swinunet3d_small": dict(image_size=[32, 32, 16], patch_size=[4, 4, 4], window_size=[8, 8, 8],
                             shift_sizes=[[0, 1], [0, 1], [0, 1]], embedding_size=[16, 32, 64],
                             n_heads=4, dropout_rate=0.0, embed_method="convolutional",
                             mlp_structure=4.0, spatial_dimensions=3, conv_type="regular",
                             link_type="conv", upscale_type="transpose", norm_type="instance",
                             padding="same", dropout_param=0.0, activation_fn="leaky_relu",
                             in_channels=2, n_classes=2, depth=[8, 16, 32],
                             kernel_sizes=[3, 3, 3], strides=[[2, 2, 1], 2, 2]),
    "swinunet2d_small": dict(image_size=[32, 64], patch_size=[4, 4], window_size=[8, 8],
                             shift_sizes=[[0, 1], [0, 1], [0, 1]], embedding_size=[16, 32, 64],
                             n_heads=4, dropout_rate=0.0, embed_method="linear",
                             mlp_structure=4.0, spatial_dimensions=2, conv_type="regular",
                             link_type="conv", upscale_type="transpose", norm_type="instance",
                             padding="same", dropout_param=0.0, activation_fn="leaky_relu",
                             in_channels=2, n_classes=2, depth=[8, 16, 32],
                             kernel_sizes=[3, 3, 3], strides=[[2, 1], 2, 2]),
}


def oracle_cfg(kw):
    return dict(depth=kw["depth"], kernel_sizes=kw["kernel_sizes"], strides=kw["strides"],
                padding=kw["padding"], norm_type=kw["norm_type"], activation=kw["activation_fn"],
                link_type=kw["link_type"], n_classes=kw["n_classes"],
                dropout_param=kw["dropout_param"])


def grad_rel_err(g, k, got):
    """Relative error of a parameter gradient. Biases that feed a normalisation have a
    mathematically zero gradient (the reference's value is rounding noise), so the
    scale is floored by 10% of the sibling weight's gradient magnitude."""
    ref = g["grad:" + k]
    scale = np.abs(ref).max()
    if k.endswith(".bias") and ("grad:" + k[:-5] + ".weight") in g.files:
        scale = max(scale, 1e-1 * np.abs(g["grad:" + k[:-5] + ".weight"]).max())
    # the same holds for a weight whose output is scale-invariant (a 1-input-channel 1x1 conv in
    # front of an instance norm: gradient 1e-9 where its neighbours have 1e-2): nothing below
    # 5e-4 of the largest gradient of the network is compared relative to itself
    top = max(float(np.abs(g[f]).max()) for f in g.files if f.startswith("grad:"))
    scale = max(scale, 5e-4 * top)
    return float(np.abs(got - ref).max() / (scale + 1e-12))


# multi-branch U-Net fixtures (oracle/make_golden.py BRUNET_CASES): constructor kwargs with the
# activation by name, per-branch input shape
BRUNET_CASES = {
    "brunet3d_two_branch": (dict(spatial_dimensions=3, n_input_branches=2, depth=[8, 16, 32],
                                 upscale_type="transpose", padding=1, strides=[2, 2, 2],
                                 kernel_sizes=[3, 3, 3], conv_type="regular",
                                 link_type="identity", norm_type="instance",
                                 activation_fn="swish", dropout_param=0.0, in_channels=1),
                            (2, 1, 16, 24, 16)),
    "brunet2d_missing_inputs": (dict(spatial_dimensions=2, n_input_branches=2, depth=[8, 16, 32],
                                     upscale_type="transpose", padding=1, strides=[2, 2, 2],
                                     kernel_sizes=[3, 3, 3], conv_type="regular",
                                     link_type="conv", norm_type="instance",
                                     activation_fn="swish", dropout_param=0.0, in_channels=2),
                                (3, 2, 32, 40)),
}

SEMISL_CASES = {
    "unet3d_semisl": dict(spatial_dimensions=3, conv_type="regular", link_type="residual",
                          upscale_type="transpose", norm_type="instance", padding=1,
                          dropout_param=0.0, activation_fn="swish", in_channels=2, n_classes=2,
                          depth=[8, 16, 32], kernel_sizes=[3] * 3, strides=[2] * 3),
    "unet2d_semisl": dict(spatial_dimensions=2, conv_type="regular", link_type="identity",
                          upscale_type="transpose", norm_type="instance", padding=1,
                          dropout_param=0.0, activation_fn="swish", in_channels=1, n_classes=2,
                          depth=[8, 16], kernel_sizes=[3] * 2, strides=[2] * 2),
}


# ---- whole-volume inference operators (utils/inference.py:262-990) ---------------------------------
def inference_net(n_out):
    """A closed-form 'network' for the inference fixtures (tests/golden/inference_ops.npz, written
    by oracle/make_golden_inference.py from the REAL reference operators): it depends on the
    position inside the patch (ramps) and on the patch as a whole (its mean), so overlapping,
    edge-adjusted and flipped windows all give different values. Takes a tensor or a dict with
    "image"; [B, C, *spatial] -> [B, n_out, *spatial]."""
    import torch

    def net(X):
        x = X["image"] if isinstance(X, dict) else X
        nd = x.dim() - 2
        ramp = 0.0
        for a in range(nd):
            shape = [1] * x.dim()
            shape[2 + a] = x.shape[2 + a]
            ramp = ramp + (a + 1) * torch.linspace(0.0, 1.0, x.shape[2 + a]).view(shape)
        mean = x.mean(dim=tuple(range(1, x.dim())), keepdim=True)
        s = x.sum(1, keepdim=True)
        return torch.cat([torch.sigmoid(0.5 * (k + 1) * s + 0.3 * ramp - mean)
                          for k in range(n_out)], 1)

    return net


INFERENCE_CASES = {
    # name: (input shape, kind, arguments)
    "sw3d_ragged_b1": ((1, 2, 20, 18, 14), "sliding", dict(window=(8, 8, 8), stride=(5, 6, 4),
                                                          n_classes=1, batch=1)),
    "sw3d_ragged_b3": ((1, 2, 20, 18, 14), "sliding", dict(window=(8, 8, 8), stride=(5, 6, 4),
                                                          n_classes=1, batch=3)),
    "sw3d_dict_2class": ((1, 3, 17, 16, 12), "sliding", dict(window=(8, 8, 8), stride=(6, 8, 5),
                                                            n_classes=2, batch=2, as_dict=True)),
    "sw3d_default_stride": ((1, 1, 16, 24, 16), "sliding", dict(window=(8, 8, 8), stride=None,
                                                               n_classes=1, batch=4)),
    "sw2d_overlap": ((1, 1, 30, 26), "sliding", dict(window=(16, 16), stride=(8, 8), n_classes=1,
                                                    batch=1)),
    "flip_tensor": ((1, 2, 6, 7, 8), "flip", dict(flips=[(2,), (4,), (2, 3)], n_out=1)),
    "flip_dict_keys": ((1, 2, 6, 7, 8), "flip", dict(flips=[(2,), (3, 4)], n_out=2, as_dict=True,
                                                   flip_keys=["image"])),
    "seg_inference_flip": ((1, 2, 20, 18, 14), "segmentation",
                           dict(window=[8, 8, 8], stride=0.5, n_classes=2, batch=2, flip=True)),
}
