"""SSL backbone -> U-Net encoder hand-off (train.py:672-734): derived U-Net configuration,
checkpoint loading and freezing. Host logic only (module construction needs no GPU)."""
import torch

from adell_mri_amd.modules.layers.adn_fn import get_adn_fn
from adell_mri_amd.modules.layers.res_net import ResNet
from adell_mri_amd.utils.handoff import unet_encoder_from_ssl


def _ssl_cfg():
    # sample_configs/ssl-resnet.yaml:5-6 scaled down
    return dict(backbone_args=dict(spatial_dim=3, in_channels=2,
                                   structure=[[8, 8, 5, 1], [16, 16, 3, 1], [32, 32, 3, 1]],
                                   maxpool_structure=[[2, 2, 1], [2, 2, 2], [2, 2, 2]],
                                   adn_fn=get_adn_fn(3, "batch", "swish", 0.0)),
                projection_head_args=dict(in_channels=32, structure=[16, 8],
                                          adn_fn=get_adn_fn(1, "batch", "swish", 0.0)),
                learning_rate=0.1, weight_decay=0.2, batch_size=3)


def test_unet_configuration_is_derived_from_the_backbone():
    cfg, enc, nets = unet_encoder_from_ssl(dict(spatial_dimensions=3, padding=1), _ssl_cfg())
    assert cfg["depth"] == [8, 8, 16, 32]                 # stem width, then the stage widths
    assert cfg["kernel_sizes"] == [3, 3, 3, 3]
    assert cfg["strides"] == [2, [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    assert len(enc) == 1 and len(enc[0]) == 4             # (stem, pool) + one pair per stage
    assert all(len(pair) == 2 for pair in enc[0])
    assert all(p.requires_grad for p in nets[0].parameters())


def test_checkpoint_is_loaded_and_encoder_frozen_when_lr_encoder_is_zero():
    donor = ResNet(**{k: v for k, v in _ssl_cfg().items()
                      if k not in ("learning_rate", "weight_decay", "batch_size")})
    with torch.no_grad():
        for p in donor.parameters():
            p.fill_(0.125)
    sd = donor.state_dict()
    sd["not.in.the.model"] = torch.zeros(1)               # strict=False tolerates extras
    cfg, enc, nets = unet_encoder_from_ssl(dict(spatial_dimensions=3), _ssl_cfg(), [sd],
                                           lr_encoder=0.0)
    bb = nets[0].backbone
    enc_params = [p for op in [bb.input_layer, *bb.operations] for p in op.parameters()]
    assert enc_params and all(float(p.detach().flatten()[0]) == 0.125 for p in enc_params)
    assert not any(p.requires_grad for p in enc_params)
    # a non-zero encoder learning rate keeps the encoder trainable
    _, _, nets2 = unet_encoder_from_ssl(dict(spatial_dimensions=3), _ssl_cfg(), [sd],
                                        lr_encoder=1e-4)
    assert all(p.requires_grad for p in nets2[0].backbone.parameters())


def _frozen_encoder_unet(device=None):
    """The U-Net the reference builds after `--encoder_checkpoint ... --lr_encoder 0.0`
    (train.py:718-724): encoder from the SSL backbone, frozen; its parameters still form the first
    optimiser group (pl.py:553-561), which then has no trainable member."""
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.pl import UNetPL

    donor = ResNet(**{k: v for k, v in _ssl_cfg().items()
                      if k not in ("learning_rate", "weight_decay", "batch_size")})
    cfg, enc, nets = unet_encoder_from_ssl(dict(spatial_dimensions=3, padding=1), _ssl_cfg(),
                                           [donor.state_dict()], lr_encoder=0.0)
    kw = dict(cfg, encoding_operations=enc[0], in_channels=2, n_classes=2, norm_type="instance",
              upscale_type="transpose", link_type="identity", dropout_param=0.0,
              activation_fn=activation_factory["swish"])
    net = UNetPL(image_key="image", label_key="mask", learning_rate=1e-2, weight_decay=5e-3,
                 lr_encoder=0.0, **kw)
    return net if device is None else net.to(device)


def test_optimizer_accepts_a_group_of_frozen_parameters():
    """every fused optimiser raised on a parameter group without a trainable
    parameter; torch.optim accepts it."""
    from adell_mri_amd.optim import FusedAdamW, FusedSGD

    frozen = [torch.nn.Parameter(torch.ones(3), requires_grad=False)]
    live = [torch.nn.Parameter(torch.ones(5))]
    for cls in (FusedSGD, FusedAdamW):
        opt = cls([{"params": frozen, "lr": 0.0}, {"params": live}], lr=0.1)
        empty, full = opt.flat_groups
        assert empty.data.numel() == 0 and empty.params == [] and full.data.numel() == 8
        opt.zero_grad()
        sd = opt.state_dict()
        assert [g["params"] for g in sd["param_groups"]] == [[0], [1]] and sd["state"] == {}
        opt.load_state_dict(sd)
    net = _frozen_encoder_unet()
    opt = net.configure_optimizers()["optimizer"]
    assert opt.param_groups[0]["lr"] == 0.0 and len(opt.param_groups[0]["params"]) > 0
    assert opt.flat_groups[0].params == [] and len(opt.flat_groups[1].params) > 0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_train_step_after_frozen_encoder_handoff():
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    dev = torch.device("cuda:0")
    net = _frozen_encoder_unet(dev).train()
    net.loss_fn = lambda p, y: torch.nn.functional.binary_cross_entropy(p, y)
    opt = net.configure_optimizers()["optimizer"]
    enc0 = {k: p.detach().clone() for k, p in net.named_parameters() if not p.requires_grad}
    dec0 = {k: p.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    assert enc0 and dec0
    runner = StepRunner(net, opt, GradSync(opt))
    g = torch.Generator().manual_seed(0)
    x = torch.rand((1, 2, 32, 32, 16), generator=g).to(dev)
    y = (torch.rand((1, 1, 32, 32, 16), generator=g) > 0.8).float().to(dev)
    loss = float(runner.train_step({"image": x, "mask": y}))
    assert loss == loss and loss > 0
    now = dict(net.named_parameters())
    assert all(torch.equal(now[k], v) for k, v in enc0.items())          # frozen: untouched
    assert any(not torch.equal(now[k], v) for k, v in dec0.items())       # the rest stepped
