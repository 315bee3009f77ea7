"""Weight gradients on a second HIP stream (functional.side_run): the bookkeeping that decides when
that is safe (CPU), and on the GPU that the training step is bit-identical with the branch on and
off, that a weight used twice in one graph or accumulating into an existing .grad stays on the main
stream, and that nothing reads a gradient before the join."""
import pytest
import torch

from adell_mri_amd import functional as HF


def test_use_counting_decides_when_the_side_stream_is_safe(monkeypatch):
    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", True)
    w = torch.nn.Parameter(torch.zeros(4, 4))
    b = torch.nn.Parameter(torch.zeros(4))
    # opt-in: only parameters owned by optim.FlatParameters (whose collect / zero_grad join the side
    # stream before a gradient is read) may have their gradients produced off the main stream
    HF._note_use(w)
    with torch.no_grad():
        assert not HF._side_ok(w, b, None)      # a plain parameter (torch optimiser, DDP, ...)
    w._adell_flat = b._adell_flat = True        # what FlatParameters.__init__ sets
    HF._note_use(w)
    with torch.no_grad():
        b._adell_flat = False
        assert not HF._side_ok(w, b)            # every parameter of the node must qualify
        b._adell_flat = True
    with torch.no_grad():                       # (a backward pass runs with grad mode off)
        HF._note_use(w)                         # no graph is being built: not counted
        assert getattr(w, "_adell_uses", 0) == 0
    HF._note_use(w)
    with torch.no_grad():
        assert HF._side_ok(w, b, None)
        assert w._adell_uses == 0
        assert not HF._side_ok(w, b)            # a node nobody announced
    # two nodes of one graph: neither may leave the main stream (the engine adds their gradients)
    HF._note_use(w)
    HF._note_use(w)
    with torch.no_grad():
        assert not HF._side_ok(w, b)
        assert not HF._side_ok(w, b)
    HF._note_use(w)
    with torch.no_grad():
        assert HF._side_ok(w, b)                # the flag clears once both are done
    # an existing gradient (accumulation), a tensor hook, a foreign post-accumulate hook
    for spoil in ("grad", "hook", "post"):
        w2 = torch.nn.Parameter(torch.zeros(4, 4))
        w2._adell_flat = True
        HF._note_use(w2)
        if spoil == "grad":
            w2.grad = torch.zeros(4, 4)
        elif spoil == "hook":
            w2.register_hook(lambda g: g)
        else:
            w2.register_post_accumulate_grad_hook(lambda p: None)
        with torch.no_grad():
            assert not HF._side_ok(w2, b), spoil
    # create_graph: the backward itself is recorded
    HF._note_use(w)
    assert not HF._side_ok(w, b)
    # forgotten graphs are dropped at the next zero_grad
    HF._note_use(w)
    HF._note_use(w)
    HF.reset_uses([w])
    assert w._adell_uses == 0 and not w._adell_multi
    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", False)
    HF._note_use(w)
    with torch.no_grad():
        assert not HF._side_ok(w, b)
    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", True)
    # a forward pass under torch DistributedDataParallel marks its weights: main stream
    monkeypatch.setattr(HF, "_inside_torch_ddp", lambda: True)
    HF._note_use(w)
    with torch.no_grad():
        assert not HF._side_ok(w, b)
    monkeypatch.setattr(HF, "_inside_torch_ddp", lambda: False)
    HF._note_use(w)
    with torch.no_grad():
        assert HF._side_ok(w, b)
    # with a process group up, only parameters whose exchange goes through parallel.GradSync
    monkeypatch.setattr(torch.distributed, "is_initialized", lambda: True)
    HF._note_use(w)
    with torch.no_grad():
        assert not HF._side_ok(w, b)
    w._adell_gradsync = b._adell_gradsync = True
    HF._note_use(w)
    with torch.no_grad():
        assert HF._side_ok(w, b)


def _small_unet(cuda, seed=0):
    from adell_mri_amd.modules.layers.adn_fn import activation_factory
    from adell_mri_amd.modules.segmentation.unet import UNet

    torch.manual_seed(seed)
    return UNet(spatial_dimensions=3, upscale_type="transpose", norm_type="instance", padding=1,
                dropout_param=0.1, activation_fn=activation_factory["swish"], in_channels=2,
                n_classes=2, depth=[16, 32, 64], kernel_sizes=[3, 3, 3],
                strides=[2, 2, 2]).to(cuda).train()


def _two_steps(cuda, monkeypatch, on):
    import itertools

    from adell_mri_amd.optim import FusedSGD
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", on)
    monkeypatch.setattr(HF, "_dropout_counter", itertools.count(500))
    net = _small_unet(cuda)
    opt = FusedSGD(net.parameters(), lr=0.05, momentum=0.9, nesterov=True)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 2, 32, 32, 32, generator=g).to(cuda)
    t = (torch.rand(2, 1, 32, 32, 32, generator=g) > 0.5).float().to(cuda)

    class _Mod(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net = net

        def training_step(self, batch, idx):
            out = self.net(batch[0])
            prob = out[0] if isinstance(out, tuple) else out
            return ((prob - batch[1]) ** 2).mean()

    mod = _Mod()
    runner = StepRunner(mod, opt, GradSync(opt))
    losses = [float(runner.train_step((x, t)).detach()) for _ in range(2)]
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in net.parameters()]


@pytest.mark.gpu
def test_training_steps_are_bit_identical_with_the_side_stream(cuda, monkeypatch):
    calls = []
    real = HF.side_run
    monkeypatch.setattr(HF, "side_run", lambda fn, reads: calls.append(1) or real(fn, reads))
    l_on, p_on = _two_steps(cuda, monkeypatch, True)
    assert len(calls) >= 20, "the weight gradients did not go to the side stream"
    n = len(calls)
    l_off, p_off = _two_steps(cuda, monkeypatch, False)
    assert len(calls) == n
    assert l_on == l_off
    for a, b in zip(p_on, p_off):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_shared_weight_and_accumulation_stay_on_the_main_stream(cuda, monkeypatch):
    from adell_mri_amd.modules.layers.conv import Conv3d

    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", True)
    calls = []
    real = HF.side_run
    monkeypatch.setattr(HF, "side_run", lambda fn, reads: calls.append(1) or real(fn, reads))
    torch.manual_seed(1)
    conv = Conv3d(16, 16, 3, padding=1).to(cuda)
    for p in conv.parameters():
        p._adell_flat = True     # (as optim.FlatParameters marks its parameters; the test joins by
                                 # the end-of-backward callback)
    x = torch.randn(1, 16, 8, 8, 8, device=cuda)
    # one use: side stream; the gradient is complete when backward() returns (engine callback)
    conv(x).sum().backward()
    assert len(calls) == 1
    g1 = conv.weight.grad.clone()
    b1 = conv.bias.grad.clone()
    # accumulation into the existing .grad: main stream, and the sum is right
    conv(x).sum().backward()
    assert len(calls) == 1
    assert torch.allclose(conv.weight.grad, 2 * g1, rtol=1e-6, atol=1e-6)
    assert torch.allclose(conv.bias.grad, 2 * b1, rtol=1e-6, atol=1e-6)
    # the same weight twice in one graph: main stream for both nodes
    conv.zero_grad(set_to_none=True)
    HF.reset_uses(conv.parameters())
    (conv(x).sum() + conv(2 * x).sum()).backward()
    assert len(calls) == 1
    assert torch.allclose(conv.weight.grad, 3 * g1, rtol=1e-5, atol=1e-5)
    # and a single use afterwards goes back to the side stream
    conv.zero_grad(set_to_none=True)
    conv(x).sum().backward()
    assert len(calls) == 2
    assert torch.equal(conv.weight.grad, g1)


@pytest.mark.gpu
def test_forward_on_a_non_default_stream_is_joined_on_that_stream(cuda, monkeypatch):
    """The backward nodes run on the stream of their forward; the gradient must be complete for
    work queued on THAT stream after backward() returns, whatever stream the engine's final
    callback runs under."""
    from adell_mri_amd.modules.layers.conv import Conv3d

    monkeypatch.setitem(HF.FLAGS, "wgrad_stream", True)
    torch.manual_seed(2)
    conv = Conv3d(32, 32, 3, padding=1).to(cuda)
    for p in conv.parameters():
        p._adell_flat = True
    x = torch.randn(2, 32, 48, 48, 48, device=cuda)
    conv(x).sum().backward()
    torch.cuda.synchronize()
    want = conv.weight.grad.clone()
    conv.zero_grad(set_to_none=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        conv(x).sum().backward()
        got = conv.weight.grad.clone()       # queued on s right behind the backward pass
    torch.cuda.synchronize()
    assert torch.equal(got, want)
