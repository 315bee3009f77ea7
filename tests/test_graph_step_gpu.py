"""trainer.StepRunner.enable_graph: one training step captured in a HIP graph and replayed -- losses
and parameters bit-identical to the eager loop, dropout included (the graph advances the library's
replay counter of the dropout offsets, adell_rng_advance), different batches through the static
inputs, and back to eager steps afterwards. Reference loop: lightning.Trainer.fit over eager torch
ops (adell_mri/entrypoints/segmentation/train.py:799-819)."""
import copy
import itertools
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

# hipStreamEndCapture of a whole training step segfaults INSIDE the ROCm runtime on some boxes of
# this pool (config 4's step always, config 3's on some days: the same commit and binary replayed
# fine in the morning and crashed in the afternoon -- DESIGN.md, known gaps). A segfault would take
# the whole test process down, so every capture runs in a child process: a child that dies inside
# capture_end skips the test with that reason; any other failure fails it.


def _in_child(name, *args, extra_env=None):
    env = dict(os.environ, ADELL_CHECK_DENSE="1", **(extra_env or {}))
    r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), name, *map(str, args)],
                       capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if r.returncode < 0 and "capture_end" in r.stderr:
        pytest.skip("hipStreamEndCapture crashed inside the ROCm runtime on this box (signal "
                    f"{-r.returncode}); the replay path could not be exercised here")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _unetr(cuda, dropout_rate):
    from adell_mri_amd.modules.activations import activation_factory
    from adell_mri_amd.modules.segmentation.losses import (CompoundLoss, binary_focal_loss,
                                                           binary_generalized_dice_loss)
    from adell_mri_amd.modules.segmentation.pl import UNETRPL

    torch.manual_seed(3)
    loss_fn = CompoundLoss([(binary_generalized_dice_loss, {"smooth": 1e-5, "eps": 1e-6}),
                            (binary_focal_loss, {"gamma": 1.0, "eps": 1e-6})])
    net = UNETRPL(image_key="image", label_key="mask", learning_rate=5e-3, weight_decay=5e-4,
                  loss_fn=loss_fn, image_size=[32, 32, 32], patch_size=[8, 8, 8], number_of_blocks=4,
                  return_at=[1, 2], embedding_size=64, n_heads=4, dropout_rate=dropout_rate,
                  mlp_structure=[128], spatial_dimensions=3, conv_type="regular",
                  link_type="residual", upscale_type="transpose", norm_type="instance", padding=1,
                  dropout_param=0.1, activation_fn=activation_factory["leaky_relu"], in_channels=1,
                  n_classes=2, depth=[8, 16, 32], kernel_sizes=[3] * 3)
    return net.to(cuda).train()


def _batches(cuda, n):
    g = torch.Generator().manual_seed(7)
    return [{"image": torch.rand((2, 1, 32, 32, 32), generator=g).to(cuda),
             "mask": (torch.rand((2, 1, 32, 32, 32), generator=g) > 0.8).float().to(cuda)}
            for _ in range(n)]


@pytest.mark.parametrize("dropout_rate", [0.1, 0.0])
def test_replayed_steps_equal_eager_steps_bit_for_bit(cuda, dropout_rate):
    _in_child("replayed_steps_equal_eager_steps", dropout_rate)


def test_a_child_that_dies_inside_capture_end_skips_and_any_other_death_fails(cuda):
    """The guard itself: a child killed by a signal inside a frame called capture_end is a skip with
    the reason spelled out; a child that dies anywhere else is a failure."""
    with pytest.raises(pytest.skip.Exception, match="hipStreamEndCapture"):
        _in_child("replayed_steps_equal_eager_steps", 0.0, extra_env={"ADELL_TEST_FAKE_CRASH": "capture_end"})
    with pytest.raises(AssertionError):
        _in_child("replayed_steps_equal_eager_steps", 0.0, extra_env={"ADELL_TEST_FAKE_CRASH": "elsewhere"})


def replayed_steps_equal_eager_steps(cuda, dropout_rate):
    from adell_mri_amd import functional as HF
    from adell_mri_amd import ops
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    batches = _batches(cuda, 7)
    base = _unetr(cuda, dropout_rate)
    runs = {}
    for mode in ("eager", "graph"):
        net = copy.deepcopy(base)
        opt = net.configure_optimizers()["optimizer"]
        runner = StepRunner(net, opt, GradSync(opt))
        HF._dropout_counter = itertools.count(1)
        ops.rng_advance(0, set_value=True)
        losses = []
        if mode == "graph":
            # (warm-up steps run on the first batch, as the eager loop below does)
            losses.append(float(runner.train_step(batches[0]).detach()))
            n0 = runner.step_idx
            runner.enable_graph(batches[0], warmup=1)
            assert runner.step_idx == n0 + 1
            losses.append(None)                          # (the warm-up step's loss is not returned)
            for b in batches[1:5]:
                losses.append(float(runner.train_step(b).detach()))
            runner.disable_graph()                       # and two eager steps after the graph
            for b in batches[5:]:
                losses.append(float(runner.train_step(b).detach()))
        else:
            for b in [batches[0], batches[0]] + batches[1:]:
                losses.append(float(runner.train_step(b).detach()))
        torch.cuda.synchronize()
        runs[mode] = (losses, {k: v.detach().clone() for k, v in net.named_parameters()})
    runs["eager"][0][1] = None
    assert runs["eager"][0] == runs["graph"][0], (runs["eager"][0], runs["graph"][0])
    for k, v in runs["eager"][1].items():
        assert torch.equal(v, runs["graph"][1][k]), k
    if dropout_rate > 0:
        # the masks do change from replay to replay: the same batch twice gives different losses
        net = copy.deepcopy(base)
        opt = net.configure_optimizers()["optimizer"]
        for g in opt.param_groups:
            g["lr"] = 0.0
        runner = StepRunner(net, opt, GradSync(opt))
        runner.enable_graph(batches[0], warmup=1)
        a = float(runner.train_step(batches[0]).detach())
        b = float(runner.train_step(batches[0]).detach())
        runner.disable_graph()
        assert a != b


def test_graph_mode_refuses_hooked_gradient_buckets(cuda):
    from adell_mri_amd.parallel import GradSync
    from adell_mri_amd.trainer import StepRunner

    net = _unetr(cuda, 0.0)
    opt = net.configure_optimizers()["optimizer"]
    sync = GradSync(opt)
    sync.overlap = True          # what a world of > 1 ranks sets up
    with pytest.raises(RuntimeError, match="backward hooks"):
        StepRunner(net, opt, sync).enable_graph(_batches(cuda, 1)[0])


def capture_end():          # (the frame name the guard looks for, for the guard's own test)
    import signal

    os.kill(os.getpid(), signal.SIGSEGV)


def _elsewhere():
    import signal

    os.kill(os.getpid(), signal.SIGSEGV)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if os.environ.get("ADELL_TEST_FAKE_CRASH") == "capture_end":
        capture_end()
    elif os.environ.get("ADELL_TEST_FAKE_CRASH"):
        _elsewhere()
    globals()[sys.argv[1]](torch.device("cuda:0"), *[float(v) for v in sys.argv[2:]])
