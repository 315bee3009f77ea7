"""Fused optimisers: torch.optim-compatible state_dict round trip (CPU: no kernel launch is
needed to load / save), and on the GPU the step after a load and the treatment of parameters
without a gradient against torch.optim itself (reference: torch.optim.SGD / AdamW as configured
at adell_mri/modules/segmentation/pl.py:563-569, self_supervised/pl.py:245-250)."""
import copy
import io

import pytest
import torch

from adell_mri_amd.optim import (FusedAdagrad, FusedAdamax, FusedAdamW, FusedNAdam, FusedRAdam,
                                 FusedRMSprop, FusedSGD)


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    m.frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
    m.unused = torch.nn.Parameter(torch.full((7,), 2.0))
    return m


def _torch_steps(m, make, n):
    opt = make(m.parameters())
    g = torch.Generator().manual_seed(1)
    for _ in range(n):
        opt.zero_grad()
        m(torch.randn(4, 6, generator=g)).pow(2).mean().backward()
        opt.step()
    return opt


SGD = dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=0.01)
ADAMW = dict(lr=0.01, betas=(0.8, 0.9), eps=1e-6, weight_decay=0.1)
# name -> (torch.optim class, fused class, keyword arguments): the eight names of the
# reference's optimizer factory (utils/optimizer_factory.py:5-14)
KINDS = {
    "sgd": (torch.optim.SGD, FusedSGD, SGD),
    "adamw": (torch.optim.AdamW, FusedAdamW, ADAMW),
    "adamax": (torch.optim.Adamax, FusedAdamax, dict(lr=0.01, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.05)),
    "adagrad": (torch.optim.Adagrad, FusedAdagrad, dict(lr=0.05, lr_decay=0.01, eps=1e-8, weight_decay=0.05)),
    "nadam": (torch.optim.NAdam, FusedNAdam, dict(lr=0.01, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.05)),
    "radam": (torch.optim.RAdam, FusedRAdam, dict(lr=0.01, betas=(0.8, 0.9), eps=1e-6, weight_decay=0.05)),
    "rmsprop": (torch.optim.RMSprop, FusedRMSprop, dict(lr=0.01, alpha=0.9, eps=1e-6, weight_decay=0.05)),
}


@pytest.mark.parametrize("kind", list(KINDS))
def test_load_torch_state_dict_keeps_parameters_aliased_and_round_trips(kind):
    tcls, fcls, kw = KINDS[kind]
    m = _model()
    topt = _torch_steps(m, lambda p: tcls(p, **kw), 2)
    sd = topt.state_dict()
    m2 = copy.deepcopy(m)
    fused = fcls(m2.parameters(), lr=1.0)
    flat = fused.flat_groups[0]
    assert "_flat" not in fused.param_groups[0]
    fused.load_state_dict(sd)
    # parameters still alias the flat buffer after the load
    for p, o in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.data.data_ptr() + 4 * o
    assert fused.flat_groups[0] is flat
    g = fused.param_groups[0]
    assert g["lr"] == kw["lr"]
    out = fused.state_dict()
    assert [pg["params"] for pg in out["param_groups"]] == [pg["params"] for pg in sd["param_groups"]]
    # frozen / unused parameters: no state (torch's Adagrad creates zero-step entries for every
    # parameter at construction: those carry no information)
    stepped = {k for k, v in sd["state"].items() if float(v.get("step", 1)) > 0}
    assert set(out["state"]) == stepped
    for pid, ent in sd["state"].items():
        if pid not in stepped:
            continue
        for k, v in ent.items():
            got = out["state"][pid][k]
            if torch.is_tensor(v):
                assert torch.equal(got.reshape(v.shape).to(v.dtype), v), (pid, k)
    # and it survives torch.save / torch.load(weights_only=True) like a torch optimiser's
    buf = io.BytesIO()
    torch.save(out, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=True)
    fused.load_state_dict(again)
    assert set(fused.state_dict()["state"]) == stepped


@pytest.mark.gpu
@pytest.mark.parametrize("kind", list(KINDS))
def test_step_after_load_and_gradientless_parameters_match_torch(cuda, kind):
    tcls, fcls, kw = KINDS[kind]
    m = _model()
    topt = _torch_steps(m, lambda p: tcls(p, **kw), 2)
    m2 = copy.deepcopy(m).to(cuda)
    fused = fcls(m2.parameters())
    fused.load_state_dict(topt.state_dict())
    g = torch.Generator().manual_seed(9)
    for _ in range(6 if kind == "radam" else 2):   # RAdam: past the rectification threshold
        x = torch.randn(4, 6, generator=g)
        topt.zero_grad()
        m(x).pow(2).mean().backward()
        topt.step()
        fused.zero_grad()
        m2(x.to(cuda)).pow(2).mean().backward()
        fused.step()
    for (k, a), b in zip(m.named_parameters(), m2.parameters()):
        assert torch.allclose(a, b.cpu(), rtol=2e-5, atol=1e-6), k
    # no gradient -> untouched (no weight decay, no momentum), as torch.optim
    assert torch.equal(m2.unused.detach().cpu(), torch.full((7,), 2.0))
    # a parameter that starts receiving gradients later begins its own state then
    for mm, opt in ((m, topt), (m2, fused)):
        dev = next(mm.parameters()).device
        opt.zero_grad()
        (mm(torch.ones(2, 6, device=dev)).sum() + (mm.unused ** 2).sum()).backward()
        opt.step()
    assert torch.allclose(m.unused, m2.unused.cpu(), rtol=2e-5, atol=1e-6)
    for (k, a), b in zip(m.named_parameters(), m2.parameters()):
        assert torch.allclose(a, b.cpu(), rtol=2e-5, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sgd", "adamw"])
def test_a_gradient_cleared_after_collect_is_skipped_as_torch_skips_it(cuda, kind):
    """``p.grad = None`` between ``collect()`` (GradSync.all_reduce runs it) and ``step()``: the
    parameter is left alone, like torch.optim leaves it -- not stepped with the stale slot."""
    tcls, fcls, kw = KINDS[kind]
    m = _model()
    m2 = copy.deepcopy(m).to(cuda)
    topt, fused = tcls(m.parameters(), **kw), fcls(m2.parameters(), **kw)
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(3))
    for step in range(2):
        for mm, opt in ((m, topt), (m2, fused)):
            opt.zero_grad()
            mm(x.to(next(mm.parameters()).device)).pow(2).mean().backward()
            if opt is fused:
                fused.collect_grads()
            if step == 1:
                mm[0].bias.grad = None
            opt.step()
    for (k, a), b in zip(m.named_parameters(), m2.parameters()):
        assert torch.allclose(a, b.cpu(), rtol=2e-5, atol=1e-6), k


def test_get_optimizer_covers_the_reference_factory_names():
    # the module path of the reference (utils/optimizer_factory.py:5-54), its three public names
    from adell_mri_amd.modules.segmentation import pl
    from adell_mri_amd.utils.optimizer_factory import (OPTIMIZER_EPS_DEFAULT, OPTIMIZER_MATCH,
                                                       get_optimizer,
                                                       optimizer_eps_from_precision)

    assert pl.get_optimizer is get_optimizer and pl.OPTIMIZER_MATCH is OPTIMIZER_MATCH
    assert optimizer_eps_from_precision("16-true") == 1e-4
    for prec in (None, "32", "16", "16-mixed", "bf16", "bf16-mixed", 32):
        assert optimizer_eps_from_precision(prec) == OPTIMIZER_EPS_DEFAULT == 1e-8

    assert sorted(OPTIMIZER_MATCH) == sorted(["adam", "adamw", "adamax", "sgd", "adagrad", "nadam",
                                              "radam", "rmsprop"])
    assert get_optimizer("str", []) is None       # unknown name -> None, as the reference
