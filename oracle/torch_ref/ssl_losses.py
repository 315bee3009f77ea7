"""CPU restatement (stock torch, explicit arithmetic) of the cosine-similarity losses
SelfSLBasePL.init_loss selects (TEST INFRASTRUCTURE ONLY;
adell_mri/modules/self_supervised/losses/functional.py:138-164, losses/ntxent.py:11-46), pinned
to tests/golden/ssl_pair_losses.npz by tests/test_ssl.py."""
import torch


def _unit(x, eps=1e-8):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def pair_loss(x1, x2, kind, temperature=1.0, apply_relu=True):
    if kind in ("simsiam", "byol"):
        s = -(_unit(x1) * _unit(x2)).sum(-1).mean()
        return s if kind == "simsiam" else 2 * s + 2
    assert kind == "ntxent"
    if apply_relu:
        x1, x2 = torch.relu(x1), torch.relu(x2)
    z = _unit(torch.cat([x1, x2], 0))
    n = z.shape[0]
    sim = z @ z.T / temperature
    partner = (torch.arange(n) + n // 2) % n
    positives = sim[torch.arange(n), partner]
    others = sim.masked_fill(torch.eye(n, dtype=torch.bool), float("-inf"))
    return (-positives + torch.logsumexp(others, dim=-1)).mean()
