"""Register / scratch / LDS use of every kernel in a .hip file, as the compiler reports it
(`-Rpass-analysis=kernel-resource-usage`; no GPU needed). Usage:

    python tools/resource_usage.py adell_mri_amd/csrc/conv_wgrad_s2.hip [extra hipcc flags]
    python tools/resource_usage.py --all          # every source of the library

`parse()` is also what tests/test_resource_usage.py reads the build's `*.ru.txt` files with.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "adell_mri_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]

_FIELDS = {"vgpr": r"VGPRs", "agpr": r"AGPRs", "sgpr": r"SGPRs",
           "scratch": r"ScratchSize \[bytes/lane\]", "occupancy": r"Occupancy \[waves/SIMD\]",
           "lds": r"LDS Size \[bytes/block\]"}


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True,
                                 check=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except Exception:
            continue
    return names


def parse(text):
    """-> list of dicts {name, vgpr, agpr, sgpr, scratch, occupancy, lds} from the remark text."""
    rows = []
    for blk in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        row = {"name": blk.split("\n")[0].strip().split(" [")[0]}
        for key, pat in _FIELDS.items():
            m = re.search(pat + r": (\d+)", blk)
            row[key] = int(m.group(1)) if m else -1
        rows.append(row)
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return rows


def compile_report(path, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        p = subprocess.run([HIPCC, *FLAGS, *extra, "-c", path, "-o", os.path.join(tmp, "o.o")],
                           capture_output=True, text=True, cwd=os.path.dirname(path) or ".")
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-4000:])
        raise SystemExit(p.returncode)
    return parse(p.stderr)


def show(rows, only_scratch=False):
    for r in rows:
        if only_scratch and r["scratch"] == 0:
            continue
        print(f"{r['name'][:96]:96s} V{r['vgpr']:>4} A{r['agpr']:>4} S{r['sgpr']:>4} "
              f"scratch {r['scratch']:>5} occ {r['occupancy']} lds {r['lds']}")


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--all":
        for f in sorted(os.listdir(CSRC)):
            if f.endswith(".hip"):
                print("==", f)
                show(compile_report(os.path.join(CSRC, f)), only_scratch="--scratch" in args)
    else:
        show(compile_report(os.path.abspath(args[0]), args[1:]))
