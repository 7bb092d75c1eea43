"""Copy the reference's hot-path modules, UNMODIFIED, into the git-ignored directory oracle/_ref/ so that they travel to the GPU box
(`gpurun` ships git-ignored files; `/root/reference` itself does not exist there) and bench.py's two baseline legs can time the REAL
thing instead of the oracle's restatement:

  * `cpu_baseline.kind = "reference"`: models.unet.UNetModelSwin / ldm.models.autoencoder.VQModelTorch / GaussianDiffusion.p_sample_loop
    on the GPU box's host cores;
  * `torch_rocm_autocast_baseline`: the same modules moved to the MI355X under torch.autocast(float16), as sampler.py:185 runs them.

    python -m oracle.make_ref_copy            (build container only: needs /root/reference; __graft_entry__.build() calls it)

Which files: the import closure of the three entry points, found by importing them and listing every loaded module whose file lies under
the reference tree (today: models/{unet,swin_transformer,basic_ops,fp16_util,gaussian_diffusion,respace,script_util,losses}.py,
ldm/{util.py,models/autoencoder.py,modules/{attention,ema}.py,modules/diffusionmodules/{model,util}.py,modules/distributions/...,
modules/vqvae/quantize.py}).  The copies are byte-identical (a sha256 manifest is written beside them and checked when they are loaded:
oracle/ref_import.py).  Nothing is tracked by git (`.gitignore: oracle/_ref/`): reference SOURCES never enter this repository's history.
Test / measurement infrastructure only - the product path (resshift_amd/) never imports oracle/ (tests/test_host_cpu.py greps for it).
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEST = os.path.join(ROOT, "oracle", "_ref")
SRC = os.environ.get("RESSHIFT_REFERENCE", "/root/reference")


def _sha(path: str) -> str:
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def closure() -> "list[str]":
    """relative paths of every reference file the three entry points pull in (plus the package __init__ files on the way)"""
    from oracle import ref_import

    ref_import.load(prefer=SRC)
    src = os.path.realpath(SRC) + os.sep
    rel = set()
    for m in list(sys.modules.values()):
        f = getattr(m, "__file__", None)
        if f and os.path.realpath(f).startswith(src) and f.endswith(".py"):
            r = os.path.relpath(os.path.realpath(f), src)
            rel.add(r)
            d = os.path.dirname(r)
            while d:   # namespace / regular packages above it
                init = os.path.join(d, "__init__.py")
                if os.path.exists(os.path.join(src, init)):
                    rel.add(init)
                d = os.path.dirname(d)
    return sorted(rel)


def main(verbose: bool = True) -> "str | None":
    if not os.path.isdir(os.path.join(SRC, "models")):
        if verbose:
            print(f"[make_ref_copy] {SRC} not present: nothing to do (the GPU box uses the copy that travelled with the snapshot)")
        return None
    files = closure()
    if os.path.isdir(DEST):
        shutil.rmtree(DEST)
    manifest = {}
    for r in files:
        dst = os.path.join(DEST, r)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(SRC, r), dst)
        manifest[r] = _sha(dst)
    with open(os.path.join(DEST, "MANIFEST.json"), "w") as fh:
        json.dump({"source": SRC, "note": "byte-identical copies of the reference's hot-path modules; git-ignored; see oracle/make_ref_copy.py",
                   "sha256": manifest}, fh, indent=1)
    if verbose:
        print(f"[make_ref_copy] {len(files)} files -> {os.path.relpath(DEST, ROOT)}/: " + " ".join(files))
    return DEST


if __name__ == "__main__":
    main()
