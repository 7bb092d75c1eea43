"""Pack the reference's hot-path modules, UNMODIFIED, into ONE archive in the git-ignored directory oracle/_ref/ (reference_modules.zip - Python
imports straight from it, like a built library) so that they travel to the GPU box
(`gpurun` ships git-ignored files; `/root/reference` itself does not exist there) and bench.py's two baseline legs can time the REAL
thing instead of the oracle's restatement:

  * `cpu_baseline.kind = "reference"`: models.unet.UNetModelSwin / ldm.models.autoencoder.VQModelTorch / GaussianDiffusion.p_sample_loop
    on the GPU box's host cores;
  * `torch_rocm_autocast_baseline`: the same modules moved to the MI355X under torch.autocast(float16), as sampler.py:185 runs them.

    python -m oracle.make_ref_copy            (build container only: needs /root/reference; __graft_entry__.build() calls it)

Which files: the import closure of the three entry points, found by importing them and listing every loaded module whose file lies under
the reference tree (today: models/{unet,swin_transformer,basic_ops,fp16_util,gaussian_diffusion,respace,script_util,losses}.py,
ldm/{util.py,models/autoencoder.py,modules/{attention,ema}.py,modules/diffusionmodules/{model,util}.py,modules/distributions/...,
modules/vqvae/quantize.py}).  The members are byte-identical (a sha256 manifest of every member and of the archive is written beside it and
checked before anything is imported: oracle/ref_import.py).  Nothing is tracked by git (`.gitignore: oracle/_ref/`): reference SOURCES never
enter this repository's history - oracle/_ref/ holds a build product of this recipe, exactly like the compiled .so next to the kernels.
Test / measurement infrastructure only - the product path (resshift_amd/) never imports oracle/ (tests/test_host_cpu.py greps for it).
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys
import zipfile

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
    os.makedirs(DEST)
    zpath = os.path.join(DEST, "reference_modules.zip")
    manifest = {}
    dirs = set()
    for r in files:
        d = os.path.dirname(r)
        while d and d not in dirs:
            dirs.add(d)
            d = os.path.dirname(d)
    with zipfile.ZipFile(zpath, "w", zipfile.ZIP_DEFLATED) as z:
        for d in sorted(dirs):                       # explicit directory entries: zipimport finds the namespace packages (models/, ldm/) through them
            z.writestr(zipfile.ZipInfo(d + "/", date_time=(1980, 1, 1, 0, 0, 0)), "")
        for r in files:
            with open(os.path.join(SRC, r), "rb") as fh:
                data = fh.read()
            manifest[r] = hashlib.sha256(data).hexdigest()
            zi = zipfile.ZipInfo(r, date_time=(1980, 1, 1, 0, 0, 0))
            zi.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(zi, data)
    with open(os.path.join(DEST, "MANIFEST.json"), "w") as fh:
        json.dump({"source": SRC, "note": "byte-identical members of oracle/_ref/reference_modules.zip = the reference's hot-path modules; git-ignored build "
                                          "product of oracle/make_ref_copy.py", "archive_sha256": _sha(zpath), "sha256": manifest}, fh, indent=1)
    if verbose:
        print(f"[make_ref_copy] {len(files)} files -> {os.path.relpath(zpath, ROOT)}: " + " ".join(files))
    return DEST


if __name__ == "__main__":
    main()
