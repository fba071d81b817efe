"""Imports the reference implementation in-process (build container only).

Recipe from SURVEY.md Appendix B: torchvision and skimage are absent and only used
off-path, so empty stub modules are injected before the import.  Nothing is copied from
the reference; the modules are used solely to validate the oracle and to generate golden
vectors (tests/golden/make_golden.py).
"""
import os
import sys
import types

REFERENCE_DIR = "/root/reference/Backend"


def _stub_modules():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True          # the reference tree is read-only
    for name in ("torchvision", "torchvision.transforms", "torchvision.models",
                 "skimage", "skimage.metrics"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = lambda *a, **k: None
    sys.modules["skimage.metrics"].structural_similarity = lambda *a, **k: None
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)


def import_reference():
    _stub_modules()
    import DDIM.DDIMModel as ddim
    import cddpm.cddpmModels as cddpm
    return types.SimpleNamespace(ddim=ddim, cddpm=cddpm)


def import_hybrid():
    """The hybrid router file with its own copies of UNetDiffusion / DiffusionDenoiser
    (hybrid/hybrid3diffusionspeed.py:308-418) and HybridDenoisingRouter (:560-628)."""
    _stub_modules()
    import hybrid.hybrid3diffusionspeed as hyb
    return hyb
