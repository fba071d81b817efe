"""Makes the package directory ``medical-image-denoising-using-diffusion_amd/`` importable.

The directory name is fixed by the project layout and contains hyphens, so it cannot be
imported by name.  ``load()`` registers it in ``sys.modules`` under the alias ``midd_amd``;
afterwards ``import midd_amd.config`` etc. work as usual.
"""
import importlib.util
import os
import sys

ALIAS = "midd_amd"
PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                       "medical-image-denoising-using-diffusion_amd")


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
