"""Load the product package the way ComfyUI loads a custom node: by directory path.

The directory name (`comfyui-egregora-audio-super-resolution_amd`) is not a valid Python identifier,
so tests, bench.py and __graft_entry__.py import it under the alias `egregora_amd`.
"""
import importlib.util
import sys
from pathlib import Path

PACK_DIR = Path(__file__).resolve().parent / "comfyui-egregora-audio-super-resolution_amd"


def load_pack(alias: str = "egregora_amd"):
    if alias in sys.modules:
        return sys.modules[alias]
    spec = importlib.util.spec_from_file_location(alias, PACK_DIR / "__init__.py",
                                                  submodule_search_locations=[str(PACK_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return mod
