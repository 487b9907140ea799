"""evenvizion_amd -- MI355X (gfx950) implementation of EvenVizion's frame-to-frame homography hot path.

evenvizion_amd.processing mirrors the public surface of the reference's evenvizion.processing; every kernel is
hand-written HIP reached through the C ABI of libevhip.so (include/evhip.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def install_as_evenvizion():
    """Register evenvizion_amd.processing under the reference's module names (evenvizion.processing.*), so code
    written against the reference -- e.g. evenvizion/examples/evenvizion_component.py:30-35 -- imports this
    implementation unchanged.  Only the processing sub-package is aliased: when the real `evenvizion` package is
    importable its search path is kept (without executing its __init__, which pulls in every example script), so
    `evenvizion.visualization.*` (out of scope here) still resolves to the reference's own files and finds this
    implementation's `evenvizion.processing.constants` / `.utils` underneath."""
    import importlib
    import importlib.util
    import sys
    import types
    pkg = importlib.import_module("evenvizion_amd.processing")
    root = sys.modules.get("evenvizion")
    if root is None:
        root = types.ModuleType("evenvizion")
        root.__path__ = []
        try:
            spec = importlib.util.find_spec("evenvizion")     # a top-level lookup executes nothing
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.submodule_search_locations:
            root.__path__ = list(spec.submodule_search_locations)
            root.__file__ = spec.origin
            root.__spec__ = spec
        sys.modules["evenvizion"] = root
    elif not hasattr(root, "__path__"):
        root.__path__ = []
    sys.modules["evenvizion.processing"] = pkg
    root.processing = pkg
    for name in ("constants", "frame_processing", "fixed_coordinate_system", "matching", "utils", "video_processing"):
        sys.modules["evenvizion.processing." + name] = importlib.import_module("evenvizion_amd.processing." + name)
    return pkg
