"""evenvizion_amd -- MI355X (gfx950) implementation of EvenVizion's frame-to-frame homography hot path.

evenvizion_amd.processing mirrors the public surface of the reference's evenvizion.processing; every kernel is
hand-written HIP reached through the C ABI of libevhip.so (include/evhip.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def install_as_evenvizion():
    """Register evenvizion_amd.processing under the reference's module names (evenvizion.processing.*), so code
    written against the reference -- e.g. evenvizion/examples/evenvizion_component.py:30-35 -- imports this
    implementation unchanged.  Only the processing sub-package is aliased (visualisation is out of scope)."""
    import importlib
    import sys
    import types
    pkg = importlib.import_module("evenvizion_amd.processing")
    root = sys.modules.get("evenvizion") or types.ModuleType("evenvizion")
    root.__path__ = getattr(root, "__path__", [])
    sys.modules["evenvizion"] = root
    sys.modules["evenvizion.processing"] = pkg
    root.processing = pkg
    for name in ("constants", "frame_processing", "fixed_coordinate_system", "matching", "utils", "video_processing"):
        sys.modules["evenvizion.processing." + name] = importlib.import_module("evenvizion_amd.processing." + name)
    return pkg
