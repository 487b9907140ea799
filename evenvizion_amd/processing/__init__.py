"""evenvizion_amd.processing -- same public surface as evenvizion.processing (evenvizion/processing/__init__.py:16-24)."""
__version__ = "0.9"

__all__ = ['constants', 'frame_processing', 'fixed_coordinate_system', 'matching', 'utils', 'video_processing']

from .constants import *  # noqa: F401,F403
from .frame_processing import *  # noqa: F401,F403
from .fixed_coordinate_system import *  # noqa: F401,F403
from .matching import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
from .video_processing import *  # noqa: F401,F403
