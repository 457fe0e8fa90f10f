"""volxel_amd -- MI355X-native drop-in for the raymarch hot path of Volxel's volxel-3d-viewer.

Exports mirror volxel-3d-viewer/src/index.ts:1-4 for the parts on the path
(viewer render core + utils/data transfer-function helpers).
"""
from .renderer import (Volxel3DRenderer, VolxelError, compute_params, sample_weight,  # noqa: F401
                       register_volxel_components)
from .transfer import (default_transfer_function, generate_transfer_function,  # noqa: F401
                       parse_transfer_function)
from .settings import (ViewerSettings, BENCHMARK_SETTINGS, BENCHMARK_COLLECTION_MODES, verify_settings,  # noqa: F401
                       verify_benchmark, load_settings)
from .scene import Camera, Volume, Grid  # noqa: F401
from .preprocessor import read_u16_stack_to_grid, read_dicoms_to_grid, BrickGridMessage  # noqa: F401
from .environment import Environment  # noqa: F401,E402
from .containers import ZipReadError, read_zip_slices, decode_environment  # noqa: F401,E402
