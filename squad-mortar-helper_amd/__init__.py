"""squad-mortar-helper_amd: MI355X-native vision hot path of squad-mortar-helper.

Product code only -- nothing here imports oracle/.  Importing the binding submodules requires the
built HIP library (see _lib.load); geometry helpers and the synthetic generator also need it
because the screen-relative bounds are computed by the library itself.
"""
from . import _lib  # noqa: F401
from ._lib import VisionError, STAGE_ALL, STAGE_EXACT_STATS, STAGE_MARKERS, STAGE_MINIMAP, STAGE_LSD_HELPERS, STAGE_OCR, STAGE_SCALES, STAGE_UI_MAP  # noqa: F401
from .vision import DebugView, HipVision, VisionResults, VisionState, button_bounds, map_bounds, parse_ocr_labels  # noqa: F401
from .batch import FrameBatch, Pipeline, make_anchors, results_to_dicts  # noqa: F401
from .ingest import IngestQueue, crc32_device  # noqa: F401
