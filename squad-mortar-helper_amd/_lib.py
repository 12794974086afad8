"""ctypes binding of libsmh_vision_hip.so (the C ABI declared in include/smh_vision_hip.h).

The product path has NO CPU fallback: if the shared library (and with it the gfx950 code object) is
missing, importing this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C squad-mortar-helper_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMH_VISION_HIP_LIB") or os.path.join(_HERE, "libsmh_vision_hip.so")   # env override: diagnostic builds only

MAX_LINES = 32
MAX_SCALES = 3

STAGE_MARKERS, STAGE_UI_MAP, STAGE_OCR, STAGE_SCALES, STAGE_ALL = 0x1, 0x2, 0x4, 0x8, 0xF
STAGE_MINIMAP = 0x10
STAGE_EXACT_STATS = 0x20
STAGE_LSD_HELPERS = 0x40
VIEW_NONE, VIEW_OCR_INPUT, VIEW_FIND_SCALES_INPUT, VIEW_LSD_PREPROCESS, VIEW_LSD_INPUT, VIEW_CROPPED_BRQ = range(6)
IMAGE_UI_MAP = 100

E_INVALID, E_GEOMETRY, E_HIP, E_NO_DEVICE, E_STATE = -1, -2, -3, -4, -5
FRAME_OK, FRAME_LSD_STUCK = 0, 1          # smhv_frame_result.status


class Line(C.Structure):
    _fields_ = [("x0", C.c_float), ("y0", C.c_float), ("x1", C.c_float), ("y1", C.c_float)]


class FrameResult(C.Structure):
    _fields_ = [
        ("map_open", C.c_uint32), ("n_lines", C.c_uint32),
        ("lines", Line * MAX_LINES),
        ("mpx", C.c_double), ("has_mpx", C.c_uint32), ("n_mask_px", C.c_uint32),
        ("red_pixels", C.c_uint32), ("rounds", C.c_uint32), ("ray_steps", C.c_uint64),
        ("length_px", C.c_double * MAX_LINES), ("meters", C.c_double * MAX_LINES), ("angle", C.c_float * MAX_LINES),
        ("minimap", C.c_uint32 * 4), ("has_minimap", C.c_uint32), ("status", C.c_uint32),
    ]


class Anchors(C.Structure):
    _fields_ = [("n", C.c_uint32), ("scales_start_y", C.c_uint32), ("scales", (C.c_uint32 * 3) * MAX_SCALES)]


class BatchLayout(C.Structure):
    _fields_ = [
        ("frame_w", C.c_uint32), ("frame_h", C.c_uint32), ("roi", C.c_uint32 * 4), ("button", C.c_uint32 * 4),
        ("brq_w", C.c_uint32), ("brq_h", C.c_uint32),
        ("ui_pitch", C.c_uint64), ("ui_stride", C.c_uint64), ("ui_offset", C.c_uint64),
        ("mask_pitch", C.c_uint64), ("mask_stride", C.c_uint64), ("mask_offset", C.c_uint64),
        ("ocr_pitch", C.c_uint64), ("ocr_stride", C.c_uint64), ("ocr_offset", C.c_uint64),
        ("scales_pitch", C.c_uint64), ("scales_stride", C.c_uint64), ("scales_offset", C.c_uint64),
        ("bits_pitch_words", C.c_uint64), ("bits_stride", C.c_uint64), ("bits_xoff", C.c_uint64),
    ]


LOG_FN = C.CFUNCTYPE(None, C.c_int, C.c_char_p)

# name -> (restype, argtypes); every symbol include/smh_vision_hip.h declares
SIGNATURES = {
    "smhv_init": (C.c_int, [C.c_int, LOG_FN, C.POINTER(C.c_void_p)]),
    "smhv_shutdown": (None, [C.c_void_p]),
    "smhv_thread_ctx": (C.c_int, [C.c_void_p]),
    "smhv_set_ray_table": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "smhv_last_error": (C.c_char_p, []),
    "smhv_map_bounds": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "smhv_button_bounds": (C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "smhv_load_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "smhv_load_frame_view": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 6),
    "smhv_load_frame_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "smhv_crop_to_map": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.c_void_p]),
    "smhv_red_pixels": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "smhv_ocr_preprocess": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "smhv_find_scales_preprocess": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smhv_isolate_map_markers": (C.c_int, [C.c_void_p]),
    "smhv_mask_marker_lines": (C.c_int, [C.c_void_p]),
    "smhv_get_lsd_image": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smhv_find_longest_line": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(Line), C.POINTER(C.c_float)]),
    "smhv_find_marker_lines": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Line), C.POINTER(C.c_uint32)]),
    "smhv_lsd_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "smhv_calc_meters_to_px_ratio": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "smhv_find_minimap": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "smhv_get_debug_view": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smhv_batch_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "smhv_batch_destroy": (None, [C.c_void_p]),
    "smhv_batch_layout_get": (C.c_int, [C.c_void_p, C.POINTER(BatchLayout)]),
    "smhv_batch_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]),
    "smhv_batch_device_ptrs": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 6),
    "smhv_batch_tile_mask": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "smhv_batch_read_tile_mask": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "smhv_batch_read_results": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(FrameResult)]),
    "smhv_batch_read_image": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]),
    "smhv_batch_set_scales_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smhv_batch_wait_map_pass": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smhv_batch_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "smhv_batch_lsd_coop_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "smhv_batch_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "smhv_pipeline_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "smhv_pipeline_create_ex": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "smhv_pipeline_hold": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "smhv_debug_pipeline_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "smhv_debug_pipeline_peek": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "smhv_pipeline_destroy": (None, [C.c_void_p]),
    "smhv_pipeline_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "smhv_pipeline_wait": (C.c_int, [C.c_void_p, C.c_uint32]),
    "smhv_pipeline_wait_all": (C.c_int, [C.c_void_p]),
    "smhv_pipeline_slot": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "smhv_shard_range": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "smhv_node_create": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, LOG_FN, C.POINTER(C.c_void_p)]),
    "smhv_node_destroy": (None, [C.c_void_p]),
    "smhv_node_ctx": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "smhv_node_run": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "smhv_node_gather": (C.c_int, [C.c_void_p, C.POINTER(FrameResult), C.POINTER(C.c_uint32)]),
    "smhv_debug_lsd_classic": (C.c_int, [C.c_int]),
    "smhv_debug_lsd_tile_cap": (C.c_int, [C.c_uint32]),
    "smhv_debug_lsd_spin_limit": (C.c_int, [C.c_uint32]),
    "smhv_debug_lsd_threads": (C.c_int, [C.c_uint32]),
    "smhv_debug_skip_line_search": (C.c_int, [C.c_int]),
    "smhv_debug_no_host_atomics": (C.c_int, [C.c_int]),
    "smhv_debug_side_kernel": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "smhv_trait_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "smhv_ui_map": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smhv_debug_pattern_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "smhv_debug_marker_table": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smhv_ingest_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "smhv_ingest_create_ex": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "smhv_crc32_host": (C.c_uint32, [C.c_void_p, C.c_uint64]),
    "smhv_debug_map_band_rows": (C.c_int, [C.c_uint32]),
    "smhv_debug_band_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "smhv_debug_ingest_feed": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "smhv_debug_crc32_host_level": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]),
    "smhv_ingest_destroy": (None, [C.c_void_p]),
    "smhv_ingest_local_cpus": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "smhv_ingest_bind_thread": (C.c_int, [C.c_void_p]),
    "smhv_ingest_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "smhv_ingest_commit": (C.c_int, [C.c_void_p]),
    "smhv_ingest_push": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smhv_ingest_commit_pixels": (C.c_int, [C.c_void_p, C.c_uint32]),
    "smhv_ingest_push_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "smhv_ingest_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smhv_ingest_reset": (C.c_int, [C.c_void_p]),
    "smhv_ingest_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "smhv_crc32_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]),
}


class PipelineOptions(C.Structure):
    """smhv_pipeline_options (include/smh_vision_hip.h): every 0 is the library's default."""
    _fields_ = [("size", C.c_uint32), ("search", C.c_uint32), ("streams", C.c_uint32), ("idle_close_us", C.c_uint32), ("occupancy_policy", C.c_uint32),
                ("late_helpers", C.c_uint32), ("service_workgroups", C.c_uint32), ("flags", C.c_uint32), ("remote_after", C.c_uint32), ("remote_tickets", C.c_uint32), ("remote_last", C.c_uint32), ("room_for_others", C.c_uint32)]


SEARCH_AUTO, SEARCH_BATCH, SEARCH_FRAME = 0, 1, 2
PIPE_NO_TEAM_HELP, PIPE_NO_STREAM_PRIORITY, PIPE_NO_PROLOGUE, PIPE_NO_REMOTE_HELP, PIPE_HELP_FIRST, PIPE_WALK_BIT_ROWS, PIPE_THREE_LOAD_SETS = 1, 2, 4, 8, 16, 32, 64


class VisionError(RuntimeError):
    """Any non-zero status from the library (the reference's anyhow::Error)."""

    def __init__(self, code, msg):
        super().__init__("smh_vision_hip error %d: %s" % (code, msg))
        self.code = code
        self.records = None   # E_STATE from read_results / gather: the records that were copied out before the error was raised


_lib = None


def load():
    """dlopen the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: the HIP extension is not built (run __graft_entry__.build() or "
                "`make -C squad-mortar-helper_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in
        # one process do not work ("No HIP GPUs are available"), so when torch is installed it is
        # imported first and this library binds to the runtime torch loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise VisionError(rc, load().smhv_last_error().decode("utf-8", "replace"))
