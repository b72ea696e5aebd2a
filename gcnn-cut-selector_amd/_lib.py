"""ctypes binding of libgcnn_hip.so (C ABI declared in include/gcnn_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is raised."""

from __future__ import annotations

import ctypes as C
import os

# Kernel arguments in device memory: with them in host memory every launch of this library (argument blocks of 1-3 KB) pays
# PCIe round trips -- measured +0.1 ms per training step.  Must be set before the HIP runtime initialises; this image
# already defaults to 1, so this only guards against an environment that does not.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: F401  -- MUST be imported before the CDLL below: torch ships its own libamdhip64; loading ours first
#                              would register the kernels with a second HIP runtime (hipErrorNoDevice at first launch)

LIB_PATH = os.environ.get("GCNN_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libgcnn_hip.so")  # GCNN_LIB: A/B builds
ABI_VERSION = 11


class GcnnError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("n_cons", C.c_int32), ("n_vars", C.c_int32), ("n_cuts", C.c_int32),
                ("n_cons_edges", C.c_int32), ("n_cut_edges", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("l_ptr", C.c_void_p), ("l_oth", C.c_void_p), ("l_coef", C.c_void_p),
                ("v_ptr", C.c_void_p), ("v_oth", C.c_void_p), ("v_coef", C.c_void_p),
                ("l_max_deg", C.c_int32), ("v_max_deg", C.c_int32)]


class AdamArgs(C.Structure):
    _fields_ = [("params", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("lr_t", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float)]


class InferLayout(C.Structure):
    _fields_ = [("in_bytes", C.c_size_t), ("in_off", C.c_size_t * 8), ("out_bytes", C.c_size_t), ("out_off", C.c_size_t * 3),
                ("arena_bytes", C.c_size_t), ("dev_off", C.c_size_t * 8)]


class CollateJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("unit_kind", C.c_int32), ("width", C.c_int32),
                ("add_kind", C.c_int32), ("is_ptr", C.c_int32)]


_P, _I, _F, _Z = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
_DP, _GP = C.POINTER(Dims), C.POINTER(Graph)

# name -> (restype, argtypes); every symbol include/gcnn_hip.h declares
SIGNATURES = {
    "gcnn_abi_version": (C.c_int, []),
    "gcnn_profile_begin": (C.c_int, []),
    "gcnn_profile_end": (C.c_int, [_I, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
    "gcnn_param_count": (C.c_int, []),
    "gcnn_param_total_floats": (C.c_int, []),
    "gcnn_param_info": (C.c_int, [C.c_int] + [C.POINTER(C.c_int)] * 4),
    "gcnn_graph_temp_bytes": (_Z, [_I]),
    "gcnn_graph_check": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "gcnn_collate": (C.c_int, [_P, _I, _P, _P, _I, C.c_int64, _P]),
    "gcnn_graph_build": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "gcnn_seg_sum_f32": (C.c_int, [_P, _P, _P, _I, _P, _P]),
    "gcnn_seg_bcast_f32": (C.c_int, [_P, _P, _P, _I, _P, _P]),
    "gcnn_linear_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "gcnn_linear_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P]),
    "gcnn_conv_edge_fwd": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "gcnn_conv_edge_bwd_recv": (C.c_int, [_P, _P, _P, _I, _P, _P]),
    "gcnn_conv_edge_bwd_send": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int32), _I, _P]),
    "gcnn_workspace_floats": (_Z, [_DP]),
    "gcnn_forward": (C.c_int, [_DP, _P, _P, _P, _P, _GP, _GP, _P, _Z, _P, _I, _P]),
    "gcnn_infer_layout_for": (C.c_int, [_DP, C.POINTER(InferLayout)]),
    "gcnn_infer": (C.c_int, [_DP, _P, _P, _P, _P, _Z, _I, _P]),
    "gcnn_mse_loss": (C.c_int, [_P, _P, _I, _F, _P, _P, _P]),
    "gcnn_forward_loss": (C.c_int, [_DP, _P, _P, _P, _P, _GP, _GP, _P, _Z, _P, _P, _F, _P]),
    "gcnn_backward": (C.c_int, [_DP, _P, _P, _P, _P, _GP, _GP, _P, _Z, _P, _P, _P, _P, _P, _P]),
    "gcnn_prenorm_stats": (C.c_int, [_DP, _P, _P, _P, _P, _GP, _GP, _P, _Z, _I, _P, _P]),
    "gcnn_adam_step": (C.c_int, [_P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _I, _P]),
    "gcnn_ranking_metric": (C.c_int, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _F, _P, _P]),
    "gcnn_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _I, _P]),
    "gcnn_host_sort_edges_by_row": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gcnn_host_pack_edges": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P]),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises GcnnError when it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GcnnError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(or gcnn-cut-selector_amd/csrc/build.sh); there is no CPU fallback")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as exc:  # pragma: no cover
            raise GcnnError(f"cannot load {LIB_PATH}: {exc}") from exc
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gcnn_abi_version() != ABI_VERSION:
            raise GcnnError(f"ABI mismatch: library {handle.gcnn_abi_version()} vs binding {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "bad argument", -2: "workspace too small", -3: "HIP call failed", -4: "unsupported size"}.get(rc, f"hipError_t {rc}" if rc > 0 else f"code {rc}")
        raise GcnnError(f"{what} failed: {kind}")


class launch_profile:
    """`with launch_profile() as prof: ...` -> prof.launches = [(kernel name, milliseconds)] of every library launch inside."""

    def __enter__(self):
        check(lib().gcnn_profile_begin(), "gcnn_profile_begin")
        self.launches = []
        return self

    def __exit__(self, *exc):
        cap = 512
        names, ms = (C.c_char_p * cap)(), (C.c_float * cap)()
        n = lib().gcnn_profile_end(cap, names, ms)
        if n < 0:
            check(n, "gcnn_profile_end")
        self.launches = [(names[i].decode(), float(ms[i])) for i in range(min(n, cap))]
        return False


def param_layout():
    """[(offset, rows, cols, trainable)] for the 62 checkpoint arrays, and the flat buffer size, from the library."""
    handle = lib()
    out = []
    for i in range(handle.gcnn_param_count()):
        off, rows, cols, tr = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(handle.gcnn_param_info(i, C.byref(off), C.byref(rows), C.byref(cols), C.byref(tr)), "gcnn_param_info")
        out.append((off.value, rows.value, cols.value, bool(tr.value)))
    return out, handle.gcnn_param_total_floats()
