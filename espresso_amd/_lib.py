"""ctypes binding of the C ABI declared in include/espresso_amd.h (libespresso_amd.so).

The prototypes are parsed from the header itself so the binding cannot drift from the ABI.  There
is NO fallback: if the shared library is missing or fails to load, every kernel call raises
(`EspressoAmdLibraryError`) — the product path never silently runs on a CPU/eager substitute.
"""
import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "espresso_amd.h")
LIB_PATH = os.path.join(_HERE, "csrc", "libespresso_amd.so")


class EspressoAmdLibraryError(RuntimeError):
    pass


class EaGemmParams(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p),
        ("B", ctypes.c_void_p),
        ("C", ctypes.c_void_p),
        ("C2", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("resid", ctypes.c_void_p),
        ("aux", ctypes.c_void_p),
        ("M", ctypes.c_int),
        ("N", ctypes.c_int),
        ("K", ctypes.c_int),
        ("batch", ctypes.c_int),
        ("zdiv", ctypes.c_int),
        ("a_kstrided", ctypes.c_int),
        ("b_kstrided", ctypes.c_int),
        ("c_f32", ctypes.c_int),
        ("accumulate", ctypes.c_int),
        ("resid_f32", ctypes.c_int),
        ("act", ctypes.c_int),
        ("lda", ctypes.c_long),
        ("ldb", ctypes.c_long),
        ("ldc", ctypes.c_long),
        ("ldc2", ctypes.c_long),
        ("ldr", ctypes.c_long),
        ("ldaux", ctypes.c_long),
        ("sA_hi", ctypes.c_long),
        ("sA_lo", ctypes.c_long),
        ("sB_hi", ctypes.c_long),
        ("sB_lo", ctypes.c_long),
        ("sC_hi", ctypes.c_long),
        ("sC_lo", ctypes.c_long),
        ("sR_hi", ctypes.c_long),
        ("sR_lo", ctypes.c_long),
        ("sX_hi", ctypes.c_long),
        ("sX_lo", ctypes.c_long),
        ("alpha", ctypes.c_float),
        ("out_scale", ctypes.c_float),
        ("drop_seed", ctypes.c_uint64),
        ("drop_thr", ctypes.c_uint32),
        ("drop_scale", ctypes.c_float),
        ("splitk", ctypes.c_int),
        ("kchunk", ctypes.c_int),
        ("workspace", ctypes.c_void_p),
        ("q_u", ctypes.c_void_p),
        ("q_v", ctypes.c_void_p),
        ("pos_u", ctypes.c_void_p),
        ("pos_v", ctypes.c_void_p),
        ("ld_q", ctypes.c_long),
        ("qsplit_n", ctypes.c_int),
        ("qscale", ctypes.c_float),
    ]


_F, _V = ctypes.c_void_p, ctypes.c_void_p


def _mk(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": [(f, ctypes.c_void_p) for f in fields]})


EaFfnParams = _mk("EaFfnParams", ["ln_g", "ln_b", "w1", "b1", "w2", "b2"])
EaFfnGrads = _mk("EaFfnGrads", ["ln_g", "ln_b", "w1", "b1", "w2", "b2"])
EaAttnParams = _mk("EaAttnParams", ["ln_g", "ln_b", "wqkv", "bqkv", "wo", "bo", "pos_u", "pos_v", "wpos"])
EaAttnGrads = _mk("EaAttnGrads", ["ln_g", "ln_b", "wqkv", "bqkv", "wo", "bo", "pos_u", "pos_v", "wpos"])
EaConvParams = _mk("EaConvParams", ["ln_g", "ln_b", "pw1", "dw", "bn_g", "bn_b", "bn_rm", "bn_rv", "pw2"])
EaConvGrads = _mk("EaConvGrads", ["ln_g", "ln_b", "pw1", "dw", "bn_g", "bn_b", "pw2"])


class EaLayerGrads(ctypes.Structure):
    _fields_ = [("ffn1", EaFfnGrads), ("attn", EaAttnGrads), ("conv", EaConvGrads), ("ffn2", EaFfnGrads),
                ("final_ln_g", ctypes.c_void_p), ("final_ln_b", ctypes.c_void_p)]


class EaConformerLayer(ctypes.Structure):
    _fields_ = [("ffn1", EaFfnParams), ("attn", EaAttnParams), ("conv", EaConvParams), ("ffn2", EaFfnParams),
                ("final_ln_g", ctypes.c_void_p), ("final_ln_b", ctypes.c_void_p), ("grads", EaLayerGrads), ("wt", ctypes.c_void_p)]


class EaLayerShape(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("T", ctypes.c_int), ("C", ctypes.c_int), ("H", ctypes.c_int), ("F", ctypes.c_int),
                ("KW", ctypes.c_int), ("training", ctypes.c_int), ("p_drop", ctypes.c_float), ("p_act", ctypes.c_float),
                ("p_attn", ctypes.c_float), ("seed", ctypes.c_uint64), ("has_attn_mask", ctypes.c_int), ("scratch_clean", ctypes.c_int),
                ("pos_mode", ctypes.c_int), ("act", ctypes.c_int), ("S", ctypes.c_int), ("defer", ctypes.c_int), ("wt_fresh", ctypes.c_int)]


class EaLayerChain(ctypes.Structure):
    _fields_ = [("next", ctypes.c_void_p), ("next_saved", ctypes.c_void_p), ("next_saved_bytes", ctypes.c_long), ("ln1_done", ctypes.c_int),
                ("prev", ctypes.c_void_p), ("prev_saved", ctypes.c_void_p), ("prev_saved_bytes", ctypes.c_long),
                ("prev_seed", ctypes.c_uint64), ("prev_pre", ctypes.c_void_p), ("final_ln_done", ctypes.c_int), ("pre_in", ctypes.c_void_p)]


class EaStackLayer(ctypes.Structure):
    _fields_ = [("layer", ctypes.c_void_p), ("shape", EaLayerShape), ("saved", ctypes.c_void_p), ("saved_bytes", ctypes.c_long),
                ("x_in", ctypes.c_void_p)]


class EaWgradProblem(ctypes.Structure):
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("dbias", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("ld_dy", ctypes.c_long), ("ld_x", ctypes.c_long),
                ("ldw", ctypes.c_long)]


class EaWgradGroup(ctypes.Structure):
    _fields_ = [("count", ctypes.c_int), ("p", EaWgradProblem * 16)]


EaXAttnParams = _mk("EaXAttnParams", ["ln_g", "ln_b", "wq", "bq", "wkv", "bkv", "wo", "bo"])
EaXAttnGrads = _mk("EaXAttnGrads", ["ln_g", "ln_b", "wq", "bq", "wkv", "bkv", "wo", "bo"])


class EaDecoderLayer(ctypes.Structure):
    _fields_ = [("self_attn", EaAttnParams), ("cross", EaXAttnParams), ("ffn", EaFfnParams), ("g_self", EaAttnGrads),
                ("g_cross", EaXAttnGrads), ("g_ffn", EaFfnGrads), ("wt", ctypes.c_void_p)]


_SCALARS = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "float": ctypes.c_float,
    "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "ea_stream_t": ctypes.c_void_p,
}


def _ctype_of(decl: str):
    decl = decl.strip()
    if "*" in decl:
        if "EaGemmParams" in decl:
            return ctypes.POINTER(EaGemmParams)
        return ctypes.c_void_p
    toks = [t for t in decl.replace("const", " ").split() if t]
    # last token is the parameter name (if any)
    ty = toks[0]
    if ty not in _SCALARS:
        raise ValueError(f"unknown C type in header: {decl!r}")
    return _SCALARS[ty]


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    """Return {symbol: (restype, [argtypes])} for every `ea_*` function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long|uint64_t)\s+(ea_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        args = " ".join(args.split())
        if args in ("void", ""):
            argtypes = []
        else:
            argtypes = [_ctype_of(a) for a in args.split(",")]
        protos[name] = (_SCALARS[ret], argtypes)
    return protos


_lib = None


def lib():
    """Load (once) and return the shared library with prototypes attached.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EspressoAmdLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  espresso_amd has no CPU/eager fallback."
        )
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the box
        raise EspressoAmdLibraryError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(handle, name, None)
        if fn is None:
            raise EspressoAmdLibraryError(f"{LIB_PATH} does not export {name} (declared in espresso_amd.h)")
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"espresso_amd kernel {what} failed with code {rc}")
