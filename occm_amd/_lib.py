"""ctypes binding of libocc_hip.so.  Prototypes are read from include/occ_hip.h so the header is the
single source of truth for the ABI.  Loading fails loudly: the product has no fallback path."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCC_LIB (developer switch): another build of the same ABI next to the default one, for A/B timing of kernel variants on one device
LIB_PATH = os.path.join(_HERE, os.environ.get("OCC_LIB") or "libocc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "occ_hip.h")

OCC_F32, OCC_BF16, OCC_F64, OCC_F32_AS_BF16, OCC_AF32_WBF16, OCC_FP8_E4M3, OCC_FP8_E5M2, OCC_F32X3 = 0, 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_GELU, ACT_SELU, ACT_RELU, ACT_TANH, ACT_GELU_GRAD, ACT_GELU_KEEP_GRAD, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6, 7


class OccError(RuntimeError):
    pass


class RowMap(ctypes.Structure):
    _fields_ = [("rows_per_batch", ctypes.c_int64), ("batch_stride", ctypes.c_int64), ("row_stride", ctypes.c_int64),
                ("rows_per_line", ctypes.c_int64), ("line_stride", ctypes.c_int64)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
                ("A", ctypes.c_void_p), ("a_map", RowMap), ("a_nseg", ctypes.c_int64),
                ("a_seg_len", ctypes.c_int64), ("a_seg_stride", ctypes.c_int64),
                ("W", ctypes.c_void_p), ("ldw", ctypes.c_int64),
                ("bias", ctypes.c_void_p),
                ("R", ctypes.c_void_p), ("r_map", RowMap), ("r_dtype", ctypes.c_int),
                ("C", ctypes.c_void_p), ("c_map", RowMap), ("c_dtype", ctypes.c_int),
                ("ab_dtype", ctypes.c_int), ("act", ctypes.c_int), ("alpha", ctypes.c_float),
                ("n_groups", ctypes.c_int64), ("a_group_stride", ctypes.c_int64),
                ("w_group_stride", ctypes.c_int64), ("c_group_stride", ctypes.c_int64), ("aux", ctypes.c_void_p),
                ("a_dequant", ctypes.c_void_p), ("w_dequant", ctypes.c_void_p),
                ("c_f8", ctypes.c_void_p), ("c_f8_scale", ctypes.c_void_p), ("c_f8_amax", ctypes.c_void_p), ("c_f8_fmt", ctypes.c_int),
                ("c_colsum", ctypes.c_void_p), ("c_colsum_ws", ctypes.c_void_p), ("c_colsum_ws_floats", ctypes.c_int64), ("c_colsum_defer", ctypes.c_int)]


class FinalizeJob(ctypes.Structure):
    _fields_ = [("partials", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p), ("out2", ctypes.c_void_p),
                ("kind", ctypes.c_int32), ("n0", ctypes.c_int32), ("n1", ctypes.c_int32), ("n2", ctypes.c_int32), ("first_block", ctypes.c_int32),
                ("n_blocks", ctypes.c_int32)]


class GemmTnDesc(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int64), ("N1", ctypes.c_int64), ("N2", ctypes.c_int64),
                ("A", ctypes.c_void_p), ("a_map", RowMap),
                ("B", ctypes.c_void_p), ("b_map", RowMap), ("b_nseg", ctypes.c_int64), ("b_seg_len", ctypes.c_int64),
                ("b_seg_stride", ctypes.c_int64),
                ("C", ctypes.c_void_p), ("ldc", ctypes.c_int64), ("alpha", ctypes.c_float), ("colsum", ctypes.c_void_p),
                ("a_dtype", ctypes.c_int), ("b_dtype", ctypes.c_int), ("compute", ctypes.c_int),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
                ("n_groups", ctypes.c_int64), ("a_group_stride", ctypes.c_int64), ("b_group_stride", ctypes.c_int64), ("c_group_stride", ctypes.c_int64),
                ("c_is_zero", ctypes.c_int)]


_P = ctypes.c_void_p


class MasterDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64), ("Do", ctypes.c_int64),
                ("x", _P), ("master", _P), ("master_bstride", ctypes.c_int64),
                ("att_projM_w", _P), ("att_projM_b", _P), ("att_weightM", _P), ("proj_with_attM_w", _P), ("proj_with_attM_b", _P),
                ("proj_without_attM_w", _P), ("proj_without_attM_b", _P),
                ("inv_temp", ctypes.c_float), ("out", _P), ("am", _P), ("agg", _P)]


class MasterGrads(ctypes.Structure):
    _fields_ = [("dout", _P), ("dx", _P), ("dx_accumulate", ctypes.c_int), ("dmaster", _P), ("dmaster_bstride", ctypes.c_int64),
                ("d_att_projM_w", _P), ("d_att_projM_b", _P), ("d_att_weightM", _P), ("d_proj_with_attM_w", _P),
                ("d_proj_with_attM_b", _P), ("d_proj_without_attM_w", _P), ("d_proj_without_attM_b", _P)]


class ReadoutDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int64), ("Nt", ctypes.c_int64), ("Ns", ctypes.c_int64), ("Dg", ctypes.c_int64), ("n_classes", ctypes.c_int64),
                ("T1", _P), ("T2", _P), ("S1", _P), ("S2", _P), ("M1", _P), ("M2", _P),
                ("mask_T1", _P), ("mask_T2", _P), ("mask_S1", _P), ("mask_S2", _P), ("mask_M1", _P), ("mask_M2", _P), ("mask_last", _P),
                ("p_way", ctypes.c_float), ("p_last", ctypes.c_float),
                ("out_w", _P), ("out_b", _P), ("emb", _P), ("logits", _P)]


class ReadoutGrads(ctypes.Structure):
    _fields_ = [("demb", _P), ("dlogits", _P), ("dT1", _P), ("dT2", _P), ("dS1", _P), ("dS2", _P), ("dM1", _P), ("dM2", _P),
                ("d_out_w", _P), ("d_out_b", _P)]


_SCALARS = [("uint64_t", ctypes.c_uint64), ("int64_t", ctypes.c_int64), ("int32_t", ctypes.c_int32),
            ("double", ctypes.c_double), ("float", ctypes.c_float), ("int", ctypes.c_int)]


def parse_header(path=HEADER_PATH):
    """Returns {name: (restype, [argtypes])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef\s+struct.*?}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"enum\s+\w+\s*{.*?}\s*;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(occ_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    continue
                for key, ct in _SCALARS:
                    if re.search(r"\b%s\b" % key, a):
                        argtypes.append(ct)
                        break
                else:
                    raise OccError("cannot map C argument %r of %s" % (a, name))
        protos[name] = (restype, argtypes)
    return protos


_lib = None


def lib():
    """The loaded library; raises OccError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OccError("libocc_hip.so is missing at %s -- build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C occm_amd/csrc`.  occm_amd has no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(handle, name)          # AttributeError -> header/library mismatch, surfaces loudly
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().occ_last_error()
        raise OccError("%s failed (%d): %s" % (what or "libocc_hip call", rc, msg.decode() if msg else "?"))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise OccError("occm_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and there is no CPU fallback")


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host, for *_host arguments) pointer of a contiguous tensor / numpy array, or NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def dtype_code(t):
    import torch
    return {torch.float32: OCC_F32, torch.bfloat16: OCC_BF16, torch.float64: OCC_F64}[t.dtype]
