"""ctypes binding of libfithubert_hip.so (include/fithubert_hip.h).

The HIP library is the product: there is NO fallback.  Importing symbols from a missing or
stale library raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FH_LIB_PATH: another build of the SAME interface (same-box A/B of kernel changes, tools/scratch/ab.sh); the loader still checks
# every symbol of include/fithubert_hip.h, and bench.py's `binary` field names the file that was loaded
LIB_PATH = os.environ.get("FH_LIB_PATH") or os.path.join(HERE, "libfithubert_hip.so")

R_CONTIG, R_STRIDED = 0, 1
ACT_NONE, ACT_GELU = 0, 1
DT_F32, DT_BF16, DT_F16 = 0, 1, 2


class Drop(C.Structure):
    _fields_ = [("p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32)]


class GemmOperand(C.Structure):
    _fields_ = [
        ("p", C.c_void_p), ("layout", C.c_int), ("ld", C.c_long), ("rpb", C.c_int), ("bstride", C.c_long),
        ("gstride", C.c_long), ("nseg", C.c_int), ("tshift", C.c_int * 2), ("tvalid", C.c_int),
        ("segoff", C.c_long * 2), ("seg1_limit", C.c_int),
    ]


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("alpha", C.c_float), ("bias", C.c_void_p), ("bias_gstride", C.c_long), ("bias_mod", C.c_int),
        ("preact", C.c_void_p), ("act", C.c_int), ("drop", Drop), ("gelu_aux", C.c_void_p),
        ("residual", C.c_void_p), ("row_len", C.c_void_p), ("out_f32", C.c_int), ("atomic", C.c_int),
        ("idiv", C.c_int), ("si1", C.c_long), ("si0", C.c_long), ("jdiv", C.c_int), ("sj1", C.c_long),
        ("sj0", C.c_long), ("tail_cols", C.c_int), ("aux_factor", C.c_int),
        ("kd_teacher", C.c_void_p), ("kd_tdtype", C.c_int), ("kd_lstride", C.c_long), ("kd_layer", C.c_void_p), ("kd_w", C.c_void_p),
        ("kd_Tt", C.c_int), ("kd_P", C.c_int), ("kd_type", C.c_int), ("kd_grad_scale", C.c_float), ("kd_count", C.c_float),
        ("kd_partials", C.c_void_p),
        ("c0_wave", C.c_void_p), ("c0_w", C.c_void_p), ("c0_gamma", C.c_void_p), ("c0_beta", C.c_void_p), ("c0_mean", C.c_void_p),
        ("c0_rstd", C.c_void_p), ("c0_wpart", C.c_void_p), ("c0_S", C.c_int), ("c0_K", C.c_int), ("c0_stride", C.c_int),
    ]


class GemmDesc(C.Structure):
    _fields_ = [
        ("X", GemmOperand), ("Y", GemmOperand), ("C", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
        ("Kseg", C.c_int), ("ldc", C.c_long), ("c_rpb", C.c_int), ("c_bstride", C.c_long),
        ("c_gstride", C.c_long), ("groups", C.c_int), ("splitk", C.c_int), ("ep", GemmEpilogue),
    ]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -m fithubert_amd.build` "
                "(fithubert_amd has no CPU / eager fallback)")
        # torch owns device memory and streams: its HIP runtime (torch/lib/libamdhip64.so, same SONAME as ROCm's)
        # must be the one this library binds to, so torch is imported before the dlopen.
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long

FH_MAX_CONV = 12


class StudentConfigC(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("n_conv", C.c_int), ("conv_dim", C.c_int * FH_MAX_CONV), ("conv_k", C.c_int * FH_MAX_CONV),
        ("conv_stride", C.c_int * FH_MAX_CONV), ("embed_dim", C.c_int), ("ffn_dim", C.c_int), ("heads", C.c_int),
        ("layers", C.c_int), ("conv_pos", C.c_int), ("conv_pos_groups", C.c_int), ("tr_factor", C.c_int),
        ("pred_dim", C.c_int), ("dropout", C.c_float), ("attention_dropout", C.c_float),
        ("activation_dropout", C.c_float), ("dropout_input", C.c_float), ("feature_grad_mult", C.c_float),
        ("n_tasks", C.c_int), ("inter_dim", C.c_int), ("cnn_head", C.c_int), ("tr_index", C.c_int),
    ]


class StudentGradsC(C.Structure):
    _fields_ = [("d_pred", C.c_void_p), ("d_hidden", C.c_void_p), ("d_cnn", C.c_void_p), ("d_qkv_last", C.c_void_p)]


GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_long, C.c_long)


class GradNotifyC(C.Structure):
    _fields_ = [("comm_stream", C.c_void_p), ("ready", GRAD_READY_FN), ("user", C.c_void_p)]


class StudentKdC(C.Structure):
    _fields_ = [("teacher", C.c_void_p), ("teacher_dtype", C.c_int), ("teacher_lstride", C.c_long), ("teacher_layer", C.c_void_p),
                ("layer_w", C.c_void_p), ("Tt", C.c_int), ("loss_type", C.c_int), ("grad_scale", C.c_float), ("partials", C.c_void_p),
                ("out_losses", C.c_void_p)]


class TeacherLayerC(C.Structure):
    """fh_teacher_layer (include/fithubert_hip.h): device pointers of one teacher layer's weights"""
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "g1", "b1", "w1", "c1", "w2", "c2", "g2", "b2")]


class StudentCallC(C.Structure):
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("training", C.c_int), ("heads_on", C.c_int), ("seed", C.c_uint32)]


_ENGINE_SIGS = {
    "fh_student_param_count": (_L, [C.POINTER(StudentConfigC)]),
    "fh_student_param_info": (_I, [C.POINTER(StudentConfigC), _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_I), C.POINTER(_L)]),
    "fh_student_wc_bytes": (_L, [C.POINTER(StudentConfigC)]),
    "fh_student_ws_bytes": (_L, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC)]),
    "fh_student_ws_info": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), C.c_char_p, C.POINTER(_L), C.POINTER(_I),
                                C.POINTER(_L), C.POINTER(_I)]),
    "fh_student_frames": (_I, [C.POINTER(StudentConfigC), _I, C.POINTER(_I), C.POINTER(_I)]),
    "fh_student_prepare": (_I, [C.POINTER(StudentConfigC), _P, _P, _P]),
    "fh_student_forward": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, _P]),
    "fh_student_kd_partials": (_L, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC)]),
    "fh_student_forward_kd": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, C.POINTER(StudentKdC), _P]),
    "fh_student_backward": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fh_student_backward_cnn": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "fh_student_backward_ex": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, _P, C.POINTER(StudentGradsC), _P]),
    "fh_student_backward_notify": (_I, [C.POINTER(StudentConfigC), C.POINTER(StudentCallC), _P, _P, _P, _P, _P, _P, C.POINTER(StudentGradsC),
                                        C.POINTER(GradNotifyC), _P]),
}


# name -> (restype, argtypes) ; mirrors include/fithubert_hip.h one to one
SIGNATURES = {
    "fh_gemm": (_I, [C.POINTER(GemmDesc), _I, _P]),
    "fh_gemm_set_path": (_I, [_I]),
    "fh_gemm_stream_loops": (_I, [_P, C.c_uint]),
    "fh_gemm_last_loop": (_I, []),
    "fh_gemm_reserve_cus": (_I, [_I]),
    "fh_gemm_stream_reserve_cus": (_I, [_P, _I]),
    "fh_stream_create_cu_range": (_I, [_I, _I, C.POINTER(_P)]),
    "fh_stream_destroy": (_I, [_P]),
    "fh_stream_cu_census": (_I, [_P, _I, _P]),
    "fh_debug_copy_reduce": (_I, [_P, _P, C.c_long, _I, _P]),
    "fh_student_side_stream": (_I, [_P]),
    "fh_gemm_tune": (_I, [_I]),
    "fh_gemm_tune_export": (_I, [_P, _P, _P, _I]),
    "fh_gemm_tune_import": (_I, [_P, _P, _P, _I]),
    "fh_prof_enable": (_I, [_I]),
    "fh_prof_collect": (_I, [_P, _P, _P]),
    "fh_prof_collect_launches": (_I, [_P, _P, _P, _I]),
    "fh_prof_tag": (_I, [_I]),
    "fh_conv0_scratch_floats": (_L, [_I, _I, _I, _I, _I]),
    "fh_conv0_forward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fh_conv0_backward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fh_layernorm_forward": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _F, Drop, _P]),
    "fh_add_layernorm_forward": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "fh_layernorm_gelu_forward": (_I, [_I, _P, _P, _P, _P, _I, _I, _F, _P]),
    "fh_gelu_forward": (_I, [_I, _P, _P, _L, _P]),
    "fh_layernorm_backward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, Drop, Drop, _P]),
    "fh_colsum": (_I, [_I, _P, _P, _I, _I, _L, _I, _L, _L, _I, _F, _P]),
    "fh_posconv_prepare": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fh_posconv_forward": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fh_posconv_backward_data": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fh_posconv_backward_weight": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fh_posconv_weightnorm_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fh_attention_forward": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _F, Drop, _P]),
    "fh_attention_backward": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, Drop, _P]),
    "fh_attention_backward_scratch_floats": (_L, [_I, _I, _I, _I, _I]),
    "fh_attention_backward_ws": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, Drop, _P, _P]),
    "fh_attention_backward_bias": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, Drop, _P, _P]),
    "fh_kd_loss_nblk": (_I, [_I, _I, _I, _I]),
    "fh_kd_loss": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _F, _P]),
    "fh_kd_loss_rows_nblk": (_I, [_I, _I]),
    "fh_kd_loss_rows": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _F, _F, _P]),
    "fh_kd_loss_rows_scaled": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _L, _I, _F, _F, _F, _P]),
    "fh_relation_tensor": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fh_relation_kd_nblk": (_I, [_I, _I, _I]),
    "fh_relation_kd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "fh_conv1d_forward": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fh_linear_forward": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fh_conv0_layernorm_gelu_forward": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fh_teacher_encoder_scratch_bytes": (_L, [_I, _I, _I, _I, _I]),
    "fh_teacher_encoder_forward": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "fh_posconv_wide_prepare": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fh_posconv_wide_forward": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fh_cast": (_I, [_I, _P, _P, _L, _P]),
    "fh_permute3_cast": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fh_adamw": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _F, _F, _P, _P]),
    "fh_adamw_amp": (_I, [_P, _P, _P, _P, _L, _F, _I, _I, _F, _F, _F, _F, _F, _P, _P]),
}


SIGNATURES.update(_ENGINE_SIGS)


def _declare(l: C.CDLL) -> None:
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise HipLibraryMissing(f"{LIB_PATH} is stale: missing symbols {missing}; rebuild with `python -m fithubert_amd.build`")


def drop(p: float = 0.0, seed: int = 0, site: int = 0) -> Drop:
    d = Drop()
    d.p, d.seed, d.site = p, seed, site
    return d


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def operand(t_ptr, layout, ld, rpb=0, bstride=0, gstride=0, nseg=1, tshift=(0, 0), tvalid=0, segoff=(0, 0), seg1_limit=0):
    o = GemmOperand()
    o.p = t_ptr
    o.layout = layout
    o.ld = ld
    o.rpb = rpb
    o.bstride = bstride
    o.gstride = gstride
    o.nseg = nseg
    o.tshift[0], o.tshift[1] = tshift
    o.tvalid = tvalid
    o.segoff[0], o.segoff[1] = segoff
    o.seg1_limit = seg1_limit
    return o


def epilogue(**kw):
    e = GemmEpilogue()
    e.alpha = 1.0
    e.kd_grad_scale = e.kd_count = 1.0
    for k, v in kw.items():
        if k == "drop":
            e.drop.p, e.drop.seed, e.drop.site = v
        else:
            setattr(e, k, v)
    return e


def save_gemm_table(path: str) -> int:
    """Write the per-shape main-loop table (fh_gemm_tune) to ``path`` (JSON); returns the number of entries."""
    import json
    n = lib().fh_gemm_tune_export(None, None, None, 0)
    keys, paths, sks = (C.c_ulonglong * n)(), (C.c_int * n)(), (C.c_int * n)()
    lib().fh_gemm_tune_export(keys, paths, sks, n)
    with open(path, "w") as f:
        json.dump({"entries": [[int(keys[i]), int(paths[i]), int(sks[i])] for i in range(n)]}, f)
    return n


def load_gemm_table(path: str) -> int:
    """Merge a table written by save_gemm_table; returns the table size (0 if the file does not exist)."""
    import json
    import os
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        ent = json.load(f)["entries"]
    n = len(ent)
    if n == 0:
        return 0
    keys, paths, sks = (C.c_ulonglong * n)(*[e[0] for e in ent]), (C.c_int * n)(*[e[1] for e in ent]), (C.c_int * n)(*[e[2] for e in ent])
    return lib().fh_gemm_tune_import(keys, paths, sks, n)
