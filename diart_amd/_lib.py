"""ctypes binding of ``libdiart_amd.so`` (the C ABI declared in ``include/diart_amd.h``).

There is no CPU fallback: if the shared object is missing or a call fails this module
raises.  ``load()`` does not need a GPU (the CPU test-suite checks that every declared
symbol is exported); creating a context does.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

# DZ_EXPERIMENTS=1 loads the experiments build (`python -m diart_amd.build --experiments`: the never-default
# kernel generations and the DZ_* switches that select them, include/diart_amd_experiments.h)
EXPERIMENTS = os.environ.get("DZ_EXPERIMENTS", "0") not in ("", "0")
_LIB_PATH = Path(__file__).resolve().parent / ("libdiart_amd_exp.so" if EXPERIMENTS else "libdiart_amd.so")
_lib: Optional[C.CDLL] = None

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
vp = C.c_void_p


class SincNetWeights(C.Structure):
    _fields_ = [("wav_gamma", C.c_float), ("wav_beta", C.c_float)] + [
        (n, vp) for n in ("filt", "in0_g", "in0_b", "w1", "b1", "in1_g", "in1_b",
                          "w2", "b2", "in2_g", "in2_b", "w1_split", "w2_split", "filt_split")]


class SegWeights(C.Structure):
    _fields_ = [("sinc", SincNetWeights), ("wih", vp * 4), ("bih", vp * 4), ("whh", vp * 4),
                ("lin0_w", vp), ("lin0_b", vp), ("lin1_w", vp), ("lin1_b", vp),
                ("cls_w", vp), ("cls_b", vp),
                ("num_classes", C.c_int), ("powerset", C.c_int), ("num_speakers", C.c_int),
                ("wih_split", vp * 4), ("lin0_split", vp), ("lin1_split", vp), ("whh_split", vp * 4), ("lstm_variant", C.c_int), ("wih0_split_kb", vp)]


class EmbWeights(C.Structure):
    _fields_ = [("sinc", SincNetWeights), ("tw", vp * 5), ("tb", vp * 5), ("ts", vp * 5),
                ("th", vp * 5), ("emb_w", vp), ("emb_b", vp), ("dimension", C.c_int),
                ("tw_split", vp * 5), ("tw0_split_kb", vp), ("pool_nearest", C.c_int)]


class Layer(C.Structure):
    _fields_ = [("w", vp), ("b", vp), ("s", vp), ("h", vp), ("wsplit", vp)]


class SeRes2Net(C.Structure):
    _fields_ = [("tdnn1", Layer), ("res", Layer * 7), ("tdnn2", Layer), ("se1", Layer), ("se2", Layer)]


class EcapaWeights(C.Structure):
    _fields_ = [("dft", vp), ("mel", vp), ("block0", Layer), ("ser", SeRes2Net * 3), ("mfa", Layer),
                ("asp_tdnn", Layer), ("asp_wms", vp), ("asp_conv", Layer), ("fc", Layer), ("zeros", vp),
                ("dft_split", vp)]


# name -> (restype, argtypes); must list every function of include/diart_amd.h
SIGNATURES = {
    "dz_last_error": (C.c_char_p, []),
    "dz_version": (C.c_int, []),
    "dz_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "dz_get_option": (C.c_int, [C.c_char_p, c_int_p]),
    "dz_has_experiments": (C.c_int, []),
    "dz_host_pool_set_spin": (C.c_int, [C.c_int]),
    "dz_abi_struct_sizes": (C.c_int, [C.POINTER(C.c_int * 5)]),
    "dz_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "dz_ctx_destroy": (C.c_int, [vp]),
    "dz_range_check": (C.c_int, [vp, C.c_int]),
    "dz_seg_frames_for": (C.c_int, [C.c_int]),
    "dz_emb_frames_for": (C.c_int, [C.c_int]),
    "dz_seg_create": (C.c_int, [vp, C.POINTER(SegWeights), C.c_int, C.c_int, C.POINTER(vp)]),
    "dz_seg_forward": (C.c_int, [vp, vp, C.c_longlong, C.c_int, vp, vp]),
    "dz_seg_forward_osp": (C.c_int, [vp, vp, C.c_longlong, C.c_int, vp, C.c_float, C.c_float, C.c_int, vp, vp]),
    "dz_seg_destroy": (C.c_int, [vp]),
    "dz_emb_create": (C.c_int, [vp, C.POINTER(EmbWeights), C.c_int, C.c_int, C.POINTER(vp)]),
    "dz_emb_forward": (C.c_int, [vp, vp, C.c_longlong, vp, C.c_int, C.c_int, vp, vp]),
    "dz_emb_forward_multi": (C.c_int, [vp, vp, C.c_longlong, vp, C.c_int, C.c_int, C.c_int,
                                       C.c_int, vp, vp]),
    "dz_emb_frames": (C.c_int, [vp, vp, C.c_longlong, C.c_int, vp]),
    "dz_seg_front": (C.c_int, [vp, vp, C.c_longlong, C.c_int, vp]),
    "dz_seg_back": (C.c_int, [vp, C.c_int, vp, C.c_float, C.c_float, C.c_int, vp, vp]),
    "dz_wave_stats_floats": (C.c_int, []),
    "dz_wave_stats": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp, vp]),
    "dz_seg_use_wave_stats": (C.c_int, [vp, vp]),
    "dz_emb_use_wave_stats": (C.c_int, [vp, vp]),
    "dz_emb_pool": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "dz_emb_destroy": (C.c_int, [vp]),
    "dz_ecapa_frames_for": (C.c_int, [C.c_int]),
    "dz_ecapa_create": (C.c_int, [vp, C.POINTER(EcapaWeights), C.c_int, C.c_int, C.POINTER(vp)]),
    "dz_ecapa_forward": (C.c_int, [vp, vp, C.c_longlong, vp, C.c_int, C.c_int, vp, vp]),
    "dz_ecapa_peek": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dz_ecapa_destroy": (C.c_int, [vp]),
    "dz_prof_enable": (C.c_int, [C.c_int]),
    "dz_prof_pause": (C.c_int, [C.c_int]),
    "dz_prof_collect": (C.c_int, []),
    "dz_prof_get": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                              C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "dz_osp": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                         C.c_int, vp, vp]),
    "dz_l2_normalize": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "dz_results_to_host": (C.c_int, [vp, vp, vp, C.c_longlong, vp, vp, C.c_longlong, vp]),
    "dz_cdist_cosine": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "dz_clu_create": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(vp)]),
    "dz_clu_reset": (C.c_int, [vp]),
    "dz_clu_step": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    "dz_clu_step_batch": (C.c_int, [C.POINTER(vp), C.c_int, vp, C.c_int, C.c_int, vp, C.c_int,
                                    vp, vp, C.c_int]),
    "dz_clu_get_centers": (C.c_int, [vp, vp, C.c_int]),
    "dz_clu_get_active": (C.c_int, [vp, vp]),
    "dz_clu_dim": (C.c_int, [vp]),
    "dz_clu_set_state": (C.c_int, [vp, vp, vp, C.c_int]),
    "dz_clu_destroy": (C.c_int, [vp]),
    "dz_lsap": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "dz_ring_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "dz_ring_reset": (C.c_int, [vp]),
    "dz_ring_destroy": (C.c_int, [vp]),
    "dz_ring_push": (C.c_int, [vp, vp, C.c_longlong, C.c_int, vp]),
    "dz_ring_window": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "dz_ring_read": (C.c_int, [vp, vp, vp]),
    "dz_ring_push_rows": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.c_int, vp]),
    "dz_ring_filled_row": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "dz_ring_gather": (C.c_int, [vp, C.POINTER(C.c_int), C.c_int, vp, C.c_longlong, vp]),
    "dz_ring_reset_row": (C.c_int, [vp, C.c_int]),
    "dz_tail_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                 C.c_int, vp, C.POINTER(vp)]),
    "dz_tail_reset": (C.c_int, [vp]),
    "dz_tail_destroy": (C.c_int, [vp]),
    "dz_tail_max_rows": (C.c_int, [vp]),
    "dz_tail_step": (C.c_int, [vp, vp, C.c_double, C.c_double, vp, vp, vp, vp, vp, C.c_int, vp]),
    "dz_tail_step_batch": (C.c_int, [C.POINTER(vp), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                     C.c_int, vp, C.c_int]),
    "dz_file_step_batch": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int,
                                     C.c_int, vp, C.c_double, vp, C.c_int, vp, vp, C.c_int]),
    # kernel-level entry points
    "dz_k_convgemm": (C.c_int, [vp, vp, vp]),
    "dz_k_gemm_f32": (C.c_int, [vp, vp, vp]),
    "dz_k_gemm_split": (C.c_int, [vp, vp, vp]),
    "dz_k_gemm_pre": (C.c_int, [vp, vp, vp]),
    "dz_k_mlp_head": (C.c_int, [vp, vp, C.c_longlong, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_float, C.c_float, vp, vp, vp]),
    "dz_k_seg_head": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float,
                                C.c_float, C.c_int, vp, vp]),
    "dz_k_conv_pool": (C.c_int, [vp, vp, vp]),
    "dz_k_convgemm_ntile": (C.c_int, [C.c_int]),
    "dz_k_wave_stats": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp, vp]),
    "dz_k_sinc_conv0": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp, C.c_float,
                                  C.c_float, vp, vp, vp, vp]),
    "dz_k_sinc_conv0_split": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp, C.c_float,
                                        C.c_float, vp, vp, vp, vp]),
    "dz_k_conv0_split_ntile": (C.c_int, [C.c_int]),
    "dz_k_finalize_norm": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "dz_k_lstm": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "dz_k_lstm_mfma": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "dz_k_lstm_planes": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, C.c_longlong, C.c_int, C.c_int, vp]),
    "dz_k_stats_pool": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int,
                                  C.c_int, vp, C.c_int, vp]),
    "dz_k_powerset": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
}


# include/diart_amd_experiments.h: bound only when the loaded library is the experiments build
EXPERIMENT_SIGNATURES = {
    "dz_k_gemm_g2": (C.c_int, [vp, vp, C.c_int, vp]),
    "dz_k_gemm_g3": (C.c_int, [vp, vp, C.c_int, vp]),
    "dz_k_conv_pool_debug": (C.c_int, [vp]),
    "dz_sinc_conv0_pair": (C.c_int, [vp, vp, vp, C.c_longlong, C.c_int, vp, vp, vp, vp]),
    "dz_k_sinc_conv0_pair": (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp, vp, vp, C.c_float, C.c_float, vp, vp,
                                       vp, vp, vp]),
}


class ConvGemmDesc(C.Structure):
    _fields_ = [(n, vp) for n in ("X", "W", "bias", "e0", "e1", "nscale", "nshift", "Y", "partials")] + [
        (n, C.c_int) for n in ("B", "Tin", "Tout", "Cin", "taps", "dil", "K", "Kpad", "Npad",
                               "Nstore", "ldx", "ldy", "nld", "Tstore")] + [
        ("xbs", C.c_longlong), ("ybs", C.c_longlong), ("norm_on_load", C.c_int), ("epi", C.c_int),
        ("ksplit", C.c_int), ("ysplit", C.c_longlong), ("agroup", C.c_int), ("pad", C.c_int),
        ("X2", vp), ("rowbias", vp), ("Wsplit", vp), ("Xsplit", vp), ("xplane", C.c_longlong),
        ("Ysplit", vp), ("yplane", C.c_longlong), ("npart", vp), ("ngamma", vp), ("nbeta", vp),
        ("npart_tiles", C.c_int), ("npart_T", C.c_int), ("oflag", vp)]


(EPI_BIAS, EPI_BIAS_LEAKY, EPI_BIAS_SIGMOID, EPI_TDNN, EPI_POOL3, EPI_BIAS_RELU, EPI_RELU_BN,
 EPI_RELU_BN_TANH) = range(8)


class DiartAmdError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """dlopen the library and attach prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise DiartAmdError(
                f"{_LIB_PATH} not found: build it with `python -m diart_amd.build` "
                "(hipcc, gfx950).  diart_amd has no CPU fallback.")
        lib = C.CDLL(str(_LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.dz_has_experiments():
            for name, (res, args) in EXPERIMENT_SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
        elif EXPERIMENTS:
            raise DiartAmdError(f"DZ_EXPERIMENTS=1 but {_LIB_PATH} is not an experiments build")
        sizes = (C.c_int * 5)()
        lib.dz_abi_struct_sizes(C.byref(sizes))
        mine = [C.sizeof(t) for t in (SincNetWeights, SegWeights, EmbWeights, EcapaWeights, ConvGemmDesc)]
        if list(sizes) != mine:
            raise DiartAmdError(f"{_LIB_PATH} was built from a different include/diart_amd.h: struct sizes "
                                f"{list(sizes)} (library) vs {mine} (this binding); rebuild it")
        _lib = lib
    return _lib


def experiments() -> bool:
    """Is the loaded library the experiments build?  (The shipped one ignores every kernel-selection switch.)"""
    return bool(load().dz_has_experiments())


def exp_env(name: str, default: str) -> str:
    """A DZ_* switch that only the experiments build honours: `default` in the shipped configuration."""
    return os.environ.get(name, default) if EXPERIMENTS else default


def set_option(name: str, value: int) -> None:
    check(load().dz_set_option(name.encode(), int(value)), "dz_set_option")


def get_option(name: str) -> int:
    v = C.c_int()
    check(load().dz_get_option(name.encode(), C.byref(v)), "dz_get_option")
    return v.value


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dz_last_error().decode("utf-8", "replace")
        raise DiartAmdError(f"{what or 'libdiart_amd'} failed (code {rc}): {msg}")


_contexts = {}


def range_check(device_index: int, reset: bool = True) -> None:
    """Raise if a split-f16 kernel of this GPU's context met an operand outside +-65504 (it was
    clamped) since the last check.  The caller has synchronised the stream(s) that did the work."""
    ctx = _contexts.get(device_index)
    if ctx is not None:
        check(load().dz_range_check(ctx, int(reset)), "f16x3 operand range")


def context(device_index: int) -> vp:
    """One dz_ctx per (process, GPU)."""
    if device_index not in _contexts:
        h = vp()
        check(load().dz_ctx_create(int(device_index), C.byref(h)), "dz_ctx_create")
        _contexts[device_index] = h
    return _contexts[device_index]
