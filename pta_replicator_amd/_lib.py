"""ctypes binding of libpta_replicator_amd.so (the C ABI declared in include/pta_replicator_amd.h).

There is no CPU fallback: if the shared library is missing the import fails loudly, and every entry
point raises ``PtaError`` with the library's own message on a non-zero return code.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p

# torch must own the HIP runtime of the process: it ships its own libamdhip64.so.7 and our library must
# bind to that same copy (same SONAME) so that tensors' device pointers and streams are valid inside it.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTA_REPLICATOR_AMD_LIB: an explicitly named build of the SAME library (probe builds with diagnostic defines: scripts/probe_src/) - never a fallback
LIB_PATH = os.environ.get("PTA_REPLICATOR_AMD_LIB") or os.path.join(_HERE, "libpta_replicator_amd.so")


class PtaError(RuntimeError):
    """A call into libpta_replicator_amd.so returned a PTA_E_* code."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP library first (python -m pta_replicator_amd.build, or "
        "__graft_entry__.build()). pta_replicator_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p  # device / host raw pointers are passed as integers


class EnginePlan(ctypes.Structure):
    """pta_engine_plan (include/pta_replicator_amd.h)."""
    _fields_ = [
        ("n_toa", c_int32), ("n_psr", c_int32), ("rn_k", c_int32), ("gw_npts", c_int32), ("tnequad", c_int32),
        ("n_tiles", c_int32),
        ("tile_psr", _P), ("tile_start", _P), ("tile_count", _P), ("tile_ep0", _P), ("tile_epn", _P),
        ("idx_in_psr", _P), ("Ft", _P), ("ldf", c_int64), ("rn_coef", _P), ("gw_G", _P),
        ("gw_jlo", _P), ("gw_w", _P), ("wn_a", _P), ("wn_b", _P), ("epoch_of", _P),
        ("ecorr_toa", _P), ("det", _P), ("wn_c", _P), ("rng_fast", c_int32), ("synth_variant", c_int32),
    ]


class EngineTables(ctypes.Structure):
    """pta_engine_tables (include/pta_replicator_amd.h)."""
    _fields_ = [
        ("rn_amp", _P), ("Mchol", _P), ("gw_nf", c_int32), ("gw_i0", c_int32), ("use_czt", c_int32), ("czt_variant", c_int32),
        ("czt_pre", _P), ("czt_FB", _P), ("czt_tw", _P), ("czt_post", _P), ("Tsym", _P), ("rot", _P),
        ("ws_coef", _P), ("ws_G0", _P), ("ws_G", _P), ("idft_variant", c_int32), ("mix_variant", c_int32),
    ]


class TdPlan(ctypes.Structure):
    """pta_td_plan (include/pta_replicator_amd.h)."""
    _fields_ = [
        ("Lbase", _P), ("blk_pos", _P), ("blk_ld", _P), ("blk_n", _P), ("blk_off", _P), ("item_blk", _P), ("item_n0", _P),
        ("n_blocks", c_int32), ("n_items", c_int32), ("rows_per_real", c_int32), ("stream_kind", c_uint32),
        ("rng_fast", c_int32), ("gw_npts", c_int32),
        ("gw_G", _P), ("gw_jlo", _P), ("gw_w", _P), ("det", _P), ("z", _P), ("ld_z", c_int64), ("blk_zoff", _P), ("item_rows", _P),
    ]


_SIGNATURES = {
    "pta_abi_version": (c_int, []),
    "pta_last_error": (c_char_p, []),
    "pta_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "pta_rng_philox_raw": (c_int, [_P, _P, c_int, _P, _P]),
    "pta_rng_fill_normal": (c_int, [c_uint64, c_uint64, c_int, c_uint32, c_int, c_int, _P, _P, c_int64, c_int, _P]),
    "pta_rng_fill_normal_blocks": (c_int, [c_uint64, c_uint64, c_int, c_uint32, c_int, _P, _P, c_int, _P, c_int64, c_int, _P]),
    "pta_rn_basis": (c_int, [_P, c_int, c_double, _P, _P, c_int, c_int, _P, c_int64, _P]),
    "pta_rn_synth": (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, c_int, _P, c_int64, c_int, _P]),
    "pta_wn": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, c_int64, c_int, _P, c_int64, c_int, _P]),
    "pta_quantize_epochs": (c_int, [_P, c_int, c_double, _P, _P, _P, POINTER(c_int)]),
    "pta_dot3_host": (c_int, [_P, c_int64, _P, _P]),
    "pta_pow_host": (c_int, [_P, c_double, c_int64, _P]),
    "pta_legacy_randn": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int]),
    "pta_ecorr": (c_int, [_P, _P, c_int, c_int, _P, c_int64, c_int, _P, c_int64, c_int, _P]),
    "pta_orf_pair_arguments": (c_int, [_P, c_int, _P, _P]),
    "pta_orf_hd": (c_int, [_P, c_int, _P, _P]),
    "pta_orf_basis": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "pta_orf_combine": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "pta_potrf_batched": (c_int, [_P, c_int, c_int, _P, _P]),
    "pta_potrf_batched_ex": (c_int, [_P, c_int, c_int64, c_int64, c_int, _P, c_int, _P]),
    "pta_potrf_workspace_doubles": (c_int64, [c_int, c_int, c_int]),
    "pta_potrf_batched_ws": (c_int, [_P, c_int, c_int64, c_int64, c_int, _P, c_int, _P, c_int64, _P]),
    "pta_potrf_warmup": (c_int, [c_int]),
    "pta_td_assemble_potrf": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int, _P, c_int, _P, c_int64, _P]),
    "pta_potrf_ragged_plan_words": (c_int64, [c_int]),
    "pta_potrf_ragged_plan": (c_int, [_P, _P, _P, c_int, c_int, _P, POINTER(c_int64)]),
    "pta_potrf_ragged": (c_int, [_P, _P, _P, _P, _P, c_int64, _P]),
    "pta_gwb_twiddle": (c_int, [_P, c_int, c_int, c_int, c_double, _P, c_int64, _P]),
    "pta_gwb_idft": (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, c_int, _P, c_int64, c_int, _P]),
    "pta_gwb_twiddle_sym_size": (c_int64, [c_int, c_int, c_int, POINTER(c_int64)]),
    "pta_gwb_twiddle_sym": (c_int, [_P, c_int, c_int, c_int, c_double, _P, _P, c_int, _P]),
    "pta_gwb_idft_rng": (c_int, [c_uint64, c_uint64, c_int, c_int, c_int, _P, _P, c_int, _P, c_int64, c_int, c_int, _P]),
    "pta_gwb_czt_fits": (c_int, [c_int, c_int, c_int]),
    "pta_gwb_czt_setup": (c_int, [_P, c_int, c_int, c_int, c_double, _P, _P, _P, _P, _P]),
    "pta_gwb_czt": (c_int, [c_uint64, c_uint64, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "pta_gwb_mix": (c_int, [_P, c_int, _P, c_int, c_int, c_int64, _P, c_int, _P]),
    "pta_gwb_bracket": (c_int, [_P, c_int, _P, c_int, _P, _P]),
    "pta_gwb_weights": (c_int, [_P, c_int, _P, _P, c_int, _P, _P]),
    "pta_gwb_interp": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_double, _P, c_int64, c_int, _P]),
    "pta_cgw": (c_int, [_P, c_int, _P, _P, c_int, _P]),
    "pta_cw_catalog_workspace": (c_int, [c_int, c_int, POINTER(c_int64), POINTER(c_int)]),
    "pta_cw_catalog": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_double, _P, _P, c_int, _P]),
    "pta_engine_rn_coef": (c_int, [c_uint64, c_uint64, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "pta_engine_generate": (c_int, [POINTER(EnginePlan), POINTER(EngineTables), c_uint64, c_uint64, c_int, _P, c_int64, _P]),
    "pta_engine_synth": (c_int, [POINTER(EnginePlan), c_uint64, c_uint64, c_int, _P, c_int64, _P]),
    "pta_td_cov_assemble": (c_int, [_P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_int64, _P]),
    "pta_td_cov_assemble_all": (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "pta_td_cov_walk_items": (c_int64, [_P, c_int, c_int, c_int, _P]),
    "pta_td_cov_assemble_walk": (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int64, _P, c_int, _P]),
    "pta_td_trmm": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, c_int64, c_int, c_int, _P]),
    "pta_td_trmm_rng": (c_int, [POINTER(TdPlan), c_uint64, c_uint64, c_int, _P, c_int64, _P]),
    "pta_tm_project": (c_int, [_P, _P, c_int64, c_int, _P, c_int, _P, c_int64, c_int, _P]),
    "pta_gather_rank0": (c_int, [_P, c_int, c_int, c_int, _P, c_int64, c_int64, c_int64, _P, c_int64, _P]),
    "pta_dgemm": (c_int, [c_int, c_int, c_int, c_int, c_double, _P, c_int64, c_int64, _P, c_int64, c_double, _P, c_int64,
                          c_int, c_int, c_int64, c_int64, c_int64, c_int, _P]),
    "pta_microbench": (c_int, [c_int, c_int64, c_int, c_int, POINTER(c_double)]),
    "pta_clock_probe": (c_int, [_P, c_int, c_int, c_int, _P]),
    "pta_selftest_mfma_f64": (c_int, [POINTER(c_double)]),
}

ENGINE_TILE = 256   # PTA_ENGINE_TILE
ENGINE_EPMAX = 132  # PTA_ENGINE_EPMAX
TD_STRIP = 256      # PTA_TD_STRIP
POTRF_ZERO_UPPER, POTRF_NO_LOOKAHEAD, POTRF_SUBSTITUTION, POTRF_VALU, POTRF_REG_STAGING, POTRF_LOCKSTEP, POTRF_DIAG_AHEAD, POTRF_DIAG64 = 1, 2, 4, 8, 32, 64, 128, 16
POTRF_EPI1 = 0x100000
POTRF_LEFT, POTRF_LEFT_SPLIT, POTRF_SOLVE_ROWS = 0x200000, 0x400000, 0x800000


def POTRF_CHAINS(c):
    """PTA_POTRF_CHAINS(c): number of concurrent chains of matrices of pta_potrf_batched_ex."""
    return (int(c) & 0xF) << 16


def POTRF_NB(k):
    """PTA_POTRF_NB(k): panel width override of pta_potrf_batched_ex, k * 256 columns."""
    return (int(k) & 0xFF) << 8

EXPORTS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so is stale: rebuild
    _fn.restype = _res
    _fn.argtypes = _args

if lib.pta_abi_version() != 8:
    raise ImportError("libpta_replicator_amd.so has an unexpected ABI version: rebuild it")


def last_error():
    msg = lib.pta_last_error()
    return msg.decode() if msg else ""


def check(rc, what=""):
    if rc != 0:
        raise PtaError(f"{what or 'libpta_replicator_amd'} failed (code {rc}): {last_error()}")


def call(name, *args):
    """Call an int-returning entry point and raise on error."""
    check(getattr(lib, name)(*args), name)
