"""MagCache — host mirror of the reference module (kandinsky/magcache_utils.py).

Same entry point and arguments (`set_magcache_params(dit, mag_ratios, num_steps, no_cfg)`, called by
`get_T2V_pipeline(magcache=True)`, reference utils.py:107-113).  The reference swaps `DiffusionTransformer3D.forward`
for `magcache_forward` class-wide; here the decision state machine and the cached residuals live in the engine
(`k5_dit_set_magcache`, include/k5.h), so both `dit(...)` and the fused `dit.sample(...)` loop honour it.  This module
only prepares the ratio table exactly as the reference does (two leading 1.0, nearest-index interpolation with
numpy's round-half-to-even when the checkpoint's table was calibrated for another step count).
"""
import ctypes as C

import numpy as np

from . import _engine as E


def nearest_interp(src_array, target_length):
    """reference magcache_utils.py:6-13"""
    src_array = np.asarray(src_array)
    src_length = len(src_array)
    if target_length == 1:
        return np.array([src_array[-1]])
    scale = (src_length - 1) / (target_length - 1)
    mapped_indices = np.round(np.arange(target_length) * scale).astype(int)
    return src_array[mapped_indices]


def ratio_table(mag_ratios, num_steps):
    """reference magcache_utils.py:28-39: [1, 1] + ratios, re-sampled per cond / uncond half to 2*num_steps entries."""
    table = np.array([1.0] * 2 + list(mag_ratios), dtype=np.float64)
    if len(table) != num_steps * 2:
        print(f'interpolate MAG RATIOS: curr len {len(table)}')
        con = nearest_interp(table[0::2], num_steps)
        ucon = nearest_interp(table[1::2], num_steps)
        table = np.concatenate([con.reshape(-1, 1), ucon.reshape(-1, 1)], axis=1).reshape(-1)
    return np.ascontiguousarray(table, dtype=np.float64)


def _apply(dit):
    """push the stored parameters into the engine handle (called when the handle is (re)built)"""
    if getattr(dit, "mag_ratios", None) is None or dit._handle is None:
        return
    t = dit.mag_ratios
    E.check(E.lib().k5_dit_set_magcache(dit._handle, t.ctypes.data_as(C.POINTER(C.c_double)), len(t), int(dit.no_cfg),
                                        float(dit.magcache_thresh), int(dit.K), float(dit.retention_ratio)), "set_magcache")
    cfgp = getattr(dit, "_cfg_parallel", None)
    if cfgp is not None and not dit.no_cfg:   # this rank group runs one CFG branch only: calls branch, branch+2, ...
        E.check(E.lib().k5_dit_magcache_calls(dit._handle, int(cfgp[0]), 2), "magcache_calls")


def set_magcache_params(dit, mag_ratios, num_steps, no_cfg):
    """reference magcache_utils.py:16-39 (same attribute names on `dit`)."""
    print('using Magcache')
    dit.num_steps = num_steps * 2
    dit.magcache_thresh = 0.12
    dit.K = 2
    dit.retention_ratio = 0.2
    dit.mag_ratios = ratio_table(mag_ratios, num_steps)
    dit.no_cfg = no_cfg
    _apply(dit)


def disable_magcache(dit):
    dit.mag_ratios = None
    if dit._handle is not None:
        E.check(E.lib().k5_dit_set_magcache(dit._handle, None, 0, 0, 0.12, 2, 0.2))


def magcache_state(dit):
    """(call counter, forwards that ran the visual blocks, forwards that skipped them) since set_magcache_params"""
    cnt, ran, skipped = C.c_int(), C.c_int64(), C.c_int64()
    E.check(E.lib().k5_dit_magcache_state(dit._handle, C.byref(cnt), C.byref(ran), C.byref(skipped)))
    return cnt.value, ran.value, skipped.value
