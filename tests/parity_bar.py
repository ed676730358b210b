"""The parity bar of the random-scene sweeps, in one place (tests/test_fuzz_scenes.py, tools/fuzz_gpu.py).

north_star states an ABSOLUTE bar: max |pixel - reference| <= 1e-4. The fixed-scene tests and bench.py's `parity` object use exactly that.
The fuzz generators also produce unclamped radiances far above 1 (non-unit quaternions scale normals, light colours up to 1e3: pixels
reach 1e13 and beyond), where one ulp is already more than 1e-4: there -- and only there -- the bar is relative,
|pixel - reference| <= 1e-4 * |reference|. So: absolute 1e-4 wherever |reference| <= 1, relative above; NaN and inf in the same places.
`judge` also COUNTS the channel values that needed the relative part (|reference| > 1 and an absolute difference above 1e-4), so that a
sweep can say how much of its verdict rests on the exception (VERDICT r3, weak #1c)."""
import numpy as np

TOL = 1e-4


def judge(img, ref):
    """-> dict(ok, worst: max of |d| / max(1, |ref|), special_mismatch: NaN / inf in different places, needed_relative: values with
    |ref| > 1 whose absolute difference exceeds 1e-4 (inside the relative bar or not), above_one: values with |ref| > 1, values)"""
    special = int((np.isnan(img) ^ np.isnan(ref)).sum()) + int((np.isinf(img) ^ np.isinf(ref)).sum())
    fin = np.isfinite(img) & np.isfinite(ref)
    with np.errstate(invalid="ignore", over="ignore"):
        d = np.abs(np.where(fin, img - ref, 0.0))
        mag = np.abs(np.where(fin, ref, 0.0))
        worst = float((d / np.maximum(1.0, mag)).max()) if d.size else 0.0
        needed = int(((mag > 1.0) & (d > TOL)).sum())
        above = int((mag > 1.0).sum())
    return dict(ok=special == 0 and worst <= TOL, worst=worst, special_mismatch=special, needed_relative=needed, above_one=above, values=int(d.size))
