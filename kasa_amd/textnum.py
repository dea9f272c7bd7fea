"""Number -> text exactly as the reference prints per-read results.

The reference formats doubles with Milo Yip's Grisu2 `dtoa` (source/utils/dToStr.h:427-456, algorithm:
F. Loitsch, "Printing floating-point numbers quickly and accurately with integers", PLDI 2010) and
integers with a plain itoa (source/utils/iToStr.hpp:35-114).  Grisu2 is *not* always the shortest
round-trip representation, so `repr(float)` cannot stand in for it: this is a from-scratch Grisu2 with
the same cached-power grid (10^-348 .. 10^340, step 8), the same rounding of the DiyFp product and the
same "prettify" rules.  The cached powers are derived here with exact integer arithmetic instead of
being a pasted table; tests/test_textnum.py checks them against the published constants and checks the
formatter against every number the reference binary printed into the golden fixtures.
"""
from __future__ import annotations

import math
import struct

_MASK64 = (1 << 64) - 1


def _cached_power(e10: int):
    """64-bit normalised significand f and binary exponent e with 10^e10 ~= f * 2^e (round to nearest)."""
    num, den = (10 ** e10, 1) if e10 >= 0 else (1, 10 ** (-e10))
    e = num.bit_length() - den.bit_length() - 64
    while True:
        vn, vd = (num, den << e) if e >= 0 else (num << (-e), den)
        q, rem = divmod(vn, vd)
        if q >= (1 << 64):
            e += 1
            continue
        if q < (1 << 63):
            e -= 1
            continue
        if 2 * rem >= vd:
            q += 1
            if q == (1 << 64):
                q >>= 1
                e += 1
        return q, e


_POWERS = [_cached_power(-348 + 8 * i) for i in range(87)]
_POW10 = [1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000]


def _mul(a, b):
    """DiyFp product: upper 64 bits of the 128-bit product, rounded half up."""
    p = a[0] * b[0]
    h = p >> 64
    if p & (1 << 63):
        h += 1
    return (h, a[1] + b[1] + 64)


def _normalize(f, e):
    s = 64 - f.bit_length()
    return (f << s, e - s)


def _count_digits32(n):
    if n < 10: return 1
    if n < 100: return 2
    if n < 1000: return 3
    if n < 10000: return 4
    if n < 100000: return 5
    if n < 1000000: return 6
    if n < 10000000: return 7
    if n < 100000000: return 8
    if n < 1000000000: return 9
    return 10


def _grisu_round(buf, delta, rest, ten_kappa, wp_w):
    while rest < wp_w and delta - rest >= ten_kappa and (
            rest + ten_kappa < wp_w or wp_w - rest > rest + ten_kappa - wp_w):
        buf[-1] -= 1
        rest += ten_kappa


def _grisu2(value: float):
    bits = struct.unpack("<Q", struct.pack("<d", value))[0]
    biased = (bits >> 52) & 0x7FF
    frac = bits & ((1 << 52) - 1)
    if biased != 0:
        f, e = frac | (1 << 52), biased - 1075
    else:
        f, e = frac, -1074
    # normalized boundaries
    pf, pe = (f << 1) + 1, e - 1
    s = 64 - pf.bit_length()
    pl = (pf << s, pe - s)
    if f == (1 << 52):
        mf, me = (f << 2) - 1, e - 2
    else:
        mf, me = (f << 1) - 1, e - 1
    mi = (mf << (me - pl[1]), pl[1])
    # cached power
    dk = (-61 - pl[1]) * 0.30102999566398114 + 347
    k = int(dk)
    if k != dk:
        k += 1
    index = (k >> 3) + 1
    K = -(-348 + (index << 3))
    c = _POWERS[index]
    W = _mul(_normalize(f, e), c)
    Wp = _mul(pl, c)
    Wm = _mul(mi, c)
    Wm = (Wm[0] + 1, Wm[1])
    Wp = (Wp[0] - 1, Wp[1])
    delta = Wp[0] - Wm[0]
    # digit generation
    one_e = Wp[1]
    one_f = 1 << (-one_e)
    wp_w = Wp[0] - W[0]
    p1 = (Wp[0] >> (-one_e)) & 0xFFFFFFFF
    p2 = Wp[0] & (one_f - 1)
    kappa = _count_digits32(p1)
    buf = []
    while kappa > 0:
        div = _POW10[kappa - 1]
        d, p1 = divmod(p1, div)
        if d or buf:
            buf.append(d)
        kappa -= 1
        tmp = (p1 << (-one_e)) + p2
        if tmp <= delta:
            K += kappa
            _grisu_round(buf, delta, tmp, _POW10[kappa] << (-one_e), wp_w)
            return buf, K
    while True:
        p2 = (p2 * 10) & _MASK64
        delta = (delta * 10) & _MASK64
        d = p2 >> (-one_e)
        if d or buf:
            buf.append(d)
        p2 &= one_f - 1
        kappa -= 1
        if p2 < delta:
            K += kappa
            # the reference indexes its 10-entry table out of bounds past 10^9 (denormals only); 0 there
            mul = _POW10[-kappa] if -kappa < 10 else 0
            _grisu_round(buf, delta, p2, one_f, (wp_w * mul) & _MASK64)
            return buf, K


def _exponent(k: int) -> str:
    return ("-" + str(-k)) if k < 0 else str(k)


def dtoa(value: float) -> str:
    """Text of a double as dToStr.h:427-456 appends it."""
    value = float(value)
    if math.isnan(value):
        return "NaN"
    if math.isinf(value):
        return "inf"
    if value == 0:
        return "0.0"
    sign = ""
    if value < 0:
        sign, value = "-", -value
    digits, k = _grisu2(value)
    s = "".join(chr(48 + d) for d in digits)
    n = len(s)
    kk = n + k
    if n <= kk <= 21:
        out = s + "0" * (kk - n) + ".0"
    elif 0 < kk <= 21:
        out = s[:kk] + "." + s[kk:]
    elif -6 < kk <= 0:
        out = "0." + "0" * (-kk) + s
    elif n == 1:
        out = s + "e" + _exponent(kk - 1)
    else:
        out = s[0] + "." + s[1:] + "e" + _exponent(kk - 1)
    return sign + out


def itoa(v: int) -> str:
    return str(int(v))
