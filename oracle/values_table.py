"""CPU restatement of the reference's value-table generation -- TEST INFRASTRUCTURE ONLY.

``values_table(nCases, nControls)`` follows ``getValuesTable`` (R/Utils.R:137-159) line by line, with R's
``stats::dhyper`` restated from R's published nmath sources (R 3.x - 4.3: ``dhyper.c``, ``dbinom.c``
``dbinom_raw``, ``stirlerr.c``, ``bd0.c`` -- C. Loader's saddle-point algorithm, "Fast and accurate computation of
binomial probabilities", 2000), and R's ``sum()`` as ``rsum`` does it: one long double accumulator over the
qualifying outcomes in index order, rounded to double at the end.

Written independently of ``gcre_values_table`` (geneticscre_amd/csrc/gcre_frontend.hip): scalar Python loops over
``math.log`` / ``math.log1p`` / ``math.exp`` (the C library's functions, as R calls them), ``numpy.longdouble`` for
the sum.  Only for small tables (a 70 x 60 table takes a second).

Parity status: **unpinned** -- R is not in this image and the reference ships no table; what this module pins is
that the native builder implements the published algorithm (tests/test_host_logic.py compares the two bit for bit).

Cells that hang on the last bit of ``dhyper``.  R's ``prob_dist <= x`` (Utils.R:153) is an exact comparison of
doubles.  Outcomes x and i - x' whose probabilities are equal in exact arithmetic (nCases = nControls makes every
diagonal symmetric; otherwise coincidences like C(5,2) C(9,5) = C(5,3) C(9,4)) are computed through different
``bd0`` / ``stirlerr`` operands and may differ in the last place; the smaller of the pair then loses the other's
mass in its two-sided p-value.  ``tie_sensitive_cells`` lists them for a table: the cells whose value changes when
probabilities within 1e-12 (relative) count as equal.  For the 70 x 60 table that is the list in
``TIE_SENSITIVE_70_60`` below (each moves by the mass of one outcome, e.g. ln 2 where the tied pair is the whole
tail).
"""
from __future__ import annotations

import math

import numpy as np

_S0 = 0.083333333333333333333          # 1/12
_S1 = 0.00277777777777777777778        # 1/360
_S2 = 0.00079365079365079365079365     # 1/1260
_S3 = 0.000595238095238095238095238    # 1/1680
_S4 = 0.0008417508417508417508417508   # 1/1188
_LN_2PI = 1.837877066409345483560659472811
_DBL_MIN = 2.2250738585072014e-308

# stirlerr.c, sferr_halves at the integers 0 .. 15
_SFERR = (
    0.0,
    0.0810614667953272582196702, 0.0413406959554092940938221, 0.02767792568499833914878929,
    0.02079067210376509311152277, 0.01664469118982119216319487, 0.01387612882307074799874573,
    0.01189670994589177009505572, 0.010411265261972096497478567, 0.009255462182712732917728637,
    0.008330563433362871256469318, 0.007573675487951840794972024, 0.006942840107209529865664152,
    0.006408994188004207068439631, 0.005951370112758847735624416, 0.005554733551962801371038690,
)


def stirlerr(n: float) -> float:
    """stirlerr.c for integer n: log(n!) - log(sqrt(2 pi n) (n/e)^n)."""
    if n <= 15.0:
        return _SFERR[int(n)]
    nn = n * n
    if n > 500:
        return (_S0 - _S1 / nn) / n
    if n > 80:
        return (_S0 - (_S1 - _S2 / nn) / nn) / n
    if n > 35:
        return (_S0 - (_S1 - (_S2 - _S3 / nn) / nn) / nn) / n
    return (_S0 - (_S1 - (_S2 - (_S3 - _S4 / nn) / nn) / nn) / nn) / n


def bd0(x: float, np_: float) -> float:
    """bd0.c: x log(x/np) + np - x, through its Taylor series where |x - np| < 0.1 (x + np)."""
    if abs(x - np_) < 0.1 * (x + np_):
        v = (x - np_) / (x + np_)
        s = (x - np_) * v
        if abs(s) < _DBL_MIN:
            return s
        ej = 2 * x * v
        v = v * v
        for j in range(1, 1000):
            ej *= v
            s1 = s + ej / ((j << 1) + 1)
            if s1 == s:
                return s1
            s = s1
    return x * math.log(x / np_) + np_ - x


def dbinom_raw(x: float, n: float, p: float, q: float) -> float:
    """dbinom.c dbinom_raw, give_log = FALSE."""
    if p == 0:
        return 1.0 if x == 0 else 0.0
    if q == 0:
        return 1.0 if x == n else 0.0
    if x == 0:
        if n == 0:
            return 1.0
        lc = (-bd0(n, n * q) - n * p) if p < 0.1 else n * math.log(q)
        return math.exp(lc)
    if x == n:
        lc = (-bd0(n, n * p) - n * q) if q < 0.1 else n * math.log(p)
        return math.exp(lc)
    if x < 0 or x > n:
        return 0.0
    lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q)
    lf = _LN_2PI + math.log(x) + math.log1p(-x / n)
    return math.exp(lc - 0.5 * lf)


def dhyper(x: int, r: int, b: int, n: int) -> float:
    """dhyper.c: P(x white among n drawn from r white + b black)."""
    x, r, b, n = float(x), float(r), float(b), float(n)
    if x < 0:
        return 0.0
    if n < x or r < x or n - x > b:
        return 0.0
    if n == 0:
        return 1.0 if x == 0 else 0.0
    p = n / (r + b)
    q = (r + b - n) / (r + b)
    p1 = dbinom_raw(x, r, p, q)
    p2 = dbinom_raw(n - x, b, p, q)
    p3 = dbinom_raw(n, r + b, p, q)
    return p1 * p2 / p3


def values_table(n_cases: int, n_ctrls: int, tie_slack: float = 0.0) -> np.ndarray:
    """getValuesTable (R/Utils.R:137-159).  ``tie_slack`` > 0 only serves ``tie_sensitive_cells``."""
    n = n_cases + n_ctrls
    table = np.full((n_cases + 1, n_ctrls + 1), np.nan)
    for i in range(n + 1):
        lo, hi = max(0, i - n_ctrls), min(i, n_cases)                     # Utils.R:149-151
        prob = [dhyper(x, n_cases, n_ctrls, i) for x in range(lo, hi + 1)]   # :144
        for k, x in enumerate(range(lo, hi + 1)):
            acc = np.longdouble(0.0)
            bound = prob[k] * (1.0 + tie_slack)
            for pj in prob:                                               # sum(prob_dist[prob_dist <= x]), :153
                if pj <= bound:
                    acc = acc + np.longdouble(pj)
            two_sided = float(acc)
            table[x, i - x] = -math.log(two_sided) if two_sided > 0.0 else math.inf   # :154
    finite = np.isfinite(table)
    table[~finite] = table[finite].max() + 1.0                           # :156
    return table


# tie_sensitive_cells(70, 60): the cells of the 70 x 60 table (4,331 cells) whose value hangs on the last place of dhyper
TIE_SENSITIVE_70_60 = [(8, 57), (13, 52), (25, 40), (28, 37), (36, 29), (41, 24), (52, 13)]


def tie_sensitive_cells(n_cases: int, n_ctrls: int):
    """Cells (x, i - x) whose value depends on how outcomes that tie in exact arithmetic compare in their last place."""
    a, b = values_table(n_cases, n_ctrls), values_table(n_cases, n_ctrls, tie_slack=1e-12)
    return [(int(r), int(c)) for r, c in zip(*np.nonzero(a != b))]
