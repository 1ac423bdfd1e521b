"""The exclusion test that lets k_ff_tiles' leaf code postpone the division of Moeller-Trumbore (geom_kernels.hip,
leaf_blocked_mask): for float32 numerator a and determinant d, with r = fl(1/d) and u = fl(a * r) as the definition computes
them,
    |a| > fl(|d| * (1 + 2^-21))                              =>  u is not in [0, 1]
    sign(a) != sign(d)  and  |a| > fl(|d| * 2^-100)          =>  u is not >= 0   (in particular not -0)
Checked here in numpy's float32 arithmetic (IEEE, correctly rounded -- what the device's correctly rounded division and
plain multiplication do) over random values of every magnitude, denormals, zeros, infinities and the boundary cases."""
import numpy as np

C_HI = np.float32(1.0000005)          # 1 + 2^-21
C_LO = np.float32(7.888609e-31)       # 2^-100


def excluded(a, d):
    a, d = a.astype(np.float32), d.astype(np.float32)
    ad = np.abs(d)
    with np.errstate(all="ignore"):
        big = np.abs(a) > ad * C_HI
        sign = (a.view(np.int32) ^ d.view(np.int32)) < 0
        neg = sign & (np.abs(a) > ad * C_LO)
    return big | neg


def u_of(a, d):
    with np.errstate(all="ignore"):
        r = np.float32(1.0) / d.astype(np.float32)
        return a.astype(np.float32) * r


def in_unit(u):
    with np.errstate(all="ignore"):
        return (u >= np.float32(0.0)) & (u <= np.float32(1.0))


def _check(a, d):
    ex = excluded(a, d)
    u = u_of(a, d)
    bad = ex & in_unit(u)
    assert not bad.any(), (a[bad][:5], d[bad][:5], u[bad][:5])
    return ex


def test_constants_are_the_powers_of_two_the_argument_uses():
    assert C_HI == np.float32(1.0) + np.float32(2.0 ** -21)
    assert C_LO == np.float32(2.0 ** -100)


def test_random_values_of_every_magnitude():
    rs = np.random.RandomState(1)
    n = 4_000_000
    for spread in (3, 30, 120):                                   # exponent ranges: near each other ... anywhere in float32
        ea, ed = rs.uniform(-spread, spread, n), rs.uniform(-spread, spread, n)
        a = (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * 2.0 ** ea).astype(np.float32)
        d = (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * 2.0 ** ed).astype(np.float32)
        ex = _check(a, d)
        assert ex.mean() > 0.4                                    # the test is not vacuous


def test_ratios_close_to_the_two_ends_of_the_interval():
    """a / d within a few ulps of 1 (where the first rule has to stay on the safe side) and tiny a of the other sign"""
    rs = np.random.RandomState(2)
    n = 2_000_000
    d = (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * 2.0 ** rs.uniform(-60, 60, n)).astype(np.float32)
    k = rs.randint(-12, 13, n)
    a = d.copy()
    for _ in range(12):                                           # step a by up to 12 ulps away from d, both ways
        up = k > 0
        a = np.where(up, np.nextafter(a, np.float32(np.inf) * np.sign(a)), np.where(k < 0, np.nextafter(a, np.float32(0)), a)).astype(np.float32)
        k = k - np.sign(k)
    _check(a, d)
    _check(-a, d)
    tiny = (d * np.float32(2.0 ** -100) * rs.uniform(0.25, 4, n).astype(np.float32)).astype(np.float32)
    _check(-tiny, d)
    _check(tiny, d)


def test_special_values():
    vals = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -1e-40, 1.1754944e-38, -1.1754944e-38, 1e-20, -1e-20, 1.0, -1.0, 3.0, -3.0,
                     1e20, -1e20, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan], np.float32)
    a, d = np.meshgrid(vals, vals)
    ex = _check(a.ravel(), d.ravel())
    # a = +-0 is never excluded by its sign (its u = -0 passes `u >= 0`)
    zero = a.ravel() == 0
    u = u_of(a.ravel(), d.ravel())
    assert not (ex & zero & in_unit(u)).any()


# ---- the third exclusion test: the sign of t's numerator ----------------------------------------
def t_excluded(a, d):
    """leaf_blocked_mask's rule for t = fl(a * fl(1/d)): numerator and determinant of opposite sign, or a zero numerator"""
    a, d = a.astype(np.float32), d.astype(np.float32)
    return ((a.view(np.int32) ^ d.view(np.int32)) < 0) | (a == np.float32(0.0))


def _check_t(a, d):
    ex = t_excluded(a, d)
    t = u_of(a, d)                       # the same two operations: fl(a * fl(1/d))
    with np.errstate(all="ignore"):
        bad = ex & (t > np.float32(0.0))
    assert not bad.any(), (a[bad][:5], d[bad][:5], t[bad][:5])
    return ex


def test_sign_rule_for_t_never_excludes_a_positive_t():
    """A hit needs t > 0 strictly.  With r = fl(1/d) of d's sign (also when it overflows, or d is a denormal or zero), a product
    of opposite signs is negative or -0 and a zero numerator gives 0 or NaN: never > 0.  Random values of every magnitude,
    the neighbourhood of zero, and the special values."""
    rs = np.random.RandomState(3)
    n = 4_000_000
    for spread in (3, 60, 149):
        ea, ed = rs.uniform(-spread, min(spread, 127), n), rs.uniform(-spread, min(spread, 127), n)
        a = (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * 2.0 ** ea).astype(np.float32)
        d = (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * 2.0 ** ed).astype(np.float32)
        ex = _check_t(a, d)
        assert 0.4 < ex.mean() < 0.6
    vals = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -1e-40, 1.1754944e-38, -1.1754944e-38, 1e-20, -1e-20, 1.0, -1.0, 3.0, -3.0,
                     1e20, -1e20, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan], np.float32)
    a, d = np.meshgrid(vals, vals)
    ex = _check_t(a.ravel(), d.ravel())
    # and it is not vacuous where it matters: same sign, both finite and not tiny -> kept
    keep = ~t_excluded(np.array([1e-3, -2.0], np.float32), np.array([5.0, -1e-6], np.float32))
    assert keep.all()
