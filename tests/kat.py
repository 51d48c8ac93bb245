"""Hand-derived known-answer cases for OpenCV BFMatcher semantics (SURVEY.md Appendix A;
8(c) list (1)-(6)).  Shared by the oracle tests (CPU) and the HIP parity tests (GPU).
Each case: Q, T (uint8 [n, dim]) and the expected results derived on paper."""
import numpy as np

INF = np.float32(np.inf)


def _col(vals, dim=4):
    a = np.zeros((len(vals), dim), dtype=np.uint8)
    a[:, 0] = vals
    return a


def xcheck_cases():
    cases = []
    # (1) cross-check is reverse-NN + scatter-min, not mutual NN.
    # q0=10, q1=13; t0=12, t1=5.  t0: d(q0)=2, d(q1)=1 -> elects q1.  t1: d(q0)=5, d(q1)=8
    # -> elects q0.  q0's own nearest train row is t0, which elected q1; mutual-NN would
    # leave q0 unmatched, OpenCV matches q0 to t1 at distance 5.
    cases.append(("not_mutual", _col([10, 13]), _col([12, 5]), [1, 0], [5.0, 1.0]))
    # (2a) column tie: t0=20 is at distance 3 from both q0=17 and q1=23 -> elects the
    # lower query index q0; q1 gets nothing.
    cases.append(("tie_lowest_query", _col([17, 23]), _col([20]), [0, -1], [3.0, np.inf]))
    # (2b) row tie: t0=30 and t1=36 both elect q0=33 at distance 3 -> q0 keeps the lower
    # train index t0.  q1=100 is elected by nobody.
    cases.append(("tie_lowest_train", _col([33, 100]), _col([30, 36]), [0, -1], [3.0, np.inf]))
    # (2c) scatter-min keeps the CLOSEST electing train row, not the first: t0=40 (d=5),
    # t1=44 (d=1) both elect q0=45.
    cases.append(("closest_elector", _col([45]), _col([40, 44]), [1], [1.0]))
    # (5) extreme distance: all-0 vs all-255 over 128 dims: d2 = 128*255^2 = 8 323 200.
    q = np.zeros((1, 128), dtype=np.uint8)
    t = np.full((1, 128), 255, dtype=np.uint8)
    cases.append(("extreme", q, t, [0], [float(np.sqrt(np.float32(8323200.0)))]))
    # exact duplicate rows across banks: distance 0, lowest indices win everywhere.
    # t0 == t1 == q0 == q1 = 7: t0 elects q0 (d 0), t1 elects q0 (d 0); q0 keeps t0.
    cases.append(("all_duplicates", _col([7, 7]), _col([7, 7]), [0, -1], [0.0, np.inf]))
    # (6) empty train set: every inner list empty.
    cases.append(("empty_train", _col([1, 2, 3]), np.zeros((0, 4), dtype=np.uint8), [-1, -1, -1],
                  [np.inf, np.inf, np.inf]))
    return cases


def knn2_cases():
    cases = []
    # ascending distance, lower train index first on ties: q0=50; t = 53, 47, 50, 50
    # distances 3,3,0,0 -> [t2, t3]
    cases.append(("ties", _col([50]), _col([53, 47, 50, 50]), [[2, 3]], [[0.0, 0.0]]))
    # tie for second place: q0=10; t = 12, 8, 10 -> first t2 (0), second t0 (2; lower idx than t1)
    cases.append(("tie_second", _col([10]), _col([12, 8, 10]), [[2, 0]], [[0.0, 2.0]]))
    # (4) fewer than k train rows: inner list of length 1 -> idx -1 / dist inf in slot 2
    cases.append(("nt1", _col([10, 20]), _col([14]), [[0, -1], [0, -1]], [[4.0, np.inf], [6.0, np.inf]]))
    # 3-4-5 triangle over two dims
    q = np.zeros((1, 4), dtype=np.uint8)
    t = np.zeros((2, 4), dtype=np.uint8)
    t[0, 0], t[0, 1] = 3, 4
    t[1, 0], t[1, 1] = 6, 8
    cases.append(("pythagoras", q, t, [[0, 1]], [[5.0, 10.0]]))
    return cases


def selfdist_case():
    # (3) duplicates: D = [a, a, b]; rows 0 and 1 see each other at distance 0 as r[1]
    # (row 0: r[0] = itself idx0, r[1] = idx1; row 1: r[0] = idx0, r[1] = itself idx1),
    # row 2 (b = a + 9 in one dim): r[0] = itself (0), r[1] = idx0 at distance 9.
    D = _col([100, 100, 109])
    return D, [0.0, 0.0, 9.0]


# ---- float32 square-root ties ------------------------------------------
# cv::batchDistance stores dist = sqrtf((float)d2) and runs the k-NN insertion / the cross-check compare
# on those float32 values.  From d2 = SQRT_TIE_MIN on, n and n + 1 can have the same float32 root, and
# then the LOWER INDEX wins although its d2 is larger.  First such pair: 4 197 200 / 4 197 201; no three
# integers up to 128 * 255^2 share a root (test_oracle_kat.test_sqrt_tie_table checks both facts).
SQRT_TIE_MIN = 4197200


def row_with_sumsq(target, dim=128):
    """uint8 row whose sum of squares is exactly ``target`` (greedy: largest square that fits, per slot)."""
    row = np.zeros(dim, dtype=np.uint8)
    rest = int(target)
    for k in range(dim):
        v = min(255, int(np.floor(np.sqrt(rest))))
        row[k] = v
        rest -= v * v
        if rest == 0:
            return row
    raise ValueError("cannot reach %d with %d uint8 values" % (target, dim))


def _f32root(n):
    return float(np.sqrt(np.float32(n)))


def sqrt_tie_knn2_cases():
    """(name, Q, T, expected idx, expected dist): query = the zero row, so d2(q, t) = |t|^2."""
    n = SQRT_TIE_MIN
    z = np.zeros((1, 128), dtype=np.uint8)
    lo, hi = row_with_sumsq(n), row_with_sumsq(n + 1)
    r = _f32root(n)
    assert _f32root(n + 1) == r
    cases = []
    # the larger d2 comes first in the bank: same float32 distance => it stays in front
    cases.append(("sqrt_tie_larger_first", z, np.stack([hi, lo]), [[0, 1]], [[r, r]]))
    cases.append(("sqrt_tie_smaller_first", z, np.stack([lo, hi]), [[0, 1]], [[r, r]]))
    # three rows in one tie group: index order throughout (an integer order gives [2, 0])
    cases.append(("sqrt_tie_three", z, np.stack([hi, hi, lo]), [[0, 1]], [[r, r]]))
    # tie for SECOND place only: best is far below the tie range
    near = row_with_sumsq(100)
    cases.append(("sqrt_tie_second_place", z, np.stack([hi, near, lo]), [[1, 0]], [[10.0, r]]))
    # control: 4 197 201 / 4 197 202 have different roots -> the smaller d2 wins whatever its index
    a, b = row_with_sumsq(n + 2), row_with_sumsq(n + 1)
    assert _f32root(n + 2) != _f32root(n + 1)
    cases.append(("sqrt_no_tie_control", z, np.stack([a, b]), [[1, 0]], [[_f32root(n + 1), _f32root(n + 2)]]))
    return cases


def sqrt_tie_xcheck_cases():
    n = SQRT_TIE_MIN
    z = np.zeros((1, 128), dtype=np.uint8)
    lo, hi = row_with_sumsq(n), row_with_sumsq(n + 1)
    r = _f32root(n)
    cases = []
    # scatter-min side: both train rows elect q0 (the only query); tdist equal as float32 => lowest t
    cases.append(("sqrt_tie_scatter_larger_first", z, np.stack([hi, lo]), [0], [r]))
    cases.append(("sqrt_tie_scatter_smaller_first", z, np.stack([lo, hi]), [0], [r]))
    # election side: the one train row (zero row) sees q0 at n + 1 and q1 at n: same float32 distance
    # => it elects the LOWER query index q0; q1 stays unmatched
    cases.append(("sqrt_tie_election_larger_first", np.stack([hi, lo]), z, [0, -1], [r, np.inf]))
    cases.append(("sqrt_tie_election_smaller_first", np.stack([lo, hi]), z, [0, -1], [r, np.inf]))
    # three queries in one tie group
    cases.append(("sqrt_tie_election_three", np.stack([hi, hi, lo]), z, [0, -1, -1], [r, np.inf, np.inf]))
    # control without a tie: the smaller d2 is elected
    a, b = row_with_sumsq(n + 2), row_with_sumsq(n + 1)
    cases.append(("sqrt_no_tie_election_control", np.stack([a, b]), z, [-1, 0], [np.inf, _f32root(n + 1)]))
    return cases


def _tie_offset(base, span=6):
    """c in 0..255 such that as many of base + c^2 + k, k < span, as possible share their float32 root
    with their successor."""
    def score(c):
        n = base + c * c
        return sum(_f32root(n + k) == _f32root(n + k + 1) for k in range(span)) if n >= SQRT_TIE_MIN else -1
    return max(range(256), key=score)


def far_banks(nq, nt, rng, small_dims=10, small_max=1, base_dims=100):
    """uint8 banks in which EVERY query/train distance lies in the tie range and many candidates of a
    row sit on both sides of a tie: dimensions [0, base_dims) are 0 in every query row and 255 in every
    train row, dimension base_dims is 0 / c with c chosen so that base = base_dims * 255^2 + c^2 shares
    its float32 root with base + 1, and ``small_dims`` further dimensions hold values 0..small_max:
    d2 = base + (a handful of small integers), the same few values over and over."""
    base = base_dims * 255 * 255
    c = _tie_offset(base)
    Q = np.zeros((nq, 128), dtype=np.uint8)
    T = np.zeros((nt, 128), dtype=np.uint8)
    T[:, :base_dims] = 255
    T[:, base_dims] = c
    lo = base_dims + 1
    Q[:, lo:lo + small_dims] = rng.integers(0, small_max + 1, (nq, small_dims), dtype=np.uint8)
    T[:, lo:lo + small_dims] = rng.integers(0, small_max + 1, (nt, small_dims), dtype=np.uint8)
    return Q, T


def far_image_pair(query, target, seed=0, base_dims=100, small_dims=10):
    """Rewrite the descriptors of a synth.image_pair so that every query/target distance lies in the
    float32 tie range (far_banks' layout; at d2 ~ 6.5e6 one integer in four shares its root with its
    successor) while the query rows stay distinct from each other (17 dimensions of their own holding
    exactly 8 ones each: the same contribution to every query/target distance, small non-zero self
    distances): ratios are in the hundreds, thresholds of a test are
    chosen accordingly.  Positions are untouched."""
    rng = np.random.default_rng(seed)
    base = base_dims * 255 * 255
    c = _tie_offset(base + 8)         # (every query row adds 8 ones of its own)
    lo = base_dims + 1

    def q_rows(d):
        out = np.zeros_like(d)
        out[:, lo:lo + small_dims] = d[:, :small_dims] & 1
        ones = np.argsort(rng.random((len(d), 17)), axis=1)[:, :8]
        np.put_along_axis(out[:, 111:128], ones, 1, axis=1)
        return out

    def t_rows(d):
        out = np.zeros_like(d)
        out[:, :base_dims] = 255
        out[:, base_dims] = c
        out[:, lo:lo + small_dims] = d[:, :small_dims] & 1
        return out

    q, t = dict(query), dict(target)
    q["descriptors"], q["thumb_descriptors"] = q_rows(query["descriptors"]), q_rows(query["thumb_descriptors"])
    t["descriptors"], t["thumb_descriptors"] = t_rows(target["descriptors"]), t_rows(target["thumb_descriptors"])
    return q, t
