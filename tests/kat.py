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
