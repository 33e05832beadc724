"""The two functions of DenseFusion/lib/transformations.py that sit on the hot path:
`quaternion_matrix` (reference :1254-1278) and `quaternion_from_matrix` (:1281-1363).  Host float64 helpers for callers
that still want numpy results; the device pipeline uses ape_pose_compose_f64 instead.  The rest of that 1900-line generic
maths library is out of scope."""
import math

import numpy

_EPS = numpy.finfo(float).eps * 4.0


def quaternion_matrix(quaternion):
    """4x4 homogeneous rotation from (w, x, y, z); identity for a near-zero quaternion."""
    q = numpy.array(quaternion, dtype=numpy.float64, copy=True)
    nq = numpy.dot(q, q)
    if nq < _EPS:
        return numpy.identity(4)
    q *= math.sqrt(2.0 / nq)
    o = numpy.outer(q, q)
    m = numpy.identity(4)
    m[0, 0] = 1.0 - o[2, 2] - o[3, 3]
    m[0, 1] = o[1, 2] - o[3, 0]
    m[0, 2] = o[1, 3] + o[2, 0]
    m[1, 0] = o[1, 2] + o[3, 0]
    m[1, 1] = 1.0 - o[1, 1] - o[3, 3]
    m[1, 2] = o[2, 3] - o[1, 0]
    m[2, 0] = o[1, 3] - o[2, 0]
    m[2, 1] = o[2, 3] + o[1, 0]
    m[2, 2] = 1.0 - o[1, 1] - o[2, 2]
    return m


def quaternion_from_matrix(matrix, isprecise=False):
    """(w, x, y, z) with w >= 0.  isprecise=True: closed form on a precise rotation matrix (the branch the live path uses,
    DenseFusion/tools/utils.py:36); otherwise the largest-eigenvector form."""
    M = numpy.asarray(matrix, dtype=numpy.float64)[:4, :4]
    if isprecise:
        q = numpy.empty((4,))
        t = numpy.trace(M)
        if t > M[3, 3]:
            q[:] = (t, M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1])
        else:
            i, j, k = 0, 1, 2
            if M[1, 1] > M[0, 0]:
                i, j, k = 1, 2, 0
            if M[2, 2] > M[i, i]:
                i, j, k = 2, 0, 1
            t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
            v = numpy.empty((4,))
            v[i] = t
            v[j] = M[i, j] + M[j, i]
            v[k] = M[k, i] + M[i, k]
            v[3] = M[k, j] - M[j, k]
            q = v[[3, 0, 1, 2]]
        q = q * (0.5 / math.sqrt(t * M[3, 3]))
    else:
        K = numpy.array([[M[0, 0] - M[1, 1] - M[2, 2], 0.0, 0.0, 0.0],
                         [M[0, 1] + M[1, 0], M[1, 1] - M[0, 0] - M[2, 2], 0.0, 0.0],
                         [M[0, 2] + M[2, 0], M[1, 2] + M[2, 1], M[2, 2] - M[0, 0] - M[1, 1], 0.0],
                         [M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1], M[0, 0] + M[1, 1] + M[2, 2]]]) / 3.0
        w, V = numpy.linalg.eigh(K)
        q = V[[3, 0, 1, 2], numpy.argmax(w)]
    if q[0] < 0.0:
        q = -q
    return q
