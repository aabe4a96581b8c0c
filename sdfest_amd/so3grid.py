"""Deterministic low-dispersion grid on SO(3): host-side mirror of ``sdfest/initialization/so3grid.py``
(Yershova et al. 2010, base grid of the Hopf fibration: an equidistant grid on S1 x the HEALPix grid on S2).

The reference gets the S2 part from healpy (``hp.ang2pix`` / ``hp.pix2ang`` with ``nest=True``, so3grid.py:43,
:173; healpy==1.15.2, requirements.txt:39), a dependency that is not in this image and not under
/root/reference.  The two functions are restated here from the published algorithm (Gorski et al. 2005,
"HEALPix", ApJ 622:759, section 4.1 and the NESTED indexing of its appendix): 12 base faces, N_side^2
pixels per face indexed by bit-interleaving the in-face coordinates.  Pinned by the defining properties
(tests/test_so3grid_cpu.py): the 12 known base-pixel centres, pixel -> angle -> pixel round trips, equal
areas, and the nesting of a pixel's 4 children; the Hopf conversions by the reference suite's own
known answers (tests/initilization/test_so3grid.py).  Plain numpy, vectorised over arrays of orientations.
"""
from typing import Tuple

import numpy as np

_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4])
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7])


def _spread_bits(v: np.ndarray) -> np.ndarray:
    """bit i of v -> bit 2i (v < 2^16)."""
    v = v.astype(np.int64)
    v = (v | (v << 8)) & 0x00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def _compress_bits(v: np.ndarray) -> np.ndarray:
    """inverse of _spread_bits on the even bits."""
    v = v.astype(np.int64) & 0x55555555
    v = (v | (v >> 1)) & 0x33333333
    v = (v | (v >> 2)) & 0x0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF
    return v


def ang2pix_nest(nside: int, theta, phi) -> np.ndarray:
    """NESTED index of the HEALPix pixel containing the direction (theta = colatitude in [0, pi],
    phi = longitude); nside a power of two <= 2^15."""
    theta = np.asarray(theta, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    z = np.cos(theta)
    za = np.abs(z)
    tt = np.mod(phi, 2 * np.pi) / (0.5 * np.pi)          # [0, 4)
    tt = np.where(tt >= 4.0, 0.0, tt)
    order = int(round(np.log2(nside)))
    # equatorial belt
    t1 = nside * (0.5 + tt)
    t2 = nside * z * 0.75
    jp = np.floor(t1 - t2).astype(np.int64)             # ascending edge line
    jm = np.floor(t1 + t2).astype(np.int64)             # descending edge line
    ifp, ifm = jp >> order, jm >> order
    face_eq = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))
    ix_eq = jm & (nside - 1)
    iy_eq = nside - (jp & (nside - 1)) - 1
    # polar caps
    ntt = np.minimum(3, tt.astype(np.int64))
    tp = tt - ntt
    tmp = nside * np.sqrt(3.0 * (1.0 - za))
    jpp = np.minimum(np.floor(tp * tmp).astype(np.int64), nside - 1)
    jmp = np.minimum(np.floor((1.0 - tp) * tmp).astype(np.int64), nside - 1)
    north = z >= 0
    face_po = np.where(north, ntt, ntt + 8)
    ix_po = np.where(north, nside - jmp - 1, jpp)
    iy_po = np.where(north, nside - jpp - 1, jmp)
    eq = za <= 2.0 / 3.0
    face = np.where(eq, face_eq, face_po)
    ix = np.where(eq, ix_eq, ix_po)
    iy = np.where(eq, iy_eq, iy_po)
    return face * nside * nside + _spread_bits(ix) + (_spread_bits(iy) << 1)


def pix2ang_nest(nside: int, pix) -> Tuple[np.ndarray, np.ndarray]:
    """(theta, phi) of the centre of NESTED pixel(s) `pix`."""
    pix = np.asarray(pix, dtype=np.int64)
    npface = nside * nside
    face = pix // npface
    p = pix % npface
    ix, iy = _compress_bits(p), _compress_bits(p >> 1)
    jr = _JRLL[face] * nside - ix - iy - 1               # ring number, 1 .. 4 nside - 1
    nl4 = 4 * nside
    fact2 = 4.0 / (12.0 * npface)
    fact1 = 2.0 * nside * fact2
    north, south = jr < nside, jr > 3 * nside
    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    z = np.where(north, 1.0 - nr * nr * fact2, np.where(south, nr * nr * fact2 - 1.0, (2 * nside - jr) * fact1))
    kshift = np.where(north | south, 0, (jr - nside) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, jp)
    jp = np.where(jp < 1, jp + nl4, jp)
    phi = (jp - (kshift + 1) * 0.5) * (0.5 * np.pi / nr)
    return np.arccos(z), phi


class SO3Grid:
    """Same methods and conventions as the reference class (so3grid.py:8-175); quaternions are
    (x, y, z, w); every method also takes arrays (index arrays / (N,4) quaternions)."""

    def __init__(self, resol: int):
        self._resol = resol
        points = 6 * 2 ** resol                                                    # :151-161
        self._s1 = np.linspace(0, 2 * np.pi, points, endpoint=False) + np.pi / points
        nside = 2 ** resol                                                         # :163-175
        self._nside = nside
        self._s2_theta, self._s2_phi = pix2ang_nest(nside, np.arange(12 * nside * nside))

    def num_cells(self) -> int:
        return len(self._s1) * len(self._s2_theta)

    def hopf_to_index(self, psi, theta, phi):
        """so3grid.py:31-45: S1 cell by floor division, S2 cell = the HEALPix pixel of (theta, phi)."""
        s1_index = np.floor_divide(psi, 2 * np.pi / len(self._s1)).astype(np.int64)
        s2_index = ang2pix_nest(self._nside, theta, phi)
        out = s1_index * len(self._s2_theta) + s2_index
        return int(out) if np.ndim(out) == 0 else out

    def index_to_hopf(self, index):
        s1_index, s2_index = np.divmod(index, len(self._s2_theta))
        return self._s1[s1_index], self._s2_theta[s2_index], self._s2_phi[s2_index]

    def quat_to_index(self, quaternion):
        return self.hopf_to_index(*SO3Grid._quat_to_hopf(quaternion))

    def index_to_quat(self, index):
        return SO3Grid._hopf_to_quat(*self.index_to_hopf(index))

    @staticmethod
    def _quat_to_hopf(quaternion):
        """so3grid.py:89-124; psi, theta, phi in [0, 2 pi), [0, pi], [0, 2 pi)."""
        q = np.asarray(quaternion)
        x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        psi = 2 * np.arctan2(x, w)
        theta = 2 * np.arctan2(np.sqrt(z ** 2 + y ** 2), np.sqrt(w ** 2 + x ** 2))
        phi = np.arctan2(z * w - x * y, y * w + x * z)
        two_pi = 2 * np.pi
        # the reference's `while` corrections (:111-123): at most one step for float64 arctan2
        psi = np.where(psi < 0, psi + two_pi, psi)
        psi = np.where(psi >= two_pi, psi - two_pi, psi)
        phi = np.where(phi < 0, phi + two_pi, phi)
        phi = np.where(phi >= two_pi, phi - two_pi, phi)
        return psi, theta, phi

    @staticmethod
    def _hopf_to_quat(psi, theta, phi):
        """so3grid.py:126-149 (equation (4) of Yershova 2010, then the x >= 0 half-sphere)."""
        psi, theta, phi = np.asarray(psi, float), np.asarray(theta, float), np.asarray(phi, float)
        q = np.stack([np.cos(theta / 2) * np.sin(psi / 2), np.sin(theta / 2) * np.cos(phi + psi / 2),
                      np.sin(theta / 2) * np.sin(phi + psi / 2), np.cos(theta / 2) * np.cos(psi / 2)], axis=-1)
        return np.where(q[..., :1] < 0, -q, q)
