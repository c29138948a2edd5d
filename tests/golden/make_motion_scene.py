"""Regenerates tests/golden/motion_scene_*.npz: the matched-point sets of the
reference's motion-estimator tests (ref:tests/test_motion_estimator.cpp:23-56).

The reference draws them from std::mt19937(42) through libstdc++'s
uniform_real_distribution<double>; both are restated here: numpy's legacy
MT19937 seeding equals std::mt19937's, and libstdc++'s generate_canonical for a
32-bit engine is (r1 + r2 * 2^32) / 2^64 evaluated in double.

The reference passes the three draws of a point as constructor arguments
(ref:tests/test_motion_estimator.cpp:39-40), whose evaluation order the language
leaves open: clang (the reference's configured compiler) evaluates them left to
right - x, y, z, the files motion_scene_<name>.npz -, g++ right to left - z, y, x,
the files motion_scene_<name>_zyx.npz (first point x = 879.69 instead of 896.54;
tests/test_oracle_cpu.py checks that against a g++ build of the same expression).
"""
import os

import numpy as np

FX, FY, CX, CY = 718.856, 718.856, 607.1928, 185.2157


def scene(R, t, n=100, order="xyz"):
    raw = np.random.RandomState(42)._bit_generator.random_raw(6 * n).astype(np.float64)
    it = iter(raw)

    def canon():
        r1, r2 = next(it), next(it)
        v = (r1 + r2 * 4294967296.0) / 18446744073709551616.0
        return np.nextafter(1.0, 0.0) if v >= 1.0 else v

    p1s, p2s = [], []
    for _ in range(n):
        if order == "xyz":
            x = canon() * (1100 - 100) + 100
            y = canon() * (300 - 50) + 50
            z = canon() * (50 - 5) + 5
        else:
            z = canon() * (50 - 5) + 5
            y = canon() * (300 - 50) + 50
            x = canon() * (1100 - 100) + 100
        P = np.array([x - CX, y - CY, z])
        P[0] *= P[2] / FX
        P[1] *= P[2] / FY
        p1 = np.array([FX * P[0] + 0 * P[1] + CX * P[2], 0 * P[0] + FY * P[1] + CY * P[2], P[2]])
        p1 /= p1[2]
        Q = R @ P + t
        p2 = np.array([FX * Q[0] + 0 * Q[1] + CX * Q[2], 0 * Q[0] + FY * Q[1] + CY * Q[2], Q[2]])
        p2 /= p2[2]
        p1s.append(p1[:2].astype(np.float32))
        p2s.append(p2[:2].astype(np.float32))
    return np.array(p1s), np.array(p2s)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # EstimatesForwardMotion
    R = np.eye(3)
    t = np.array([0.0, 0.0, 1.0])
    for order, tag in (("xyz", ""), ("zyx", "_zyx")):
        p1, p2 = scene(R, t, order=order)
        np.savez(os.path.join(here, f"motion_scene_forward{tag}.npz"), p1=p1, p2=p2, R=R, t=t)
    # EstimatesRotation: Ry(0.05), t = normalised (0.1, 0, 1)
    a = 0.05
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    t = np.array([0.1, 0.0, 1.0])
    t /= np.linalg.norm(t)
    for order, tag in (("xyz", ""), ("zyx", "_zyx")):
        p1, p2 = scene(R, t, order=order)
        np.savez(os.path.join(here, f"motion_scene_rotation{tag}.npz"), p1=p1, p2=p2, R=R, t=t)


if __name__ == "__main__":
    main()
