"""ORACLE — test infrastructure only (see oracle/__init__.py).  Parity UNPINNED: the reference holds no vectors for its
rasteriser, SH colours included; what pins this file is (a) the published real-SH basis it restates and (b) its own backward
checked against autograd through its forward in float64 (tests/test_oracle_sh.py).

numpy restatement of the spherical-harmonics colour of the reference's rasteriser:

* ``sh_to_rgb``           computeColorFromSH, cuda_rasterizer/forward.cu:20-71 (called from preprocessCUDA, :240-247, when
                          ``colors_precomp`` is NULL): unit direction camera centre -> mean, the degree-0..3 polynomial sum
                          in the reference's term order, + 0.5, clamp at 0 with the clamp mask kept.
* ``sh_to_rgb_backward``  computeColorFromSH, cuda_rasterizer/backward.cu:20-140: gradient w.r.t. the coefficients, and
                          w.r.t. the mean through the normalised direction (dnormvdv, auxiliary.h:107-117); called at backward.cu:390-391.

Arithmetic in float32 (``dtype=np.float64`` for the derivative check), vectorised over the Gaussians.
"""
import numpy as np

# auxiliary.h:22-39
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def _direction(means, campos, dtype):
    v = means.astype(dtype) - np.asarray(campos, dtype).reshape(1, 3)
    length = np.sqrt((v * v).sum(1, dtype=dtype)).astype(dtype)
    return v, length, (v / length[:, None]).astype(dtype)


def sh_to_rgb(means, campos, shs, deg, dtype=np.float32):
    """means (P,3), campos (3,), shs (P,M,3), deg 0..3 -> (colors (P,3), clamped (P,3) bool).  forward.cu:20-71."""
    f = dtype
    _, _, d = _direction(means, campos, f)
    x, y, z = (d[:, k:k + 1] for k in range(3))
    sh = shs.astype(f)
    res = f(C0) * sh[:, 0]
    if deg > 0:
        res = res - f(C1) * y * sh[:, 1] + f(C1) * z * sh[:, 2] - f(C1) * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + f(C2[0]) * xy * sh[:, 4] + f(C2[1]) * yz * sh[:, 5] + f(C2[2]) * (f(2) * zz - xx - yy) * sh[:, 6]
                   + f(C2[3]) * xz * sh[:, 7] + f(C2[4]) * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + f(C3[0]) * y * (f(3) * xx - yy) * sh[:, 9] + f(C3[1]) * xy * z * sh[:, 10]
                       + f(C3[2]) * y * (f(4) * zz - xx - yy) * sh[:, 11]
                       + f(C3[3]) * z * (f(2) * zz - f(3) * xx - f(3) * yy) * sh[:, 12]
                       + f(C3[4]) * x * (f(4) * zz - xx - yy) * sh[:, 13] + f(C3[5]) * z * (xx - yy) * sh[:, 14]
                       + f(C3[6]) * x * (xx - f(3) * yy) * sh[:, 15])
    res = (res + f(0.5)).astype(f)
    return np.maximum(res, f(0)), res < 0


def sh_to_rgb_backward(means, campos, shs, deg, clamped, dL_dcolor, dtype=np.float32):
    """-> (dL_dmeans (P,3): the part through the view direction only, dL_dshs (P,M,3)).  backward.cu:20-140."""
    f = dtype
    v, length, d = _direction(means, campos, f)
    x, y, z = (d[:, k:k + 1] for k in range(3))
    sh = shs.astype(f)
    g = np.where(clamped, f(0), dL_dcolor.astype(f)).astype(f)                  # backward.cu:33-36
    d_sh = np.zeros_like(sh)
    zero = np.zeros_like(g)
    dx, dy, dz = zero.copy(), zero.copy(), zero.copy()                          # d colour / d dir, per channel
    d_sh[:, 0] = f(C0) * g
    if deg > 0:
        d_sh[:, 1], d_sh[:, 2], d_sh[:, 3] = -f(C1) * y * g, f(C1) * z * g, -f(C1) * x * g
        dx, dy, dz = -f(C1) * sh[:, 3], -f(C1) * sh[:, 1], f(C1) * sh[:, 2]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            d_sh[:, 4], d_sh[:, 5] = f(C2[0]) * xy * g, f(C2[1]) * yz * g
            d_sh[:, 6] = f(C2[2]) * (f(2) * zz - xx - yy) * g
            d_sh[:, 7], d_sh[:, 8] = f(C2[3]) * xz * g, f(C2[4]) * (xx - yy) * g
            dx = dx + f(C2[0]) * y * sh[:, 4] - f(C2[2]) * f(2) * x * sh[:, 6] + f(C2[3]) * z * sh[:, 7] \
                + f(C2[4]) * f(2) * x * sh[:, 8]
            dy = dy + f(C2[0]) * x * sh[:, 4] + f(C2[1]) * z * sh[:, 5] - f(C2[2]) * f(2) * y * sh[:, 6] \
                - f(C2[4]) * f(2) * y * sh[:, 8]
            dz = dz + f(C2[1]) * y * sh[:, 5] + f(C2[2]) * f(4) * z * sh[:, 6] + f(C2[3]) * x * sh[:, 7]
            if deg > 2:
                d_sh[:, 9] = f(C3[0]) * y * (f(3) * xx - yy) * g
                d_sh[:, 10] = f(C3[1]) * xy * z * g
                d_sh[:, 11] = f(C3[2]) * y * (f(4) * zz - xx - yy) * g
                d_sh[:, 12] = f(C3[3]) * z * (f(2) * zz - f(3) * xx - f(3) * yy) * g
                d_sh[:, 13] = f(C3[4]) * x * (f(4) * zz - xx - yy) * g
                d_sh[:, 14] = f(C3[5]) * z * (xx - yy) * g
                d_sh[:, 15] = f(C3[6]) * x * (xx - f(3) * yy) * g
                dx = dx + (f(C3[0]) * sh[:, 9] * f(6) * xy + f(C3[1]) * sh[:, 10] * yz - f(C3[2]) * sh[:, 11] * f(2) * xy
                           - f(C3[3]) * sh[:, 12] * f(6) * xz + f(C3[4]) * sh[:, 13] * (f(4) * zz - f(3) * xx - yy)
                           + f(C3[5]) * sh[:, 14] * f(2) * xz + f(C3[6]) * sh[:, 15] * f(3) * (xx - yy))
                dy = dy + (f(C3[0]) * sh[:, 9] * f(3) * (xx - yy) + f(C3[1]) * sh[:, 10] * xz
                           + f(C3[2]) * sh[:, 11] * (f(4) * zz - xx - f(3) * yy) - f(C3[3]) * sh[:, 12] * f(6) * yz
                           - f(C3[4]) * sh[:, 13] * f(2) * xy - f(C3[5]) * sh[:, 14] * f(2) * yz
                           - f(C3[6]) * sh[:, 15] * f(6) * xy)
                dz = dz + (f(C3[1]) * sh[:, 10] * xy + f(C3[2]) * sh[:, 11] * f(8) * yz
                           + f(C3[3]) * sh[:, 12] * f(3) * (f(2) * zz - xx - yy) + f(C3[4]) * sh[:, 13] * f(8) * xz
                           + f(C3[5]) * sh[:, 14] * (xx - yy))
    ddir = np.stack(((dx * g).sum(1), (dy * g).sum(1), (dz * g).sum(1)), 1).astype(f)       # backward.cu:129
    # dnormvdv (auxiliary.h:107-117): d (v / |v|) applied to ddir
    sum2 = (v * v).sum(1, keepdims=True)
    inv32 = f(1) / np.sqrt(sum2 * sum2 * sum2)
    d_means = ((sum2 * ddir - v * (v * ddir).sum(1, keepdims=True)) * inv32).astype(f)
    return d_means, d_sh
