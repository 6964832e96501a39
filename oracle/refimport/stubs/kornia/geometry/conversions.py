"""Stand-in for `from kornia.geometry.conversions import angle_axis_to_rotation_matrix`
(editing/render_geometry_editing.py:8,56) -- TEST INFRASTRUCTURE ONLY.

kornia is a third-party dependency of the reference, pinned at 0.6.3 (environment.yml:42) and not installed here.  This
restates its published algorithm (Rodrigues' formula after ceres/rotation.h): for theta^2 = |v|^2 > 1e-6 the axis is
v / (theta + 1e-6) and R = cos I + (1 - cos) w w^T + sin [w]x; otherwise the first-order form I + [v]x.  float64 is kept
out on purpose: kornia computes in the input's dtype."""
import torch


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    if angle_axis.dim() != 2 or angle_axis.shape[-1] != 3:
        raise ValueError(f"Input size must be a (*, 3) tensor. Got {tuple(angle_axis.shape)}")
    n = angle_axis.shape[0]
    out = torch.eye(3, dtype=angle_axis.dtype, device=angle_axis.device).repeat(n, 1, 1)
    th2 = torch.einsum("ni,ni->n", angle_axis, angle_axis)
    for i in range(n) if n <= 64 else ():   # small inputs: the scalar statement, the readable form of the algorithm
        v = angle_axis[i]
        if th2[i] > 1e-6:
            th = torch.sqrt(th2[i])
            x, y, z = (v / (th + 1e-6)).unbind()
            c, s = torch.cos(th), torch.sin(th)
            out[i] = torch.stack([torch.stack([c + x * x * (1 - c), x * y * (1 - c) - z * s, y * s + x * z * (1 - c)]),
                                  torch.stack([z * s + x * y * (1 - c), c + y * y * (1 - c), -x * s + y * z * (1 - c)]),
                                  torch.stack([-y * s + x * z * (1 - c), x * s + y * z * (1 - c), c + z * z * (1 - c)])])
        else:
            x, y, z = v.unbind()
            one = torch.ones_like(x)
            out[i] = torch.stack([torch.stack([one, -z, y]), torch.stack([z, one, -x]), torch.stack([-y, x, one])])
    if n > 64:   # the same, vectorised
        th = torch.sqrt(th2)
        w = angle_axis / (th + 1e-6)[:, None]
        x, y, z = w.unbind(-1)
        c, s = torch.cos(th), torch.sin(th)
        k = 1 - c
        big = torch.stack([c + x * x * k, x * y * k - z * s, y * s + x * z * k,
                           z * s + x * y * k, c + y * y * k, -x * s + y * z * k,
                           -y * s + x * z * k, x * s + y * z * k, c + z * z * k], -1).view(n, 3, 3)
        vx, vy, vz = angle_axis.unbind(-1)
        one = torch.ones_like(vx)
        small = torch.stack([one, -vz, vy, vz, one, -vx, -vy, vx, one], -1).view(n, 3, 3)
        out = torch.where((th2 > 1e-6)[:, None, None], big, small)
    return out
