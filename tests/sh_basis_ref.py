"""Test helper: the real SH basis (degree <= 3) in PyTorch, written from the published constants of the 3DGS
SH evaluation (reference: cuda_rasterizer/auxiliary.h:20-39, forward.cu:20-71) -- the reconstruct() stand-in for
dist.allreduce_gradients_sh_factored on CPU, and the float64 cross-check of goi_raster_sh_grad_from_views."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(dirs: torch.Tensor, degree: int, M: int) -> torch.Tensor:
    """dirs [P,3] unit vectors -> [P,M] basis values (zeros above (degree+1)^2)."""
    x, y, z = dirs[:, 0], dirs[:, 1], dirs[:, 2]
    b = [torch.full_like(x, C0)]
    if degree > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if degree > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if degree > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    while len(b) < M:
        b.append(torch.zeros_like(x))
    return torch.stack(b[:M], dim=1)


def sh_grad_from_views(means3D, campos, gcol, degree, M):
    """[P,3], [V,3], [V,P,3] -> [P,M,3]: sum over views of basis(dir) x gcol."""
    out = None
    for v in range(campos.shape[0]):
        d = means3D - campos[v]
        d = d / d.norm(dim=1, keepdim=True)
        t = sh_basis(d, degree, M)[:, :, None] * gcol[v][:, None, :]
        out = t if out is None else out + t
    return out
