import sys
sys.path.insert(0, '.')
import numpy as np
from daisyriot_amd import api, scenes
N = 65536
sc = scenes.cornell_box(N, S=8)
uv = scenes.visibility_samples(50)
c = api.Context(0)
c.set_mesh(sc.vertices, sc.normals, sc.tri_v, sc.tri_n)
c.assemble(uv)
tot = 0; zero8 = 0; zero32 = 0; nnz = 0; z1 = 0; z1_64 = 0; z8_64 = 0; z1_1k=0
for r0 in range(0, N, 4096):
    F = c.read_rows(r0, 256)
    nz = F != 0
    nnz += nz.sum(); tot += nz.size
    b8 = nz.reshape(32, 8, N // 256, 256).any(axis=(1, 3))
    b32 = nz.reshape(8, 32, N // 256, 256).any(axis=(1, 3))
    zero8 += (~b8).sum() * 8 * 256; zero32 += (~b32).sum() * 32 * 256
    z1 += (~nz.reshape(256, N // 256, 256).any(axis=2)).sum() * 256
    z1_64 += (~nz.reshape(256, N // 64, 64).any(axis=2)).sum() * 64
    z8_64 += (~nz.reshape(32, 8, N // 64, 64).any(axis=(1, 3))).sum() * 8 * 64
    z1_1k += (~nz.reshape(256, N // 1024, 1024).any(axis=2)).sum() * 1024
print("1x256", z1/tot, "1x64", z1_64/tot, "8x64", z8_64/tot, "1x1024", z1_1k/tot)
print("nnz frac", nnz / tot, "zero in 8x256 blocks", zero8 / tot, "zero in 32x256 blocks", zero32 / tot)
