import sys, time, types, math, torch
sys.path.insert(0, "/root/repo/ad-gs_amd")
from adgs import env
import torch.nn.functional as F
e = env.EnvironmentMap(8192, 3)
w2v = torch.eye(4); w2v[:3, :3] = torch.tensor([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])
cam = types.SimpleNamespace(FoVx=0.87, image_width=1920, image_height=1280, world_view_transform=w2v.cuda(), cam_id=3)
up = torch.randn(3, 1280, 1920).cuda()
def run(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def hip():
    e.grid_map.grad = None; e.get_image_background(cam).backward(up)
# the reference's arithmetic with torch ops (scene/env.py:11-76), rays cached like the reference does
focal = env.fov2focal(0.87, 1920)
K = torch.tensor([[focal, 0, 960.0], [0, focal, 640.0], [0, 0, 1]], device="cuda")
grid = torch.stack(torch.meshgrid(torch.arange(0, 1920, dtype=torch.float32, device="cuda"), torch.arange(0, 1280, dtype=torch.float32, device="cuda"), indexing="xy"), -1)
rays = F.normalize((torch.inverse(K) @ torch.cat([grid, torch.ones_like(grid[..., :1])], -1)[..., None])[..., 0], dim=-1)
scale = torch.tensor([1 / math.pi, 2 / math.pi], device="cuda")
def tor():
    e.grid_map.grad = None
    v = F.normalize((cam.world_view_transform[:3, :3] @ rays[..., None]).squeeze(-1), dim=-1)
    x, y, z = v[..., 0:1], v[..., 1:2], v[..., 2:3]
    ang = torch.cat([torch.arctan2(y, x), torch.arctan2(z, torch.hypot(x, y))], -1) * scale
    torch.sigmoid(F.grid_sample(e.grid_map, ang[None], align_corners=True)).squeeze(0).backward(up)
print("HIP env background fwd+bwd (8192^2 map, dense grad) %.3f ms" % run(hip))
print("torch ops (reference arithmetic) fwd+bwd %.3f ms" % run(tor))
e.training_setup(types.SimpleNamespace(env_lr=1e-2)); hip()
print("FusedAdam over the 8192^2 x 3 map %.3f ms" % run(lambda: e.optimizer.step()))
