import sys, time, torch
sys.path.insert(0, "/root/repo/ad-gs_amd")
from adgs import loss
gt = torch.rand(3, 1280, 1920).cuda(); img = (gt + 0.05 * torch.randn_like(gt)).requires_grad_(True)
def hip():
    img.grad = None; l1, s = loss.l1_ssim(img, gt); (0.8 * l1 + 0.2 * (1 - s)).backward()
for _ in range(3): hip()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): hip()
torch.cuda.synchronize(); print("HIP fused L1+SSIM fwd+bwd %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
