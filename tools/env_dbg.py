import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
from adgs import env
from oracle import env_oracle
G = np.load('/root/repo/tests/golden/env_golden.npz')
for case in sorted({k.split('/')[0] for k in G.files}):
    gm = torch.tensor(G[case + '/grid_map'])[None].cuda()
    H, W = [int(v) for v in G[case + '/HW']]
    bg = env.image_background(gm, H, W, float(G[case + '/focal']), G[case + '/R'].tolist()).cpu().numpy()
    ref = G[case + '/bg']
    o64 = env_oracle.background(G[case + '/grid_map'], H, W, float(G[case + '/focal']), G[case + '/R'])
    o32 = env_oracle.background(G[case + '/grid_map'], H, W, float(G[case + '/focal']), G[case + '/R'], dtype=np.float32)
    e = np.abs(bg - ref)
    print(case, 'hip-ref', e.max(), 'at', np.unravel_index(e.argmax(), e.shape), 'hip-o64', np.abs(bg - o64).max(), 'o32-ref', np.abs(o32 - ref).max(), 'frac>3e-5', (e > 3e-5).mean())
