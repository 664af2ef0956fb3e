import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import drq_oracle as O
import agent_helpers as AH
for (H, n) in ((64, 6), (128, 5), (128, 70)):
    cfg = O.Config(image_keys=("a",), H=H, W=H, S=4, A=2)
    st, core = AH.make_pair(cfg, B=max(n, 4), trunk_mode="f16x3")
    img = np.random.default_rng(1).integers(0, 256, (n, H, H, 3), dtype=np.uint8)
    ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
    got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
    print(H, n, "err", AH.rel_err(got, ref), core.trunk_plan())
