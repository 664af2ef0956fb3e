import os, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import drq_oracle as O
import agent_helpers as AH
cfg = O.Config(image_keys=("a",), H=64, W=64, S=4, A=2)
st, core = AH.make_pair(cfg, B=6, trunk_mode="f16x3")
img = np.random.default_rng(1).integers(0, 256, (6, 64, 64, 3), dtype=np.uint8)
got = core.trunk_forward(torch.tensor(img, device="cuda")).cpu().numpy()
torch.cuda.synchronize()
ref = O.trunk_forward(st.trunk, torch.tensor(img), torch.float64).numpy()
print("err", AH.rel_err(got, ref), "nan", np.isnan(got).sum(), "absmax", np.abs(got).max(), np.abs(ref).max())
