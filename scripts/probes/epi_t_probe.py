"""Which epilogue mode of the row-major fused epilogue (SERL_EPI_T mask) disagrees with the C-layout one, and on which images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import agent_helpers as AH  # noqa: E402
from oracle import drq_oracle as O  # noqa: E402

n = 64
cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
st, core = AH.make_pair(cfg, B=32, trunk_mode="f16x3")
img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(8))
base = core.trunk_forward(img).clone()
scale = float(base.abs().max())
for mask in (1, 4, 8, 2, 15):
    os.environ["SERL_EPI_T"] = str(mask)
    got = core.trunk_forward(img).clone()
    d = (got - base).abs().reshape(n, -1).max(dim=1).values / scale
    bad = (d > 2e-6).nonzero().flatten().tolist()
    print(f"mask {mask}: max rel diff {float(d.max()):.3e}; images off: {len(bad)} of {n} {bad[:12]}; nan {bool(torch.isnan(got).any())}", flush=True)
os.environ.pop("SERL_EPI_T")
