"""GPU probe of the OPT-IN fused projection (SERL_PROJ_FUSE=1: a block's 1x1 stride-2 projection computed by conv0's workgroups,
conv_dma_f16x3_kernel<.., PROJ = true>).  Built at the end of round 4 without GPU time left: run this FIRST in round 5.
   python scripts/probes/proj_fuse_probe.py [n_images=1024]
Prints, for the same 128x128 images: the plan with and without the switch (b{1,2,3}_proj must read 'F' with it), the largest
feature difference between the two (expected: ~1e-7 of the scale -- the projection's products and their order are the same, only
the fp64 statistics atomics arrive in another order), the error of both against the fp64 oracle on 12 images (bound 5e-6), and
the time of 20 passes each."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import drq_oracle as O
import agent_helpers as AH

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = O.Config(image_keys=("a",), H=128, W=128, S=4, A=2)
st, core = AH.make_pair(cfg, B=n // 2, trunk_mode="f16x3")
img = torch.randint(0, 256, (n, 128, 128, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
out, plans = {}, {}
for sw in ("0", "1"):
    os.environ["SERL_PROJ_FUSE"] = sw
    out[sw] = core.trunk_forward(img).clone()
    plans[sw] = core.trunk_plan()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        core.trunk_forward(img)
    torch.cuda.synchronize()
    print(f"SERL_PROJ_FUSE={sw}: {(time.perf_counter() - t) / 20 * 1e3:.3f} ms per pass;",
          {k: v for k, v in plans[sw].items() if k.endswith("proj") or k.endswith("conv0")})
scale = float(out["0"].abs().max())
print("fused vs separate projection launches: max |diff| / scale =", float((out["1"] - out["0"]).abs().max()) / scale)
sel = list(range(6)) + list(range(n - 6, n))
ref = O.trunk_forward(st.trunk, img[sel].cpu(), torch.float64).numpy()
for sw in ("0", "1"):
    print(f"SERL_PROJ_FUSE={sw} vs fp64 oracle:", AH.rel_err(out[sw][sel].cpu().numpy(), ref))
ok = all(plans["1"].get(f"b{i}_proj", ("?",))[0] == "F" for i in (1, 2, 3))
print("projections fused:", ok)
