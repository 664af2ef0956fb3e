"""Generates tests/golden/update_*.npz by running the REFERENCE's own DrQAgent update code unmodified
(serl_launcher/agents/continuous/{drq,sac}.py, common/common.py, ... imported from /root/reference) under the
third-party stand-ins of oracle/jaxshim, in fp64.  Run in the build container (needs /root/reference):
    python tests/golden/make_golden_update.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import drq_oracle as O  # noqa: E402
from oracle import golden_update as G  # noqa: E402
from oracle import ref_update_runner as RR  # noqa: E402

PARAM_SEED, BATCH_SEED = 42, 100
CASES = {
    # CAR-style sequence: critic-only steps (zero-gradient Adam steps of actor/temperature), an actor+temperature
    # step, a UTD=2 scan -- 2x2 SpatialLearnedEmbeddings
    "drq_64_seq": (O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3), 8,
                   [("critics",), ("critics",), ("high_utd", 1), ("critics",), ("high_utd", 2)]),
    # the benchmark's image size (4x4 SLE) and dims, small batch
    "drq_128": (O.Config(image_keys=("front", "wrist"), H=128, W=128, S=24, A=6), 4, [("critics",), ("high_utd", 1)]),
    # one camera (the literal BASELINE.json configs[1] wording), non-square
    "drq_one_cam": (O.Config(image_keys=("image",), H=128, W=64, S=7, A=4), 6, [("high_utd", 1), ("critics",)]),
    # SACAgent.update (sac.py:243-299) with the default networks_to_update (all three losses at the same parameters, one
    # optimizer step) and other subsets
    "drq_update_subsets": (O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3), 6,
                           [("update", ("actor", "critic", "temperature")), ("update", ("critic", "actor")),
                            ("update", ("temperature",)), ("critics",), ("update", ("actor",))]),
    # make_optimizer's other branches (optimizers.py:14-21,36-37): warm-up -> cosine decay, clip_by_global_norm
    "drq_optimizers": (O.Config(image_keys=("front",), H=64, W=64, S=5, A=3,
                                opt={"critic": {"learning_rate": 1e-3, "warmup_steps": 3, "cosine_decay_steps": 6, "clip_grad_norm": 0.5},
                                     "actor": {"warmup_steps": 2}, "temperature": {"clip_grad_norm": 0.01, "learning_rate": 1e-2}}), 6,
                       [("critics",), ("update", ("actor", "critic", "temperature")), ("high_utd", 2), ("critics",), ("critics",), ("critics",)]),
    # the trainable SmallEncoder (vision/small_encoders.py:9-55 via create_drq(encoder_type="small"), run with the
    # 3-line kwarg adapter of oracle/ref_update_runner.py for the reference's own call-site defect): the critic loss
    # back-propagates into the conv kernels, the target critic uses the EMA'd encoder
    "drq_small_encoder": (O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3, encoder_type="small"), 6,
                          [("critics",), ("high_utd", 2), ("update", ("actor", "critic", "temperature")), ("critics",)]),
    # state-only SAC exactly as the reference's make_sac_agent builds it (launcher.py:50-76: warm-up 2000 for actor and
    # critic, none for the temperature) ...
    "sac_state": (O.Config(image_keys=(), S=10, A=4, discount=0.99, warmup=2000, temp_warmup=0), 8,
                  [("high_utd", 2), ("update", ("actor", "critic", "temperature")), ("high_utd", 1)]),
    # ... and with adamw / clipping / cosine decay (optimizers.py:39-42)
    "sac_state_adamw": (O.Config(image_keys=(), S=10, A=4, discount=0.99,
                                 opt={"actor": {"weight_decay": 0.01, "warmup_steps": 2},
                                      "critic": {"weight_decay": 0.003, "warmup_steps": 0, "clip_grad_norm": 1.0},
                                      "temperature": {"cosine_decay_steps": 5}}), 8,
                        [("high_utd", 2), ("update", ("actor", "critic", "temperature")), ("high_utd", 1), ("update", ("critic",))]),
    # create_drq / create_states options the launcher factories fix (sac.py:150-161,174-176): no REDQ subsampling (minimum
    # over all ten target members) with the entropy backup; a subsample of three
    "drq_backup_entropy_all": (O.Config(image_keys=("front",), H=64, W=64, S=5, A=3, subsample=None, backup_entropy=True), 6,
                               [("critics",), ("high_utd", 2), ("update", ("actor", "critic", "temperature"))]),
    "sac_state_subsample3": (O.Config(image_keys=(), S=10, A=4, discount=0.99, subsample=3, backup_entropy=True), 8,
                             [("high_utd", 2), ("update", ("critic",)), ("high_utd", 1)]),
    # ---- the reference's OWN random stream (case names ending in _threefry run under SERL_JAXSHIM_PRNG=threefry: the stand-in
    # jax.random then draws what jax.random draws, oracle/jaxshim/jax/threefry.py).  tests/test_golden_update_gpu.py runs the HIP
    # agent on these FROM THE SEED ONLY -- nothing injected -- and compares the integers it drew (crop offsets, REDQ indices,
    # Dropout masks) bit for bit, its normals to 1e-5, and the final state at the usual 1e-4.
    "drq_64_threefry": (O.Config(image_keys=("front", "wrist"), H=64, W=64, S=5, A=3), 8,
                        [("critics",), ("high_utd", 2), ("critics",), ("high_utd", 1)]),
    "drq_update_threefry": (O.Config(image_keys=("front",), H=64, W=64, S=5, A=3), 6,
                            [("update", ("actor", "critic", "temperature")), ("critics",), ("update", ("critic",)),
                             ("update", ("actor", "temperature"))]),
    "sac_state_threefry": (O.Config(image_keys=(), S=10, A=4, discount=0.99, warmup=2000, temp_warmup=0), 8,
                           [("high_utd", 2), ("update", ("actor", "critic", "temperature")), ("high_utd", 1)]),
}
ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (cfg, B, sched) in CASES.items():
        if ONLY and name not in ONLY:
            continue
        if name.endswith("_threefry"):
            os.environ["SERL_JAXSHIM_PRNG"] = "threefry"
        else:
            os.environ.pop("SERL_JAXSHIM_PRNG", None)
        res = RR.run_reference(cfg, B, sched, PARAM_SEED, BATCH_SEED)
        res["prng"] = os.environ.pop("SERL_JAXSHIM_PRNG", "philox")
        rec = G.pack(res, PARAM_SEED, BATCH_SEED)
        path = os.path.join(out_dir, f"update_{name}.npz")
        np.savez_compressed(path, **rec)
        print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB", "final step", res["final"]["step"],
              {k: round(v, 6) for k, v in res["steps"][-1]["info"].items()})


if __name__ == "__main__":
    main()
