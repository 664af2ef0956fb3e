"""CPU: the Python surface of serl_amd against the reference's, name by name (north star: "keeps the serl_launcher.agents
Agent/TrainState API").  tests/golden/api_surface.json holds the parameter lists of the reference's public callables on the
learner path (tests/golden/make_api_surface.py reads them from the reference's source with `ast`).  Every one of them is
either MIRRORED -- ours exists where the table below says, takes the reference's parameters under the same names, in the same
order and kind, and adds at most parameters that have defaults -- or listed in NOT_MIRRORED with the reason; nothing is
silently absent, and a NOT_MIRRORED name that starts to exist fails the test until the table is updated."""
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))

# reference callable -> where ours lives
MIRRORED = {
    ("agents/continuous/drq.py", "DrQAgent.create_drq"): ("serl_amd.agents.drq", "DrQAgent.create_drq"),
    ("agents/continuous/drq.py", "DrQAgent.update_high_utd"): ("serl_amd.agents.drq", "DrQAgent.update_high_utd"),
    ("agents/continuous/drq.py", "DrQAgent.update_critics"): ("serl_amd.agents.drq", "DrQAgent.update_critics"),
    ("agents/continuous/sac.py", "SACAgent.create_states"): ("serl_amd.agents.sac", "SACAgent.create_states"),
    ("agents/continuous/sac.py", "SACAgent.update"): ("serl_amd.agents.sac", "SACAgent.update"),
    ("agents/continuous/sac.py", "SACAgent.update_high_utd"): ("serl_amd.agents.sac", "SACAgent.update_high_utd"),
    ("agents/continuous/sac.py", "SACAgent.sample_actions"): ("serl_amd.agents.sac", "SACAgent.sample_actions"),
    ("data/data_store.py", "MemoryEfficientReplayBufferDataStore.__init__"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.__init__"),
    ("data/data_store.py", "MemoryEfficientReplayBufferDataStore.insert"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.insert"),
    ("data/data_store.py", "MemoryEfficientReplayBufferDataStore.latest_data_id"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.latest_data_id"),
    ("data/data_store.py", "MemoryEfficientReplayBufferDataStore.get_latest_data"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.get_latest_data"),
    ("data/data_store.py", "ReplayBufferDataStore.__init__"): ("serl_amd.data.data_store", "ReplayBufferDataStore.__init__"),
    ("data/data_store.py", "ReplayBufferDataStore.insert"): ("serl_amd.data.data_store", "ReplayBufferDataStore.insert"),
    ("data/data_store.py", "ReplayBufferDataStore.latest_data_id"): ("serl_amd.data.data_store", "ReplayBufferDataStore.latest_data_id"),
    ("data/data_store.py", "ReplayBufferDataStore.get_latest_data"): ("serl_amd.data.data_store", "ReplayBufferDataStore.get_latest_data"),
    ("data/data_store.py", "populate_data_store"): ("serl_amd.data.data_store", "populate_data_store"),
    ("data/data_store.py", "populate_data_store_with_z_axis_only"): ("serl_amd.data.data_store", "populate_data_store_with_z_axis_only"),
    # the data stores' sample(*args, **kwargs) forwards to the buffer's sample under the lock (data_store.py:108-110): ours
    # is compared with the buffer's parameter list
    ("data/memory_efficient_replay_buffer.py", "MemoryEfficientReplayBuffer.sample"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.sample"),
    ("data/dataset.py", "Dataset.sample"): ("serl_amd.data.data_store", "ReplayBufferDataStore.sample"),
    ("data/dataset.py", "Dataset.seed"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.seed"),
    ("data/replay_buffer.py", "ReplayBuffer.get_iterator"): ("serl_amd.data.data_store", "MemoryEfficientReplayBufferDataStore.get_iterator"),
    ("utils/launcher.py", "make_drq_agent"): ("serl_amd.utils.launcher", "make_drq_agent"),
    ("utils/launcher.py", "make_sac_agent"): ("serl_amd.utils.launcher", "make_sac_agent"),
    ("utils/launcher.py", "make_replay_buffer"): ("serl_amd.utils.launcher", "make_replay_buffer"),
    ("utils/launcher.py", "make_trainer_config"): ("serl_amd.utils.launcher", "make_trainer_config"),
    ("utils/train_utils.py", "concat_batches"): ("serl_amd.utils.train_utils", "concat_batches"),
    ("utils/train_utils.py", "load_resnet10_params"): ("serl_amd.utils.train_utils", "load_resnet10_params"),
    ("networks/reward_classifier.py", "create_classifier"): ("serl_amd.networks.reward_classifier", "create_classifier"),
    ("networks/reward_classifier.py", "load_classifier_func"): ("serl_amd.networks.reward_classifier", "load_classifier_func"),
}

FLAX_MODULES = "takes flax module definitions (actor_def / critic_def / encoder_def); the network family is fixed to the launcher's and built by create_drq / create_states (DESIGN.md section 7)"
JAX_INTERNAL = "functional building block of the JAX update (apply_fn / grad_params plumbing); the update runs inside libserl_mi355.so behind update / update_critics / update_high_utd"
NOT_MIRRORED = {
    ("agents/continuous/drq.py", "DrQAgent.create"): FLAX_MODULES,
    ("agents/continuous/sac.py", "SACAgent.create"): FLAX_MODULES,
    ("agents/continuous/sac.py", "SACAgent.create_pixels"): FLAX_MODULES,
    ("agents/continuous/drq.py", "DrQAgent.data_augmentation_fn"): "the random shift is part of the fused gather kernel (serl_rb_gather_crop); crop offsets enter through `crops=`",
    ("agents/continuous/sac.py", "SACAgent.forward_critic"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.forward_target_critic"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.forward_policy"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.forward_temperature"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.temperature_lagrange_penalty"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.critic_loss_fn"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.policy_loss_fn"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.temperature_loss_fn"): JAX_INTERNAL,
    ("agents/continuous/sac.py", "SACAgent.loss_fns"): JAX_INTERNAL,
    ("data/data_store.py", "MemoryEfficientReplayBufferDataStore.sample"): "compared through the buffer's sample (see MIRRORED)",
    ("data/data_store.py", "ReplayBufferDataStore.sample"): "compared through the buffer's sample (see MIRRORED)",
    ("data/memory_efficient_replay_buffer.py", "MemoryEfficientReplayBuffer.__init__"): "the HBM store IS the data store: one class (its `image_keys` is the buffer's `pixel_keys`)",
    ("data/memory_efficient_replay_buffer.py", "MemoryEfficientReplayBuffer.insert"): "one class with the data store",
    ("data/replay_buffer.py", "ReplayBuffer.__init__"): "one class with the data store",
    ("data/replay_buffer.py", "ReplayBuffer.insert"): "one class with the data store",
    ("data/replay_buffer.py", "ReplayBuffer.download"): "RLDS / dataset download path: out of scope (DESIGN.md section 7)",
    ("data/replay_buffer.py", "ReplayBuffer.get_download_iterator"): "RLDS / dataset download path: out of scope",
    ("data/dataset.py", "Dataset.__init__"): "offline Dataset container: the replay stores own their storage",
    ("data/dataset.py", "Dataset.np_random"): "the generator lives in the C library (serl_rb_seed / serl_rb_rng_state); seed() mirrors Dataset.seed",
    ("data/dataset.py", "Dataset.sample_jax"): "JAX-array sampling of offline datasets: not on the learner path",
    ("data/dataset.py", "Dataset.split"): "offline dataset utilities: not on the learner path",
    ("data/dataset.py", "Dataset.filter"): "offline dataset utilities: not on the learner path",
    ("data/dataset.py", "Dataset.normalize_returns"): "offline dataset utilities: not on the learner path",
    ("utils/launcher.py", "make_bc_agent"): "BC agent: out of scope (DESIGN.md section 7)",
    ("utils/launcher.py", "make_vice_agent"): "VICE agent: out of scope (DESIGN.md section 7)",
    ("utils/launcher.py", "make_wandb_logger"): "wandb logging: out of scope (DESIGN.md section 7)",
    ("utils/train_utils.py", "load_recorded_video"): "video logging: out of scope",
}


def _resolve(module, qual):
    obj = importlib.import_module(module)
    for part in qual.split("."):
        obj = getattr(obj, part)
    return obj


def _ours(module, qual):
    obj = _resolve(module, qual)
    ps = list(inspect.signature(obj).parameters.values())
    kinds = {inspect.Parameter.POSITIONAL_ONLY: "positional", inspect.Parameter.POSITIONAL_OR_KEYWORD: "positional",
             inspect.Parameter.VAR_POSITIONAL: "var_positional", inspect.Parameter.KEYWORD_ONLY: "keyword_only",
             inspect.Parameter.VAR_KEYWORD: "var_keyword"}
    return [{"name": p.name, "kind": kinds[p.kind], "default": p.default is not inspect.Parameter.empty} for p in ps]


def test_every_reference_callable_is_accounted_for():
    listed = set(MIRRORED) | set(NOT_MIRRORED)
    for rel, entries in SURFACE.items():
        for qual in entries:
            assert (rel, qual) in listed, f"{rel}:{qual} is neither mirrored nor listed as out of scope"
    for key in listed:
        assert key[1] in SURFACE[key[0]], f"{key} is not a callable of the reference"
    assert not set(MIRRORED) & {k for k in NOT_MIRRORED if "see MIRRORED" not in NOT_MIRRORED[k]}


@pytest.mark.parametrize("key", sorted(MIRRORED), ids=lambda k: f"{k[0]}:{k[1]}")
def test_mirrored_callable_takes_the_reference_parameters(key):
    ref = [p for p in SURFACE[key[0]][key[1]]["params"] if p["name"] not in ("self", "cls")]
    ours = [p for p in _ours(*MIRRORED[key]) if p["name"] not in ("self", "cls")]
    by_name = {p["name"]: p for p in ours}
    # (a) every reference parameter exists under the same name and kind; a reference default stays a default
    for p in ref:
        if p["kind"] in ("var_positional", "var_keyword"):
            continue   # ours may spell the reference's *args / **kwargs out or keep them
        assert p["name"] in by_name, f"{key}: parameter `{p['name']}` missing (ours: {[q['name'] for q in ours]})"
        q = by_name[p["name"]]
        assert q["kind"] == p["kind"], f"{key}: `{p['name']}` is {q['kind']} here, {p['kind']} in the reference"
        if p["default"]:
            assert q["default"], f"{key}: `{p['name']}` has a default in the reference"
    # (b) same relative order of the shared positional parameters (positional calls mean the same thing)
    ref_pos = [p["name"] for p in ref if p["kind"] == "positional"]
    our_pos = [p["name"] for p in ours if p["kind"] == "positional" and p["name"] in ref_pos]
    assert our_pos == ref_pos, f"{key}: positional order {our_pos} vs reference {ref_pos}"
    # (c) whatever ours adds is optional and, if positional, comes after the reference's parameters
    names = {p["name"] for p in ref}
    extra = [p for p in ours if p["name"] not in names and p["kind"] in ("positional", "keyword_only")]
    for p in extra:
        assert p["default"], f"{key}: extra parameter `{p['name']}` has no default"
    pos_all = [p["name"] for p in ours if p["kind"] == "positional"]
    if ref_pos:
        last_ref = max(pos_all.index(n) for n in ref_pos)
        for p in extra:
            if p["kind"] == "positional":
                assert pos_all.index(p["name"]) > last_ref, f"{key}: extra positional `{p['name']}` sits in front of a reference parameter"


@pytest.mark.parametrize("key", sorted(k for k in NOT_MIRRORED if "see MIRRORED" not in NOT_MIRRORED[k] and "one class" not in NOT_MIRRORED[k]
                                        and not k[1].endswith("__init__")),
                         ids=lambda k: f"{k[0]}:{k[1]}")
def test_not_mirrored_names_are_really_absent(key):
    """(keeps the table honest: when one of these appears in serl_amd, move it to MIRRORED)"""
    homes = {"agents/continuous/drq.py": "serl_amd.agents.drq", "agents/continuous/sac.py": "serl_amd.agents.sac",
             "data/replay_buffer.py": "serl_amd.data.data_store", "data/dataset.py": "serl_amd.data.data_store",
             "utils/launcher.py": "serl_amd.utils.launcher", "utils/train_utils.py": "serl_amd.utils.train_utils"}
    qual = key[1]
    if key[0] in ("data/replay_buffer.py", "data/dataset.py"):
        qual = "MemoryEfficientReplayBufferDataStore." + qual.split(".")[1]
    with pytest.raises(AttributeError):
        _resolve(homes[key[0]], qual)
