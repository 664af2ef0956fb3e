"""ctypes declarations of the agent half of the C ABI (include/serl_mi355.h)."""
import ctypes as C

from ._lib import SerlBatch


class SerlAgentCfg(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("n_cam", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("state_dim", C.c_int), ("act_dim", C.c_int), ("batch", C.c_int), ("ensemble", C.c_int),
        ("hidden", C.c_int), ("bottleneck", C.c_int), ("sle_features", C.c_int),
        ("proprio_dim", C.c_int), ("warmup_steps", C.c_int), ("temp_warmup_steps", C.c_int),
        ("discount", C.c_float), ("tau", C.c_float), ("lr", C.c_float), ("dropout", C.c_float),
        ("std_min", C.c_float), ("std_max", C.c_float), ("target_entropy", C.c_float),
        ("seed", C.c_uint64),
        # per-optimizer options of make_optimizer (common/optimizers.py:6-56), index = TX_INDEX[name]
        ("tx_lr", C.c_float * 3), ("tx_warmup", C.c_int * 3), ("tx_cosine_steps", C.c_int * 3),
        ("tx_weight_decay_on", C.c_int * 3), ("tx_weight_decay", C.c_float * 3), ("tx_clip_norm", C.c_float * 3),
        ("encoder_type", C.c_int),   # 0 = resnet-pretrained (frozen trunk), 1 = small (trainable SmallEncoder)
        ("critic_subsample_size", C.c_int),   # 0 = 2, -1 = None (all members), else 1..16
        ("backup_entropy", C.c_int),
        ("tx_lr_set", C.c_int * 3),   # != 0: tx_lr[t] given explicitly (0.0 is a valid optax learning rate)
    ]


TX_INDEX = {"actor": 0, "critic": 1, "temperature": 2}   # SERL_TX_*
NET_BITS = {"critic": 1, "actor": 2, "temperature": 4}   # SERL_NET_*


class SerlNoise(C.Structure):
    _fields_ = [
        ("eps_next", C.c_void_p), ("mask_next", C.c_void_p), ("redq_idx", C.c_void_p),
        ("eps_pi", C.c_void_p), ("mask_obs_pi", C.c_void_p),
        ("eps_temp", C.c_void_p), ("mask_next_temp", C.c_void_p),
        # jax.random keys (host uint32 words) instead of tensors: serl_mi355.h "Round 5"
        ("key_eps_next", C.c_void_p), ("key_mask_next", C.c_void_p), ("key_eps_pi", C.c_void_p), ("key_mask_obs_pi", C.c_void_p),
        ("key_eps_temp", C.c_void_p), ("key_mask_next_temp", C.c_void_p),
    ]


class SerlInfo(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "critic_loss", "predicted_qs", "target_qs", "actor_loss", "temperature", "entropy",
        "temperature_loss", "actor_lr", "critic_lr", "temperature_lr")]


def declare(lib):
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    P = C.POINTER
    sigs = {
        "serl_agent_create": [P(SerlAgentCfg), P(vp)],
        "serl_agent_destroy": [vp],
        "serl_agent_num_leaves": [vp],
        "serl_agent_leaf_info": [vp, i32, C.c_char_p, i32, P(i64)],
        "serl_agent_set": [vp, C.c_char_p, C.c_char_p, vp, i64],
        "serl_agent_get": [vp, C.c_char_p, C.c_char_p, vp, i64],
        "serl_agent_set_step": [vp, i64],
        "serl_agent_set_trunk_mode": [vp, i32],
        "serl_agent_set_chain_budget": [vp, i32],
        "serl_agent_update_critics": [vp, P(SerlBatch), P(SerlNoise), vp],
        "serl_agent_update_high_utd": [vp, P(SerlBatch), i32, P(SerlNoise), vp],
        "serl_agent_read_info": [vp, P(SerlInfo), vp],
        "serl_agent_encode": [vp, P(SerlBatch), vp],
        "serl_agent_encode_slot": [vp, P(SerlBatch), i32, vp],
        "serl_agent_encode_slot_range": [vp, P(SerlBatch), i32, i32, i32, vp],
        "serl_agent_select_slot": [vp, i32],
        "serl_agent_slot_features": [vp, i32, P(vp), P(i64)],
        "serl_agent_bind_slot": [vp, P(SerlBatch), i32],
        "serl_agent_critic_grads": [vp, i32, i32, i32, P(SerlNoise), i32, vp],
        "serl_agent_critic_grads_bucketed": [vp, i32, i32, i32, P(SerlNoise), i32, vp, vp],
        "serl_agent_grad_bucket": [vp, i32, P(vp), P(i64)],
        "serl_agent_actor_grads": [vp, i32, P(SerlNoise), vp],
        "serl_agent_apply": [vp, i32, f32, vp],
        "serl_agent_update": [vp, P(SerlBatch), i32, P(SerlNoise), vp],
        "serl_agent_begin_update": [vp, vp],
        "serl_agent_set_shard": [vp, i64, i64],
        "serl_agent_grad_view": [vp, i32, P(vp), P(i64)],
        "serl_agent_sample_actions": [vp, vp, vp, i32, vp, vp, vp],
        "serl_agent_trunk_forward": [vp, vp, i32, vp, vp],
        "serl_agent_debug_get": [vp, C.c_char_p, vp, i64],
        "serl_agent_trunk_plan": [vp, C.c_char_p, i32],
        "serl_agent_debug_set": [vp, C.c_char_p, vp, i64],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    lib.serl_debug_chain_launches.argtypes = []
    lib.serl_debug_chain_launches.restype = i64
    lib.serl_agent_get_step.argtypes = [vp]
    lib.serl_agent_get_step.restype = i64
