"""empty stand-in (imported, never used, by serl_launcher/utils/train_utils.py and common/typing.py)."""


class Tensor:   # common/typing.py: Array = Union[np.ndarray, jnp.ndarray, tf.Tensor]
    pass


class Variable:
    pass
