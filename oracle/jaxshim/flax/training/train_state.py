"""flax.training.train_state.TrainState stand-in (the reward classifier's container, reward_classifier.py:62-66):
step / apply_fn / params / tx / opt_state with create(), replace() and apply_gradients().  TEST INFRASTRUCTURE ONLY."""
import dataclasses
from typing import Any, Callable


@dataclasses.dataclass
class TrainState:
    step: int
    apply_fn: Callable
    params: Any
    tx: Any
    opt_state: Any

    @classmethod
    def create(cls, *, apply_fn, params, tx, **kwargs):
        return cls(step=0, apply_fn=apply_fn, params=params, tx=tx, opt_state=tx.init(params), **kwargs)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def apply_gradients(self, *, grads, **kwargs):
        import optax
        updates, new_opt_state = self.tx.update(grads, self.opt_state, self.params)
        return self.replace(step=self.step + 1, params=optax.apply_updates(self.params, updates), opt_state=new_opt_state, **kwargs)
