"""placeholder for `flax.linen.module` (the reference annotates `pretrained_encoder: nn.module = None`)."""
