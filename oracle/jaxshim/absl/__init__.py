"""empty stand-in for absl (serl_launcher/common/wandb.py imports absl.flags)."""
