"""empty stand-in (imported, never used, by serl_launcher/utils/train_utils.py and common/typing.py)."""
