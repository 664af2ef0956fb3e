"""empty stand-in (imported by serl_launcher/utils/launcher.py, never used on the update path)."""
