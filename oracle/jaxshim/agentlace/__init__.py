"""Import-time stand-in for agentlace (un-vendored, pinned at git cf2c337 by serl_launcher/setup.py:16): only the names
serl_launcher/utils/launcher.py and data/data_store.py import.  TEST INFRASTRUCTURE ONLY."""
