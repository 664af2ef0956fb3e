"""Networks with the reference's module names (serl_launcher/networks/): the reward classifier, inference only."""
