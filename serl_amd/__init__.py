"""serl_amd -- MI355X-native learner hot path for rail-berkeley/serl."""
__version__ = "0.1.0"
