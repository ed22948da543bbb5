"""recogym_amd — MI355X-native vectorised reco-gym-v1 step loop."""
