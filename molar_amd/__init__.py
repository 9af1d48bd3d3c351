"""molar_amd — MI355X-native engine for MolAR's per-frame hot path."""
