"""Genotype vocabulary shared with the reference's RL controller (src/rl)."""
