"""Op registry, decoders and encoder of the NAS inner loop (mirrors src/nn)."""
