"""Host-side helpers mirroring src/helpers of the reference (hot-path subset)."""
