"""Module-level constants of the hot path (mirror of the reference's params/ package)."""
