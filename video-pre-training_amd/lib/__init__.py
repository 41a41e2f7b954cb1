"""Host-side mirror of the reference's `lib/` interface for the hot path (policy, action heads, types)."""
