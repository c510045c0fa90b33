"""Import-only stand-in (map preprocessing is out of scope for the oracle)."""
