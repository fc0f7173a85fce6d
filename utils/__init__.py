"""Import-path shim for the reference's `utils.*` helpers used by its example scripts (data / IO only)."""
