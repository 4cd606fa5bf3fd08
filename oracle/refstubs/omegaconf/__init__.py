"""TEST-ONLY stub (see pytorch_lightning stub)."""
