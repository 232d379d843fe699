"""MI355X-native neural-process hot path."""
