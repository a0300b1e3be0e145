"""MI355X-native burst super-resolution hot path (drop-in for the reference package of the same name)."""
