"""vadx: MI355X-native batched VAD engine."""
