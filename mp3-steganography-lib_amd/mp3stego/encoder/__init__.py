"""WAV in, MP3 out: `Encoder` (the reference's class of that name, backed by libmp3s_hip.so) and the WAV header
reader it uses.  Nothing else is exported: the encoder's tables and stages live in the native library."""
from mp3stego.encoder.encoder import Encoder

__all__ = ["Encoder"]
