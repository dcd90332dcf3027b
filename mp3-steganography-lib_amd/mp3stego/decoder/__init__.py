"""MP3 in, WAV out: `Decoder` (the reference's class of that name, backed by libmp3s_hip.so) and the ID3v2 listing
it writes when it is not quiet.  Frame parsing and the transforms live in the native library."""
from mp3stego.decoder.decoder import Decoder

__all__ = ["Decoder"]
