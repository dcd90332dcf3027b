"""mp3stego -- drop-in for mp3stego-lib 1.1.8 whose per-frame transforms run on an MI355X.

Same public surface as the reference (reference mp3stego/__init__.py:1-4):
`Decoder`, `Encoder`, `Steganography`.  The classes call hand-written HIP kernels through the C-ABI
in include/mp3s.h; there is no CPU fallback for the transforms.
"""
from mp3stego.decoder import Decoder
from mp3stego.encoder import Encoder

from mp3stego.steganography import Steganography

__all__ = ["Decoder", "Encoder", "Steganography"]
