"""WAV reader with the reference's checks, messages and quirks (reference mp3stego/encoder/WAV_Reader.py:30-118):
header searched in the first 128 bytes, PCM fmt chunk of 16 bytes, 32/44.1/48 kHz only, samples always read as
int16, trailing bytes kept.  The parse itself is native (mp3s_wav_parse); this class keeps the reference's shape."""
import sys

import numpy as np

from mp3stego import _lib


class WavReader:
    def __init__(self, file_path: str, bit_rate: int = 320):
        self.file_path = file_path
        self.bitrate = bit_rate
        with open(file_path, 'rb') as f:
            self.data = f.read()
        try:
            w = _lib.wav_parse(self.data, bit_rate)
        except _lib.Mp3sError as e:
            if e.code == _lib.E_EXIT:
                sys.exit(e.text)
            raise ValueError(str(e)) from None      # the reference dies in struct.unpack / a division here
        self.num_of_channels = w["channels"]
        self.samplerate = w["samplerate"]
        self.bits_per_sample = w["bits_per_sample"]
        self.num_of_samples = w["num_of_samples"]
        self.__data_offset, self.__n_values = w["data_offset"], w["n_values"]

    @property
    def buffer(self):
        """the int16 values np.fromfile yields in the reference (up to twice the declared count, cut by EOF)"""
        return np.frombuffer(self.data, dtype="<i2", count=self.__n_values, offset=self.__data_offset)
