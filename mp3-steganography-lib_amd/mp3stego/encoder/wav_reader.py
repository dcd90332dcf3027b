"""Minimal WAV reader with the reference's checks, messages and quirks
(reference mp3stego/encoder/WAV_Reader.py:30-118): header searched in the first 128 bytes, PCM fmt
chunk of 16 bytes, 32/44.1/48 kHz only, samples always read as int16, trailing bytes kept."""
import struct
import sys

import numpy as np

_BITRATES_V1 = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320)


class WavReader:
    def __init__(self, file_path: str, bit_rate: int = 320):
        self.file_path = file_path
        self.bitrate = bit_rate
        with open(file_path, 'rb') as f:
            buffer = f.read(128)
            idx = buffer.find(b'RIFF')
            if idx == -1:
                sys.exit('Bad WAVE file.')
            if buffer.find(b'WAVE') == -1:
                sys.exit('Bad WAVE file.')
            idx = buffer.find(b'fmt ')
            if idx == -1:
                sys.exit('Bad WAVE file.')
            idx += 4
            if struct.unpack('<I', buffer[idx:idx + 4])[0] != 16:
                sys.exit('Unsupported WAVE file, compression used instead of PCM.')
            idx += 4
            if struct.unpack('<H', buffer[idx:idx + 2])[0] != 1:
                sys.exit('Unsupported WAVE file, compression used instead of PCM.')
            idx += 2
            self.num_of_channels = struct.unpack('<H', buffer[idx:idx + 2])[0]
            idx += 2
            self.samplerate = struct.unpack('<I', buffer[idx:idx + 4])[0]
            if self.samplerate not in (32000, 44100, 48000):
                sys.exit('Unsupported sampling frequency.')
            idx += 4 + 4 + 2
            self.bits_per_sample = struct.unpack('<H', buffer[idx:idx + 2])[0]
            if self.bits_per_sample not in (8, 16, 32):
                sys.exit('Unsupported WAVE file, samples not int8, int16 or int32 type.')
            idx = buffer.find(b'data')
            if idx == -1:
                sys.exit('Bad WAVE file.')
            idx += 4
            sub_chunk2_size = struct.unpack('<I', buffer[idx:idx + 4])[0]
            self.num_of_samples = int(sub_chunk2_size * 8 / self.bits_per_sample / self.num_of_channels)
            f.seek(idx + 4)
            self.buffer = np.fromfile(f, 'int16', self.num_of_samples * self.num_of_channels * 2)
        if self.bitrate not in _BITRATES_V1:
            sys.exit("Unsupported bitrate configuration.")
