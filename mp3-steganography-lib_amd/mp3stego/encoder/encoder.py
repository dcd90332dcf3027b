"""`Encoder` with the reference's signature and behaviour (reference mp3stego/encoder/encoder.py:8-58,
MP3_Encoder.py:596-618): WAV in, MP3 out, optional '0'/'1' string hidden in the Huffman-table choice.
WAV parsing, analysis filterbank, MDCT, rate loop and bit packing all sit behind one native call (mp3s_encode_file).
"""
import os
import sys

import numpy as np

from mp3stego import _lib
from mp3stego.encoder.wav_reader import WavReader


class Encoder:
    """
    Creates an mp3 file from a wav file.

    :param file_path: the wav file path.
    :param output_file_path: the mp3 output file path.
    :param bitrate: the bitrate (kbps) of the output
    :param hide_str: if not empty, a string of '0'/'1' hidden inside the output mp3 file.
    """

    def __init__(self, file_path: str, output_file_path: str, bitrate: int = 320, hide_str: str = ''):
        self.__file_path = file_path
        self.__output_file_path = output_file_path
        if not os.path.exists(self.__file_path):
            sys.exit(f'File {self.__file_path} not found.')
        self.__wav_file = WavReader(self.__file_path, bitrate)
        self.__hide_str = hide_str
        self.hide_str_offset = 0

    def encode(self, quiet: bool = True) -> bool:
        """
        Encode the wav file into the mp3 file.

        :return: True if the message is too long for this file (it has been trimmed).
        """
        if not quiet:
            # what MP3Encoder.print_info (reference MP3_Encoder.py:581-594) reports for an MPEG-1 file of this encoder
            w = self.__wav_file
            print(f"MPEG-I layer III, {'mono' if w.num_of_channels == 1 else 'stereo'} Psychoacoustic Model: Shine")
            print(f"Bitrate: {w.bitrate} kbps De-emphasis: none\tOriginal\t")
            print(f"Encoding \"{w.file_path}\" to \"{w.file_path[:-3]}mp3\"\n")
        hide = np.frombuffer(self.__hide_str.encode("ascii"), dtype=np.uint8) - ord("0") if self.__hide_str else None
        try:
            res = _lib.default_context().encode_file(self.__wav_file.data, self.__wav_file.bitrate, hide)
        except _lib.Mp3sError as e:
            # mono input, a partial last frame (the reference steps its cursor by 2 per sample whatever the channel count
            # and reads past the buffer: SURVEY Appendix A, E3) and a quantizer step off its table all end in IndexError
            if e.code in (_lib.E_UNSUPPORTED, _lib.E_STEP_RANGE):
                raise IndexError(e.text) from None
            raise
        self.hide_str_offset = int(res["hide_offset"])
        with open(self.__output_file_path, "wb") as f:
            f.write(res["data"])
        too_long = self.hide_str_offset < len(self.__hide_str) - 1
        if not quiet:
            if too_long:
                print("File too short for this message length, your message has been trimmed.")
            print(f"MP3 file created on {self.__output_file_path}")
        return too_long
