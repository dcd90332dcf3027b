"""`Decoder` with the reference's signature and side effects (reference mp3stego/decoder/decoder.py:9-117).

The file is parsed on the host, the per-frame transforms run on the GPU (mp3s_decode_stream), the WAV
is written in the layout scipy.io.wavfile.write produces for int16 data, and the reveal-string parse
is the reference's (decoder.py:86-108).
"""
import os
import struct
import sys
import time

import numpy as np

from mp3stego import _lib


def _wav_bytes(pcm_i16: np.ndarray, rate: int) -> bytes:
    data = np.ascontiguousarray(pcm_i16, dtype="<i2")
    nch = 1 if data.ndim == 1 else data.shape[1]
    nb = data.nbytes
    return (b"RIFF" + struct.pack("<I", 36 + nb) + b"WAVE" + b"fmt " +
            struct.pack("<IHHIIHH", 16, 1, nch, rate, rate * nch * 2, nch * 2, 16) +
            b"data" + struct.pack("<I", nb) + data.tobytes())


class Decoder:
    """
    Creates a wav file from an mp3 file.

    :param file_path: the mp3 file path.
    :param output_file_path: the wav output file path.
    """

    def __init__(self, file_path: str, output_file_path: str):
        self.__file_path = file_path
        self.__output_file_path = output_file_path
        if not os.path.exists(self.__file_path):
            sys.exit(f'File {self.__file_path} not found.')
        with open(self.__file_path, 'rb') as f:
            self.__data = f.read()
        self.__result = None

    def decode(self, quiet: bool = True, reveal: bool = False, txt_file_path: str = "") -> int:
        """
        Decode the mp3 file into the wav file; with reveal=True also write the hidden string to txt_file_path.

        :return: the bitrate (kbps) of the last frame header, as the reference does.
        """
        start = time.time()
        try:
            res = _lib.default_context().decode_stream(self.__data, _lib.MP3S_PCM_I16)
        except _lib.Mp3sError as e:
            if e.code in (_lib.E_MALFORMED, _lib.E_UNSUPPORTED):
                raise ValueError(str(e)) from None
            raise
        self.__result = res
        if not quiet:
            print('\nParsed', res["n_frames"], 'frames in', time.time() - start, 'seconds.')
        with open(self.__output_file_path, "wb") as f:
            f.write(_wav_bytes(res["pcm"], res["sampling_rate"]))
        if not quiet:
            print(f"Wav file created on {self.__output_file_path}")

        if reveal:
            if txt_file_path[-4:] != '.txt':
                sys.exit("txt_file_path must be txt file.")
            bits = res["bits"]
            n = (len(bits) // 8) * 8
            by = np.packbits(bits[:n]) if n else np.zeros(0, dtype=np.uint8)
            output_str = ''.join(chr(int(b)) for b in by)
            message_len_str = ''
            for ch in output_str:
                if ch == '#':
                    break
                message_len_str += ch
            try:
                message_len = int(message_len_str)
            except Exception:
                message_len = 0
                message_len_str = ""
            if (len(message_len_str) + 1 + message_len) > len(output_str):
                output_str = output_str[len(message_len_str) + 1:]
            else:
                output_str = output_str[len(message_len_str) + 1: len(message_len_str) + 1 + message_len]
            with open(txt_file_path, 'wb') as f:
                f.write(bytes(output_str, 'utf-8'))

        return res["bit_rate"] // 1000

    @property
    def output_bits(self) -> str:
        """The extracted stego bit string ('0'/'1'), available after decode()."""
        if self.__result is None:
            return ""
        return "".join("1" if b else "0" for b in self.__result["bits"])

    def delete_wav_file(self):
        """Deletes the output wav file."""
        if os.path.exists(self.__output_file_path):
            os.remove(self.__output_file_path)
