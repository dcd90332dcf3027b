"""`Decoder` with the reference's signature and side effects (reference mp3stego/decoder/decoder.py:9-117).

One native call (mp3s_decode_file) turns the MP3 bytes into the WAV bytes scipy.io.wavfile.write would produce
for the int16 PCM; the reveal-string parse (decoder.py:86-108) is native too (mp3s_message_reveal).  The ID3v2
listing a non-quiet decode leaves in METADATA.txt (decoder.py:37-57) is host bookkeeping: mp3stego/decoder/id3.py.
"""
import os
import sys
import time


from mp3stego import _lib
from mp3stego.decoder import id3


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
        self.__tag = id3.read_tag(self.__data)     # IndexError for a file that ends inside its tag, as in the reference
        self.__result = None

    def decode(self, quiet: bool = True, reveal: bool = False, txt_file_path: str = "") -> int:
        """
        Decode the mp3 file into the wav file; with reveal=True also write the hidden string to txt_file_path.

        :return: the bitrate (kbps) of the last frame header, as the reference does.
        """
        if not quiet and self.__tag is not None:
            with open('METADATA.txt', 'w') as f:   # in the working directory, like the reference
                f.write(id3.listing(self.__file_path, self.__tag))
        start = time.time()
        # the WAV straight into the output file (over what is there, cut to length at the end): the library writes a file's PCM chunk by
        # chunk while the later chunks are still on the device -- 46 MB per 10 000 frames, the larger part of the call.  A stream the call
        # refuses leaves no file behind that was not there (the reference raises before it writes)
        there = os.path.exists(self.__output_file_path)
        fd = os.open(self.__output_file_path, os.O_WRONLY | os.O_CREAT, 0o666)
        held = os.fstat(fd).st_size             # a file that holds something is only written when the call has succeeded (mp3s_decode_file_fd)
        try:
            res = _lib.default_context().decode_file_to_fd(self.__data, fd)
        except BaseException as e:
            # whatever ends the call (the library's refusal, an interrupt, no memory): no output is left behind that was not there; one
            # that was there and empty is empty again (chunks may have gone to it); one that held something has not been touched -- the
            # reference raises while it parses, before write_to_wav (decoder.py:59-84), and leaves the user's file as it was
            os.close(fd); fd = -1
            if not there:
                os.remove(self.__output_file_path)
            elif held == 0:
                os.truncate(self.__output_file_path, 0)
            if isinstance(e, _lib.Mp3sError) and e.code in (_lib.E_MALFORMED, _lib.E_UNSUPPORTED):
                raise ValueError(str(e)) from None
            raise
        finally:
            if fd >= 0:
                os.close(fd)
        self.__result = res
        if not quiet:
            print('\nParsed', res["n_frames"], 'frames in', time.time() - start, 'seconds.')
        if not quiet:
            print(f"Wav file created on {self.__output_file_path}")

        if reveal:
            if txt_file_path[-4:] != '.txt':
                sys.exit("txt_file_path must be txt file.")
            with open(txt_file_path, 'wb') as f:
                f.write(_lib.message_reveal(res["bits"]))

        return res["kbps"]

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
