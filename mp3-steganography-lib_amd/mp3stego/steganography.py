"""`Steganography` facade with the reference's surface, messages and file side effects
(reference mp3stego/steganography.py:10-182)."""
import os
import sys

from mp3stego import Decoder
from mp3stego import Encoder


def str_to_binary_str(string: str) -> str:
    """UTF-8 bytes of `string` as a string of bits, MSB first (reference steganography.py:10-24)."""
    return "".join(format(b, "08b") for b in string.encode('utf-8'))


class Steganography:
    """
    Encode and decode mp3/wav files, hide messages in mp3 files, reveal them and clear them.

    :param quiet: if False, prints information about the processes and the files.
    """

    def __init__(self, quiet: bool = True):
        self.quiet = quiet
        self.__encoder = None
        self.__decoder = None
        self.__bitrate = 0

    def __encode(self, wav_file_path, output_file_path, bitrate=320, quiet=True, hide=False, massage=""):
        binary_str = ""
        if hide:
            massage = str(len(massage)) + "#" + massage   # character count, UTF-8 payload
            binary_str = str_to_binary_str(massage)
        self.__encoder = Encoder(wav_file_path, output_file_path, bitrate=bitrate, hide_str=binary_str)
        return self.__encoder.encode(quiet=quiet)

    def __decode(self, input_file_path, wav_file_path, quiet=True, reveal=False, txt_file_path=""):
        self.__decoder = Decoder(input_file_path, wav_file_path)
        self.__bitrate = self.__decoder.decode(quiet, reveal=reveal, txt_file_path=txt_file_path)

    def __delete_wav_file(self, quiet=True):
        self.__decoder.delete_wav_file()
        if not quiet:
            print("Wav file has been deleted.")

    @staticmethod
    def __file_existence(file):
        if not os.path.exists(file):
            sys.exit(f'File {file} not found.')

    def __check_for_decoder(self, input_file_path, wav_file_path=""):
        self.__file_existence(input_file_path)
        if wav_file_path == '':
            wav_file_path = input_file_path[:-4] + ".wav"
        if input_file_path[-4:] != '.mp3' or wav_file_path[-4:] != '.wav':
            sys.exit("input_file_path must be mp3 file, wav_file_path must be wav file.")
        return wav_file_path

    def __check_for_encoder(self, wav_file_path, output_file_path):
        self.__file_existence(wav_file_path)
        if output_file_path[-4:] != '.mp3' or wav_file_path[-4:] != '.wav':
            sys.exit("wav_file_path must be wav file, output_file_path must be mp3 file.")

    def encode_wav_to_mp3(self, wav_file_path: str, output_file_path: str, bitrate: int = 320):
        """Encode a wav file into an mp3 file."""
        if not self.quiet:
            print(f"\n##################\nStart Encoding {wav_file_path} to  {output_file_path}.")
        self.__check_for_encoder(wav_file_path, output_file_path)
        self.__encode(wav_file_path, output_file_path, hide=False, bitrate=bitrate, quiet=self.quiet)
        if not self.quiet:
            print("\nFinished Encoding.\n##################")

    def decode_mp3_to_wav(self, input_file_path: str, wav_file_path: str = "") -> int:
        """Decode an mp3 file into a wav file; returns the bitrate (kbps)."""
        if not self.quiet:
            print(f"\n##################\nStart Decoding {input_file_path} to  {wav_file_path}.")
        wav_file_path = self.__check_for_decoder(input_file_path, wav_file_path)
        self.__decode(input_file_path, wav_file_path, reveal=False, quiet=self.quiet)
        if not self.quiet:
            print("\nFinished Decoding.\n##################")
        return self.__bitrate

    def reveal_massage(self, input_file_path: str, txt_file_path: str):
        """Write the string hidden in an mp3 file into a txt file."""
        if not self.quiet:
            print(f"\n##################\nStart Revealing hidden message in {input_file_path} to  {txt_file_path}.")
        wav_file_path = self.__check_for_decoder(input_file_path, "")
        if txt_file_path[-4:] != '.txt':
            sys.exit("txt_file_path must be txt file.")
        self.__decode(input_file_path, wav_file_path, reveal=True, quiet=self.quiet, txt_file_path=txt_file_path)
        self.__delete_wav_file(quiet=self.quiet)
        if not self.quiet:
            print("\nFinished Revealing.\n##################")

    def hide_message(self, input_file_path: str, output_file_path: str, message: str) -> bool:
        """Create output_file_path = input mp3 with `message` hidden in it; True if the message was trimmed."""
        if not self.quiet:
            print(f"\n##################\nStart Hiding {message} in {output_file_path}.")
        wav_file_path = self.__check_for_decoder(input_file_path, "")
        self.__decode(input_file_path, wav_file_path, reveal=False, quiet=self.quiet)
        self.__check_for_encoder(wav_file_path, output_file_path)
        too_long = self.__encode(wav_file_path, output_file_path, hide=True, bitrate=self.__bitrate,
                                 quiet=self.quiet, massage=message)
        self.__delete_wav_file(quiet=self.quiet)
        if not self.quiet:
            print("\nFinished Hiding.\n##################")
        return too_long

    def clear_file(self, input_file_path: str, output_file_path: str):
        """Create output_file_path = input mp3 re-encoded without any hidden string."""
        if not self.quiet:
            print(f"\n##################\nStart Cleaning {input_file_path} into {output_file_path}.")
        wav_file_path = self.__check_for_decoder(input_file_path, "")
        self.__decode(input_file_path, wav_file_path, reveal=False, quiet=self.quiet)
        self.__check_for_encoder(wav_file_path, output_file_path)
        self.__encode(wav_file_path, output_file_path, hide=False, bitrate=self.__bitrate, quiet=self.quiet)
        self.__delete_wav_file(quiet=self.quiet)
        if not self.quiet:
            print("\nFinished Cleaning.\n##################")
