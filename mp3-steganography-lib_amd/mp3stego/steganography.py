"""The `Steganography` facade of the drop-in package.

Public surface, console texts, `sys.exit` messages and file side effects (the temporary WAV next to the input that is
written and removed again) are the reference's (mp3stego/steganography.py:10-182); the work is done by `Decoder` and
`Encoder`, i.e. by libmp3s_hip.so.  Organised around three private steps -- mp3 -> temporary wav, wav -> mp3, drop
the wav -- that the five public operations combine; hide_message / clear_file take the fused native call (PCM kept in
device memory) whenever nothing of the detour would be observable.
"""
import contextlib
import os
import sys
import threading
import warnings

from mp3stego.decoder.decoder import Decoder
from mp3stego.encoder.encoder import Encoder

_RULE = "#" * 18


def str_to_binary_str(string: str) -> str:
    """'0'/'1' text of the UTF-8 bytes of `string`, most significant bit first."""
    return "".join(f"{byte:08b}" for byte in string.encode("utf-8"))


def _must_exist(path: str):
    if not os.path.exists(path):
        sys.exit(f"File {path} not found.")


_helper = None
_pending = []          # what the helper thread was given and nobody has looked at yet
_pending_lock = threading.Lock()


def _reset_helper():
    """a forked child starts without the parent's helper thread (its executor would wait for a thread that is not there)"""
    global _helper, _pending_lock
    _helper = None
    _pending_lock = threading.Lock()     # (the parent's may have been held by a thread that does not exist here)
    _pending.clear()


if hasattr(os, "register_at_fork"):
    os.register_at_fork(after_in_child=_reset_helper)


def _settle():
    """what the helper thread has finished is taken off the list; a failure of such a deferred clean-up (the un-mapping of an EARLIER
    call's input) is reported as a warning -- it is not this call's failure and must not make a valid request fail"""
    with _pending_lock:
        done = [f for f in _pending if f.done()]
        for fut in done:
            _pending.remove(fut)
    for fut in done:
        err = fut.exception()
        if err is not None:
            warnings.warn(f"mp3stego: a deferred clean-up of an earlier call failed: {err!r}", RuntimeWarning, stacklevel=3)


def _later(fn):
    """run `fn` on the module's helper thread (made on first use); the result is looked at by the next call's _settle()"""
    global _helper
    if _helper is None:
        import atexit
        from concurrent.futures import ThreadPoolExecutor
        _helper = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mp3stego-aux")
        atexit.register(lambda: (_helper.shutdown(wait=True) if _helper is not None else None))
    fut = _helper.submit(fn)
    with _pending_lock:
        _pending.append(fut)
    return fut


def _store(path: str, out):
    """the result over what is there, cut to length at the end (truncating first gives every page back and takes it again)"""
    fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o666)
    try:
        done = 0
        while done < len(out):
            done += os.write(fd, out[done:])
        os.ftruncate(fd, len(out))
    finally:
        os.close(fd)


def _same_file(a: str, b: str) -> bool:
    """one file under two names (a symbolic or hard link to the input as the output) counts as in place"""
    try:
        return os.path.samefile(a, b)
    except OSError:
        return os.path.abspath(a) == os.path.abspath(b)


def _ends(path: str, ext: str) -> bool:
    return path[-4:] == ext


class Steganography:
    """
    Encode / decode between wav and mp3, hide a message in an mp3 file, reveal it, clear it.

    :param quiet: False prints what is going on.
    """

    def __init__(self, quiet: bool = True):
        self.quiet = quiet
        self._kbps = 0            # bitrate of the mp3 decoded last: what hide / clear re-encode at
        self._decoder = None

    # ---------------------------------------------------------------- console
    @contextlib.contextmanager
    def _banner(self, start: str, done: str):
        if not self.quiet:
            print(f"\n{_RULE}\n{start}")
        yield
        if not self.quiet:
            print(f"\n{done}\n{_RULE}")

    # ---------------------------------------------------------------- the three steps
    @staticmethod
    def _wav_beside(mp3_path: str, wav_path: str = "") -> str:
        """argument checks of every operation that reads an mp3; returns the wav path to use"""
        _must_exist(mp3_path)
        wav_path = wav_path or mp3_path[:-4] + ".wav"
        if not (_ends(mp3_path, ".mp3") and _ends(wav_path, ".wav")):
            sys.exit("input_file_path must be mp3 file, wav_file_path must be wav file.")
        return wav_path

    def _to_wav(self, mp3_path: str, wav_path: str, reveal_into: str = ""):
        self._decoder = Decoder(mp3_path, wav_path)
        self._kbps = self._decoder.decode(self.quiet, reveal=bool(reveal_into), txt_file_path=reveal_into)

    def _to_mp3(self, wav_path: str, mp3_path: str, kbps: int, message=None) -> bool:
        _must_exist(wav_path)
        if not (_ends(mp3_path, ".mp3") and _ends(wav_path, ".wav")):
            sys.exit("wav_file_path must be wav file, output_file_path must be mp3 file.")
        payload = ""
        if message is not None:
            # "<number of characters>#<message>": the count is in characters, the payload in UTF-8 bytes
            payload = str_to_binary_str(f"{len(message)}#{message}")
        return Encoder(wav_path, mp3_path, bitrate=kbps, hide_str=payload).encode(quiet=self.quiet)

    def _drop_wav(self):
        self._decoder.delete_wav_file()
        if not self.quiet:
            print("Wav file has been deleted.")

    # ---------------------------------------------------------------- public operations
    def encode_wav_to_mp3(self, wav_file_path: str, output_file_path: str, bitrate: int = 320):
        """wav file -> mp3 file at `bitrate` kbps."""
        with self._banner(f"Start Encoding {wav_file_path} to  {output_file_path}.", "Finished Encoding."):
            self._to_mp3(wav_file_path, output_file_path, bitrate)

    def decode_mp3_to_wav(self, input_file_path: str, wav_file_path: str = "") -> int:
        """mp3 file -> wav file (next to the input unless named); returns the stream's bitrate in kbps."""
        with self._banner(f"Start Decoding {input_file_path} to  {wav_file_path}.", "Finished Decoding."):
            self._to_wav(input_file_path, self._wav_beside(input_file_path, wav_file_path))
        return self._kbps

    def reveal_massage(self, input_file_path: str, txt_file_path: str):
        """write the message hidden in the mp3 file into the txt file."""
        with self._banner(f"Start Revealing hidden message in {input_file_path} to  {txt_file_path}.", "Finished Revealing."):
            wav = self._wav_beside(input_file_path)
            if not _ends(txt_file_path, ".txt"):
                sys.exit("txt_file_path must be txt file.")
            self._to_wav(input_file_path, wav, reveal_into=txt_file_path)
            self._drop_wav()

    def hide_message(self, input_file_path: str, output_file_path: str, message: str) -> bool:
        """output = input re-encoded with `message` in its Huffman table choices; True if it had to be cut short."""
        with self._banner(f"Start Hiding {message} in {output_file_path}.", "Finished Hiding."):
            cut = self._recode(input_file_path, output_file_path, message)
        return cut

    def clear_file(self, input_file_path: str, output_file_path: str):
        """output = input re-encoded with nothing hidden."""
        with self._banner(f"Start Cleaning {input_file_path} into {output_file_path}.", "Finished Cleaning."):
            self._recode(input_file_path, output_file_path, None)

    def _recode(self, mp3_in: str, mp3_out: str, message) -> bool:
        """decode + re-encode (reference steganography.py:153-159 / 178-181).  The quiet case is ONE native call
        (mp3s_hide_message / mp3s_clear_file): the PCM stays in device memory instead of travelling through a temporary
        WAV file -- byte-identical output.  What the reference's detour leaves behind is reproduced: a file that happened
        to sit at the temporary WAV's path is gone afterwards (it was overwritten, then deleted).  Whatever is observable
        beyond that -- the console texts and METADATA.txt of a non-quiet run, the WAV a failing encode leaves on disk,
        which exception a stream the reference cannot re-encode raises -- comes from the step-by-step path below."""
        wav = self._wav_beside(mp3_in)
        if self.quiet and _ends(mp3_out, ".mp3"):
            from mp3stego import _lib
            import mmap
            _settle()
            # the input mapped instead of read (0.4 ms per 4 MB less; the library's uploads take the pages as they come), the output
            # written over what is there and cut to length at the end (truncating first gives every page back and takes it again)
            mapped = None
            with open(mp3_in, "rb") as f:
                try:
                    # (MAP_POPULATE: the pages mapped in one go -- 0.09 ms per 4 MB -- instead of fault by fault under the library's upload: 0.19)
                    data = mapped = mmap.mmap(f.fileno(), 0, flags=mmap.MAP_SHARED | getattr(mmap, "MAP_POPULATE", 0), prot=mmap.PROT_READ)
                except (ValueError, OSError):    # (an empty file cannot be mapped)
                    data = f.read()
            fd, made = -1, False
            try:
                ctx = _lib.default_context()
                there = os.path.exists(mp3_out)
                if there and os.path.getsize(mp3_out) > 0:
                    # an output that holds something (the input itself, in place, included) is written when the call has succeeded -- over
                    # what is there, cut to length at the end (truncating first gives every page back and takes it again) -- below, beside
                    # the helper thread's un-mapping of the input
                    res = ctx.clear_file(data) if message is None else ctx.hide_message(data, message)
                else:
                    # a new (or empty) output: the library writes the file's chunks as they come down, the first while the later ones are
                    # still on the device (mp3s_hide_message_fd: a 100 000-frame file 14.3 -> 13.0 ms); after a call that failed the file
                    # is removed again -- the reference has not created its output at that point either
                    fd = os.open(mp3_out, os.O_WRONLY | os.O_CREAT, 0o666)
                    made = not there
                    res = ctx.recode_to_fd(data, message, fd)
            except _lib.Mp3sError:
                res = None                       # the step-by-step path decides what this looks like to the caller
            except BaseException:
                # (an interrupt, no memory: nothing half written stays behind, as after a refusal)
                if fd >= 0:
                    os.close(fd); fd = -1
                    if made:
                        os.remove(mp3_out)
                    else:
                        os.truncate(mp3_out, 0)
                raise
            finally:
                if mapped is not None:
                    # (the runtime registers the pages of a mapping it uploads from with the device; taking the mapping down undoes that in
                    # the driver: 0.34 ms per 4 MB -- on the helper thread, beside the write of the result.  When the output IS the input
                    # the mapping goes first: the caller may rewrite the file the moment this call returns)
                    if _same_file(mp3_in, mp3_out):
                        mapped.close()
                    else:
                        _later(mapped.close)
                if fd >= 0:
                    os.close(fd)
            if res is None and fd >= 0:
                if made:
                    os.remove(mp3_out)
                else:
                    os.truncate(mp3_out, 0)      # (it was there, and empty)
            if res is not None:
                self._kbps = res["kbps"]
                if "data" in res:                # (written here: the helper thread takes the input's mapping down meanwhile)
                    _store(mp3_out, memoryview(res["data"]).cast("B"))
                if os.path.exists(wav):
                    os.remove(wav)
                return bool(res["too_long"])
        self._to_wav(mp3_in, wav)
        cut = self._to_mp3(wav, mp3_out, self._kbps, message)
        self._drop_wav()
        return cut
