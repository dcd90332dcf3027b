"""The reference's own five test cases (reference tests/steganography_test.py:15-60), run unchanged in
spirit against the drop-in package on the GPU, plus the error behaviour of the facade."""
import os
import shutil

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def work(tmp_path, golden_dir, mlib):
    shutil.copy(os.path.join(golden_dir, "test.mp3"), tmp_path / "test.mp3")
    os.chmod(tmp_path / "test.mp3", 0o644)
    return tmp_path


def test_decoder_encoder(work):
    from mp3stego import Steganography
    st = Steganography(quiet=True)
    bitrate = st.decode_mp3_to_wav(str(work / "test.mp3"), str(work / "out.wav"))
    assert bitrate == 320
    st.encode_wav_to_mp3(str(work / "out.wav"), str(work / "out.mp3"), bitrate)
    assert os.path.getsize(work / "out.mp3") == 37616 and os.path.getsize(work / "out.wav") == 165932


def test_hiding(work):
    from mp3stego import Steganography
    assert Steganography(quiet=True).hide_message(str(work / "test.mp3"), str(work / "out.mp3"), message='ddd') is False
    assert not os.path.exists(work / "test.wav")          # the temporary wav is deleted


def test_too_long_hiding(work):
    from mp3stego import Steganography
    assert Steganography(quiet=True).hide_message(str(work / "test.mp3"), str(work / "out.mp3"), message='ddd' * 100) is True


def test_reveal_hiding(work):
    from mp3stego import Steganography
    st = Steganography(quiet=True)
    st.hide_message(str(work / "test.mp3"), str(work / "out.mp3"), message='ddd')
    st.reveal_massage(str(work / "out.mp3"), str(work / "reveal.txt"))
    assert open(work / "reveal.txt").read() == 'ddd'


def test_reveal_cleared(work):
    from mp3stego import Steganography
    st = Steganography(quiet=True)
    st.hide_message(str(work / "test.mp3"), str(work / "out.mp3"), message='ddd')
    st.clear_file(str(work / "out.mp3"), str(work / "cleared.mp3"))
    st.reveal_massage(str(work / "cleared.mp3"), str(work / "reveal.txt"))
    assert open(work / "reveal.txt").read() == ''


def test_facade_errors(work):
    from mp3stego import Steganography, Decoder, Encoder
    st = Steganography()
    with pytest.raises(SystemExit) as e:
        st.decode_mp3_to_wav(str(work / "missing.mp3"))
    assert "not found" in str(e.value)
    with pytest.raises(SystemExit) as e:
        st.decode_mp3_to_wav(str(work / "test.mp3"), str(work / "x.txt"))
    assert str(e.value) == "input_file_path must be mp3 file, wav_file_path must be wav file."
    with pytest.raises(SystemExit) as e:
        st.reveal_massage(str(work / "test.mp3"), str(work / "x.wav"))
    assert str(e.value) == "txt_file_path must be txt file."
    open(work / "bad.wav", "wb").write(b"not a wave file")
    with pytest.raises(SystemExit) as e:
        Encoder(str(work / "bad.wav"), str(work / "o.mp3"))
    assert str(e.value) == "Bad WAVE file."
    with pytest.raises(SystemExit):
        Decoder(str(work / "nope.mp3"), str(work / "o.wav"))


def test_console_texts_when_not_quiet(work, capsys):
    """quiet=False prints the reference's banners (steganography.py:92-182) around each operation"""
    from mp3stego import Steganography
    from mp3stego.utils import safe_uint32
    st = Steganography(quiet=False)
    mp3, wav = str(work / "test.mp3"), str(work / "o.wav")
    st.decode_mp3_to_wav(mp3, wav)
    out = capsys.readouterr().out
    assert out.startswith(f"\n##################\nStart Decoding {mp3} to  {wav}.\n")
    assert "Parsed 36 frames in" in out and f"Wav file created on {wav}" in out
    assert out.endswith("\nFinished Decoding.\n##################\n")
    st.hide_message(mp3, str(work / "h.mp3"), "ddd")
    out = capsys.readouterr().out
    assert out.startswith(f"\n##################\nStart Hiding ddd in {work / 'h.mp3'}.\n")
    assert "Wav file has been deleted." in out and out.endswith("\nFinished Hiding.\n##################\n")
    st.clear_file(str(work / "h.mp3"), str(work / "c.mp3"))
    assert capsys.readouterr().out.startswith(f"\n##################\nStart Cleaning {work / 'h.mp3'} into {work / 'c.mp3'}.\n")
    with pytest.raises(SystemExit):
        st.reveal_massage(str(work / "c.mp3"), str(work / "r.doc"))           # no closing banner after an exit
    assert "Finished" not in capsys.readouterr().out
    assert safe_uint32(-1) == 0xFFFFFFFF and safe_uint32(5) == 5 and safe_uint32(-1.5) == -1.5


def test_a_file_that_does_not_start_with_a_frame(work):
    """no sync at the expected position: the reference parses nothing, writes the WAV scipy makes of an empty array at
    the header object's initial rate 0, and reports 0 kbps (MP3_Parser.py:37-46, 86-93); hiding then stops at the WAV
    reader's sampling-rate check"""
    import io
    import numpy as np
    from scipy.io import wavfile
    from mp3stego import Steganography
    junk = work / "junk.mp3"
    junk.write_bytes(b"\x00" * 1000)
    st = Steganography()
    assert st.decode_mp3_to_wav(str(junk), str(work / "junk.wav")) == 0
    ref = io.BytesIO()
    wavfile.write(ref, 0, np.zeros(0, dtype=np.int16))
    assert (work / "junk.wav").read_bytes() == ref.getvalue()
    st.reveal_massage(str(junk), str(work / "junk.txt"))
    assert (work / "junk.txt").read_bytes() == b""
    with pytest.raises(SystemExit) as e:
        st.hide_message(str(junk), str(work / "o.mp3"), "x")
    assert str(e.value) == "Unsupported sampling frequency."


def test_metadata_listing_of_a_tagged_file(work, golden_dir, monkeypatch):
    """a non-quiet decode of a file with an ID3v2 tag leaves METADATA.txt in the working directory (reference
    decoder/decoder.py:37-57, 73-74); a quiet one, or a file without a tag, does not; the audio decodes the same"""
    import json
    from mp3stego import Decoder
    monkeypatch.chdir(work)
    case = [r for r in json.load(open(os.path.join(golden_dir, "g8_id3.json"))) if r["name"] == "frame_flags_6"][0]
    data = bytes.fromhex(case["file_hex"])
    (work / "case.mp3").write_bytes(data)
    assert Decoder("case.mp3", "quiet.wav").decode(quiet=True) == 128 and not os.path.exists("METADATA.txt")
    assert Decoder("case.mp3", "loud.wav").decode(quiet=False) == 128
    assert open("METADATA.txt").read() == case["metadata"]
    assert (work / "loud.wav").read_bytes() == (work / "quiet.wav").read_bytes()
    os.remove("METADATA.txt")
    (work / "bare.mp3").write_bytes(data[case["offset"]:])
    assert Decoder("bare.mp3", "bare.wav").decode(quiet=False) == 128 and not os.path.exists("METADATA.txt")
    assert (work / "bare.wav").read_bytes() == (work / "loud.wav").read_bytes()


def test_fused_hide_equals_the_detour_through_the_wav(work, capsys):
    """quiet hide_message / clear_file are one native call (PCM stays in device memory); the non-quiet ones go decode ->
    WAV file -> encode as the reference does.  Same bytes, same return value, same files left behind."""
    from mp3stego import Steganography
    src = str(work / "test.mp3")
    for message in ("ddd", "d" * 400, None):
        (work / "test.wav").write_bytes(b"something that happened to lie at the temporary path")
        if message is None:
            Steganography(quiet=True).clear_file(src, str(work / "a.mp3"))
            cut_a = None
        else:
            cut_a = Steganography(quiet=True).hide_message(src, str(work / "a.mp3"), message)
        assert not os.path.exists(work / "test.wav")                  # overwritten and deleted by the reference, gone here too
        if message is None:
            Steganography(quiet=False).clear_file(src, str(work / "b.mp3"))
            cut_b = None
        else:
            cut_b = Steganography(quiet=False).hide_message(src, str(work / "b.mp3"), message)
        capsys.readouterr()
        assert cut_a == cut_b
        assert (work / "a.mp3").read_bytes() == (work / "b.mp3").read_bytes()
    # an output path the reference rejects only after it has decoded: the WAV stays behind, as there
    with pytest.raises(SystemExit):
        Steganography(quiet=True).hide_message(src, str(work / "out.xyz"), "ddd")
    assert os.path.exists(work / "test.wav")
