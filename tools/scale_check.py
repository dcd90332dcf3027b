"""One-off scale check on the GPU box: a 100 000-frame file through hide_message (one call), through blocks
(mp3stego.sharded, 3 ranks played in one process) and cut into 2 000 short files through hide_messages; outputs compared
with each other and, on a sample of the short files, with the oracle."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'mp3-steganography-lib_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from mp3stego import _lib as mlib, sharded
import oracle_lib as orc
from synth_pcm import synth_pcm
ctx = mlib.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pcm = np.concatenate([synth_pcm(10000, seed=77 + i) for i in range(N // 10000)])
pcm[5000 * 1152:5040 * 1152] = 0
t0 = time.time(); enc = ctx.encode_pcm(pcm, 44100, 128, None)["mp3"]; print("encode_pcm %d frames: %.3f s, %d bytes" % (N, time.time() - t0, len(enc)))
del pcm
msg = "scale " * 2000
for _ in range(2):
    t0 = time.time(); whole = ctx.hide_message(enc, msg); t1 = time.time() - t0
print("hide_message one file: %.4f s = %.2f M frames/s, too_long %s" % (t1, N / t1 / 1e6, whole["too_long"]))
comm = sharded.LocalComm(3)
t0 = time.time()
for r in range(3):
    comm.rank = r
    res = sharded.reencode_sharded(ctx, enc, msg, comm)
print("sharded x3 (sequential in one process): %.4f s, equal %s" % (time.time() - t0, res["data"] == whole["data"]))
# host scan cost: the whole file, the index walk, and blocks scanned from the index (time should follow the block size)
import ctypes as C
L = mlib.lib()
buf = np.frombuffer(enc, dtype=np.uint8)
def best(f, k=5):
    b = 1e9
    for _ in range(k):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
def full_scan():
    o, sc = C.c_void_p(), mlib.Scanned()
    mlib.check(L.mp3s_scan_stream(buf.ctypes.data, len(enc), C.byref(o), C.byref(sc))); L.mp3s_buf_free(o)
print("host scan of the whole file: %.3f ms" % best(full_scan))
print("index walk of the whole file: %.3f ms" % best(lambda: mlib.StreamIndex(enc).close()))
ix = mlib.StreamIndex(enc)
for cnt in (N // 100, N // 10, N // 3, N):
    def rng():
        o, sc = C.c_void_p(), mlib.Scanned()
        mlib.check(L.mp3s_scan_range(buf.ctypes.data, len(enc), ix.handle, N // 2 - cnt // 2, cnt, C.byref(o), C.byref(sc))); L.mp3s_buf_free(o)
    print("scan of a block of %d frames in the middle of the file: %.3f ms" % (cnt, best(rng)))
comm = sharded.LocalComm(3)
t0 = time.time()
for r in range(3):
    comm.rank = r
    got = b"".join(bytes(c.tobytes()) for c in [np.concatenate(list(ctx.decode_chunks(enc, 16384)))]) if r == 0 else None
print("decode in chunks of 16384 frames (each scanned on its own): %.4f s" % (time.time() - t0))
for chunk in (16384, 50000):
    t0 = time.time(); ch = ctx.hide_message_chunked(enc, msg, chunk); t3 = time.time() - t0
    print("hide_message_chunked %d: %.4f s, equal %s" % (chunk, t3, bytes(ch["data"]) == bytes(whole["data"])))
p = mlib.scan_stream(enc)
cuts = np.concatenate([[0], np.cumsum(p["frame_size"].astype(np.int64))])
k = 50
shorts = [enc[int(cuts[a]):int(cuts[min(a + k, N)])] for a in range(0, N, k)]
notes = [("note %d " % i) * (1 + i % 5) if i % 7 else None for i in range(len(shorts))]
for _ in range(2):
    t0 = time.time(); out = ctx.hide_messages(shorts, notes); t2 = time.time() - t0
print("hide_messages %d files of %d frames: %.4f s = %.0f files/s" % (len(shorts), k, t2, len(shorts) / t2))
bad = sum(isinstance(o, Exception) for o in out)
ok = 0
for i in range(0, len(shorts), 97):
    d = orc.decode(shorts[i])
    bits = None if notes[i] is None else np.array(mlib.message_frame(notes[i]))
    o = orc.encode(orc.pcm_to_i16(d["pcm"]), 44100, 128, bits)
    ok += out[i]["data"] == o["mp3"]
print("errors", bad, "oracle sample equal", ok, "of", len(range(0, len(shorts), 97)))
