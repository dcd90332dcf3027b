"""One MP3 stream over several GPUs: one process per GPU, every rank takes a contiguous block of frames.

The reference has nothing of the kind (it is a single-threaded loop over frames); this is the multi-GPU form of its
decode -> re-encode path.  What crosses a block boundary:

* decode: the IMDCT overlap and the synthesis fifo reach back less than one frame (reference decoder/Frame.py:151-153,
  81-92), so a rank decodes one extra frame in front of its block and drops it (mp3s_decode_block);
* encode: analysis filter bank + MDCT need 1056 earlier PCM samples (encoder/MP3_Encoder.py:356, 685, 747): one frame
  of PCM in front of the block; the padding bit is a recurrence on the frame index (:630-636), replayed on the host;
  (both halves in one call, the PCM staying in HBM: mp3s_reencode_block);
* two serial chains of the rate loop -- the message cursor (:808-809) and the address1/2/3 + quantizerStepSize a silent
  granule inherits.  They are 17 integers.  A rank encodes its block on a guess (message already hidden,
  no inherited state), receives the real values from the rank before it, re-encodes only if the library says the
  block depended on them and the guess was wrong, and sends its own to the next rank.

No collective is on the data path: one 136-byte point-to-point message per boundary, and the gather of the finished
blocks on rank 0 ("host scatters, host concatenates").  Every rank reads the whole file (the host scan is a few
milliseconds per 10 000 frames); only its block's main data goes to its GPU.
"""
import numpy as np

from mp3stego import _lib

CARRY_WORDS = 17      # cursor + 4 x (address1, address2, address3, quantizerStepSize)
_PAST_MESSAGE = 0x3fffffff


def shard_frames(n_frames, rank, world):
    """contiguous blocks, sizes differing by at most one: (first, count) of `rank`"""
    base, rem = divmod(int(n_frames), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


class SoloComm:
    """world of one"""
    rank, world = 0, 1

    def send(self, dst, words):
        raise RuntimeError("no peer")

    def recv(self, src):
        raise RuntimeError("no peer")

    def gather(self, payload):
        return [payload]


class LocalComm:
    """`world` ranks played one after the other inside one process (the chain only ever looks at the rank before):
    tests and single-GPU dry runs of the multi-rank logic"""

    def __init__(self, world):
        self.world, self.rank = int(world), 0
        self._mail, self._parts = {}, []

    def send(self, dst, words):
        self._mail[dst] = np.array(words, dtype=np.int64)

    def recv(self, src):
        return self._mail.pop(self.rank)

    def gather(self, payload):
        self._parts.append(payload)
        return list(self._parts) if len(self._parts) == self.world else None


class TorchComm:
    """torch.distributed process group (gloo on CPU tensors, nccl = RCCL on the rank's GPU)"""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if device is None:
            device = "cuda" if dist.get_backend() == "nccl" else "cpu"
        self._device = device

    def send(self, dst, words):
        t = self._torch.tensor(np.asarray(words, dtype=np.int64), device=self._device)
        self._dist.send(t, dst)

    def recv(self, src):
        t = self._torch.zeros(CARRY_WORDS, dtype=self._torch.int64, device=self._device)
        self._dist.recv(t, src)
        return t.cpu().numpy()

    def gather(self, payload):
        out = [None] * self.world if self.rank == 0 else None
        self._dist.gather_object(payload, out, dst=0)
        return out


def decode_sharded(ctx, mp3: bytes, comm=None, out_format=_lib.MP3S_PCM_I16):
    """every rank decodes its block; rank 0 returns the dict Context.decode_stream returns, the others None"""
    comm = comm or SoloComm()
    # one walk over the frame headers (no main data copied); the rank then scans its own block only
    ix = _lib.StreamIndex(mp3)
    first, count = shard_frames(ix.n_frames, comm.rank, comm.world)
    part = None
    if count > 0:
        blk = ctx.decode_block(mp3, first, count, out_format, index=ix)
        # the block's stego bits without those of the halo frame decoded in front of it
        halo_bits = len(ix.scan_range(first - 1, 1)["bits"]) if first > 0 and ix.gpu_ok else 0
        part = (np.array(blk["pcm"]), np.array(blk["bits"][halo_bits:]) if ix.gpu_ok else None, np.array(blk["bits"]))
    parts = comm.gather(part)
    if parts is None:
        return None
    parts = [p for p in parts if p is not None]
    dt = {_lib.MP3S_PCM_I16: np.int16, _lib.MP3S_PCM_F32: np.float32, _lib.MP3S_PCM_F64: np.float64}[out_format]
    pcm = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, ix.channels), dtype=dt)   # (a stream without a frame)
    if parts and parts[0][1] is not None:
        bits = np.concatenate([p[1] for p in parts])
    else:                                            # host-parsed stream: every block call returned the whole stream's bits
        bits = parts[0][2] if parts else np.zeros(0, dtype=np.uint8)
    return {"n_frames": ix.n_frames, "channels": ix.channels, "sampling_rate": ix.sampling_rate,
            "bit_rate": ix.bit_rate, "pcm": pcm, "bits": bits}


def _same_effect(a, b, n_hide):
    """two carries a block cannot tell apart: equal chains, and cursors equal or both past the message"""
    return np.array_equal(a[1:], b[1:]) and min(int(a[0]), n_hide) == min(int(b[0]), n_hide)


def reencode_sharded(ctx, mp3: bytes, message=None, comm=None):
    """Steganography.hide_message (message: str) / clear_file (None) with the stream's frames spread over the ranks of
    `comm`.  Rank 0 returns {"data": mp3 bytes, "too_long", "hide_offset", ...} exactly as Context.hide_message /
    clear_file do for the whole file on one GPU; the other ranks return None."""
    comm = comm or SoloComm()
    n_hide = 0 if message is None else len(_lib.message_frame(message))
    ix = _lib.StreamIndex(mp3)         # the block is scanned on its own, also when it has to be encoded a second time
    if comm.rank == 0:
        blk = ctx.reencode_block(mp3, message, 0, comm.world, None, index=ix)
    else:
        guess = np.zeros(CARRY_WORDS, dtype=np.int64)
        guess[0] = _PAST_MESSAGE
        blk = ctx.reencode_block(mp3, message, comm.rank, comm.world, guess, index=ix)      # overlaps the ranks in front
    last_rank = min(comm.world, int(blk["total_frames"])) - 1                     # ranks behind it hold no frame
    if 0 < comm.rank <= last_rank:
        real = comm.recv(comm.rank - 1)
        live = min(int(real[0]), n_hide) < n_hide                    # the message is still being hidden at this boundary
        if not _same_effect(real, guess, n_hide) and (blk["carry_used"] or live):
            blk = ctx.reencode_block(mp3, message, comm.rank, comm.world, real, index=ix)
        else:
            # nothing in the block looked at the carry, so every chain entry it hands on is its own (a granule that
            # inherits would have set carry_used); only the count of tables seen so far moves with the real cursor
            out = blk["carry_out"].copy()
            out[0] = int(real[0]) + (int(out[0]) - int(guess[0]))
            blk["carry_out"] = out
            blk["hide_offset"] = int(out[0])
    if comm.rank < last_rank:
        comm.send(comm.rank + 1, blk["carry_out"])
    payload = None if comm.rank > last_rank else (blk["mp3"], int(blk["hide_offset"]), comm.rank == last_rank)
    parts = comm.gather(payload)
    if parts is None:
        return None
    parts = [p for p in parts if p is not None]
    hide_offset = [p[1] for p in parts if p[2]][0]
    return {"data": b"".join(p[0] for p in parts), "kbps": blk["kbps"], "sampling_rate": blk["sampling_rate"], "channels": 2,
            "n_frames": int(blk["total_frames"]), "too_long": hide_offset < n_hide - 1, "hide_offset": hide_offset}
