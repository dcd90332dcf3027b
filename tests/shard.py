"""Frame sharding used by multi-GPU drivers: contiguous blocks, 1-frame halo in front of every block but the first
(the decoder/encoder state reaches back less than one frame)."""


def shard_frames(n_frames, rank, world):
    base, rem = divmod(n_frames, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    halo = 1 if (first > 0 and count > 0) else 0
    return first, count, halo
