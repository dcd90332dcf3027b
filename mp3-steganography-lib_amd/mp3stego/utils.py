"""Helper the reference package exports beside its classes (reference mp3stego/utils.py); nothing here uses it."""

_U32_MASK = (1 << 32) - 1


def safe_uint32(value):
    """Negative Python ints come back as the uint32 with the same low 32 bits; everything else is returned as is."""
    negative_int = isinstance(value, int) and value < 0
    return (value & _U32_MASK) if negative_int else value
