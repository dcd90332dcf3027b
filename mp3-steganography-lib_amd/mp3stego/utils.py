def safe_uint32(value):
    """Same helper the reference exports (mp3stego/utils.py:1-6): wrap negative ints into uint32."""
    if isinstance(value, int) and value < 0:
        return value & 0xFFFFFFFF
    return value
