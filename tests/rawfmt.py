"""numpy restatement of format_input() of the reference's command-line tool (src/flac/encode.c:2352-2492) and its
inverse: the raw sample bytes a WAVE/AIFF/raw file of a given format would hold for an int32 block."""
import numpy as np


def to_raw(pcm, container_bits, big_endian=False, is_unsigned=False, shift=0, channel_map=None):
    """pcm int32 [n, C] (the values the encoder must see) -> raw bytes in the file's format.
    channel_map[c] = output channel that input channel c feeds (encode.c:2360-2367)."""
    x = pcm.astype(np.int64) << shift
    n, C = x.shape
    if channel_map is not None:
        src = np.empty_like(x)
        for c in range(C):
            src[:, c] = x[:, channel_map[c]]
        x = src
    if is_unsigned:
        x = x + (1 << (container_bits - 1))
    x = x & ((1 << container_bits) - 1)
    nb = container_bits // 8
    out = np.empty((n, C, nb), dtype=np.uint8)
    for b in range(nb):
        byte = (x >> (8 * b)) & 0xff
        out[:, :, (nb - 1 - b) if big_endian else b] = byte
    return out.reshape(-1)


def format_input(raw, channels, container_bits, big_endian=False, is_unsigned=False, shift=0, channel_map=None):
    """raw bytes -> int32 [n, C]; raises ValueError on non-zero bits below `shift` (encode.c:2479-2488)."""
    nb = container_bits // 8
    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, channels, nb).astype(np.int64)
    t = np.zeros(b.shape[:2], dtype=np.int64)
    for k in range(nb):
        t |= b[:, :, (nb - 1 - k) if big_endian else k] << (8 * k)
    if is_unsigned:
        t = t - (1 << (container_bits - 1))
    else:
        t = np.where(t >= (1 << (container_bits - 1)), t - (1 << container_bits), t)
    if channel_map is not None:
        out = np.empty_like(t)
        for c in range(channels):
            out[:, channel_map[c]] = t[:, c]
        t = out
    if shift:
        if np.any(t & ((1 << shift) - 1)):
            raise ValueError("non-zero least-significant bits")
        t = t >> shift
    return t.astype(np.int32)
