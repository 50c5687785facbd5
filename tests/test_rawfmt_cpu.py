"""CPU: the numpy restatement of the reference CLI's format_input() is self-consistent (to_raw is its inverse) and
agrees with hand-checked byte patterns of src/flac/encode.c:2352-2492."""
import numpy as np
import pytest

from rawfmt import format_input, to_raw


def test_known_byte_patterns():
    # 16-bit little endian signed: bytes 0x34 0x12 -> 0x1234; 0xff 0xff -> -1
    assert format_input(bytes([0x34, 0x12, 0xff, 0xff]), 2, 16).tolist() == [[0x1234, -1]]
    # 16-bit big endian
    assert format_input(bytes([0x12, 0x34, 0x80, 0x00]), 2, 16, big_endian=True).tolist() == [[0x1234, -32768]]
    # 8-bit unsigned: 0x00 -> -128, 0xff -> 127 (encode.c: - 0x80)
    assert format_input(bytes([0x00, 0xff]), 1, 8, is_unsigned=True).tolist() == [[-128], [127]]
    # 24-bit little endian signed: 0xff 0xff 0x7f -> 8388607 ; 0x00 0x00 0x80 -> -8388608
    assert format_input(bytes([0xff, 0xff, 0x7f, 0x00, 0x00, 0x80]), 1, 24).tolist() == [[8388607], [-8388608]]
    # 24-bit big endian unsigned: 0x80 0x00 0x01 -> 1
    assert format_input(bytes([0x80, 0x00, 0x01]), 1, 24, big_endian=True, is_unsigned=True).tolist() == [[1]]
    # shift: a 12-bit sample left-justified in 16 bits
    assert format_input(bytes([0x30, 0x12]), 1, 16, shift=4).tolist() == [[0x123]]
    with pytest.raises(ValueError):
        format_input(bytes([0x31, 0x12]), 1, 16, shift=4)
    # channel map: input channel 0 -> output 1, input 1 -> output 0
    assert format_input(bytes([1, 0, 2, 0]), 2, 16, channel_map=[1, 0]).tolist() == [[2, 1]]


@pytest.mark.parametrize("bits", [8, 16, 24, 32])
@pytest.mark.parametrize("be", [False, True])
@pytest.mark.parametrize("uns", [False, True])
def test_round_trip(bits, be, uns):
    rng = np.random.default_rng(bits + 2 * be + uns)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    pcm = rng.integers(lo, hi + 1, size=(257, 3), dtype=np.int64).astype(np.int32)
    pcm[0] = lo
    pcm[1] = hi
    raw = to_raw(pcm, bits, be, uns)
    assert np.array_equal(format_input(raw, 3, bits, be, uns), pcm)
    cm = [2, 0, 1]
    assert np.array_equal(format_input(to_raw(pcm, bits, be, uns, channel_map=cm), 3, bits, be, uns, channel_map=cm), pcm)
