"""-m gpu: 1152- and 2304-sample blocks at the LPC presets on the fast kernels (round 6: evalg_kernel's runs of 16 k + 2 / + 4 samples,
pack2_kernel's 18-sample-run instance at any block that is whole passes of it) against the oracle, with the kernel record -- these
shapes used to run on the general evaluation and pack kernels, 3.5x / 2.4x slower per sample than 4096-sample blocks
(profiles/archive/r05_j_order_rate_autoc4.txt).  The reference's presets put 1152 at -0..-2 only, but `-b 1152` / `-b 2304` at any level are
what its test script and users of low-latency streams ask for (test/test_streams.sh:172-219, stream_encoder.c:748-753)."""
import numpy as np
import pytest

import signals

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blocksize", [1152, 2304, 4608, 3456])
@pytest.mark.parametrize("level", [3, 5, 6, 8])
@pytest.mark.parametrize("ch", [1, 2])
def test_other_block_sizes_at_the_lpc_presets(blocksize, level, ch, monkeypatch):
    _run(blocksize, level, ch, 16, monkeypatch)


@pytest.mark.parametrize("blocksize", [1152, 2304])
@pytest.mark.parametrize("level", [3, 8])
@pytest.mark.parametrize("ch,bps", [(2, 24), (6, 24), (3, 20), (2, 17), (2, 8)])
def test_other_block_sizes_other_sample_widths(blocksize, level, ch, bps, monkeypatch):
    """what evalg_kernel does not take at these block sizes -- channels of more than 16 bits -- must still come out right: evalw_kernel
    does not serve 18- and 36-sample runs (a 6-channel 24-bit case of the adversarial sweep found it launched on them)"""
    _run(blocksize, level, ch, bps, monkeypatch)


def _run(blocksize, level, ch, bps, monkeypatch):
    import flac_amd
    from oracle_from_settings import oracle_encode_settings
    monkeypatch.setenv("FLACGPU_POISON", "1")
    s = flac_amd.make_settings(ch, bps, 44100, level, blocksize=blocksize, streamable_subset=0)
    rng = np.random.default_rng(blocksize + level)
    n = blocksize * 9 + 77
    fs = 1 << (bps - 1)
    for name, pcm in (("music", signals.music(n, ch, bps, seed=level)), ("noise", rng.integers(-fs, fs, size=(n, ch)).astype(np.int32)),
                      ("quiet", rng.integers(-3, 4, size=(n, ch)).astype(np.int32)), ("wasted", (signals.music(n, ch, bps, seed=9) >> 4) << 4)):
        eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=16)
        try:
            data, fb = eng.encode(pcm)
            ks = eng.last_batch_kernels()
        finally:
            eng.close()
        o = oracle_encode_settings(pcm, s)
        assert np.array_equal(fb, o["frame_bytes"]) and data == o["data"], (name, blocksize, level, ch)
        if blocksize in (1152, 2304) and bps == 16:
            # (the short last block of the stream still takes the general kernels; whole blocks: the fast ones, whichever pack2 instance fits)
            assert "evalg_kernel" in ks and "pack2_kernel" in ks, (blocksize, level, ch, ks)
