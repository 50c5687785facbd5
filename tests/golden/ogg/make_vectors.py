"""Known-answer vectors for the Ogg paging restatement (flac_amd/csrc/host/ogg.c): Ogg FLAC logical streams written by the
reference WITH libogg, found in the reference tree's fuzzing seed corpus (oss-fuzz/seedcorpus/...).  Each logical stream is cut
out page by page (pages of other serial numbers dropped) and committed as a small fixture; the test re-pages its packets with
our code and expects the same bytes.  Run in the build container: python tests/golden/ogg/make_vectors.py"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [("/root/reference/oss-fuzz/seedcorpus/fuzzer_tool_flac/replaygain-which-is-not-lossless-ogg.fuzz", None),
       ("/root/reference/oss-fuzz/seedcorpus/fuzzer_seek/chained-and-multiplexed-with-vorbis-and-skeleton.fuzz", None)]


def pages(d):
    pos = d.find(b"OggS")
    while pos >= 0 and pos + 27 <= len(d) and d[pos:pos + 4] == b"OggS":
        nseg = d[pos + 26]
        n = 27 + nseg + sum(d[pos + 27:pos + 27 + nseg])
        yield d[pos:pos + n]
        pos += n


def main():
    k = 0
    for fn, _ in SRC:
        d = open(fn, "rb").read()
        streams = {}
        for pg in pages(d):
            serial = pg[14:18]
            streams.setdefault(serial, []).append(pg)
        for serial, pgs in streams.items():
            if not pgs[0][27 + pgs[0][26]:].startswith(b"\x7fFLAC"):
                continue                      # vorbis / skeleton streams of the multiplexed file
            if not pgs[-1][5] & 4:
                continue                      # a stream cut short by the fuzzer
            out = os.path.join(HERE, "oggflac_%d.bin" % k)
            open(out, "wb").write(b"".join(pgs))
            print(out, len(pgs), "pages")
            k += 1


if __name__ == "__main__":
    main()
