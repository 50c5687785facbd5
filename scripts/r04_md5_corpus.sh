#!/bin/bash
# the 10-hour corpus as 120 tracks: digests on the device against digests on the host (4 threads)
for m in device host; do
  python -m flac_amd.corpus --tracks 120 --md5 $m --md5-threads 4 2>/dev/null | tail -1
done
python -m flac_amd.corpus --tracks 1000 --md5 device 2>/dev/null | tail -1
