#!/usr/bin/env python3
"""Where K1's LDS bank conflicts come from (no GPU needed).  The FFT / untangling accesses are proven conflict-free by
tests/test_logmel_emu.py; this replays the ONE stage outside that proof -- the banded mel reduction (four lanes per filter, sixteen
filters per item, ds_read_b32: two 32-lane halves over 32 banks) -- under the same bank model and prints its ideal and conflict
cycles per frame next to the conflict-free stages' cycle count.  Measured (profiles/r03/logmel_pmc_summary.json): 109 M conflict
cycles of 289 M LDS-active = 38 %; this model: 229 of ~740 = 31 % from the mel stage alone (the per-lane twiddle lookups are the
rest)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adt_str_amd.frontend import MelBands, melscale_fbanks

sr, n_mels = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16000, 128)
mb = MelBands.from_dense(melscale_fbanks(sr, 2048, n_mels, 20.0).numpy())
lo, cnt = mb.meta[:, 0], mb.meta[:, 1]
items = (n_mels + 15) // 16
width = lambda j: int(cnt[j]) if j < n_mels else 0
trips = [max((width(g + 16 * i) + 3) // 4 for g in range(16)) for i in range(items)]
# band stride of an item (in trips): the trip count, made odd since round 5 when the padded table still fits (logmel.hip); argv[3] = "even": the old layout
odd = 1 if (len(sys.argv) < 4 or sys.argv[3] != "even") and 64 * sum(t | 1 for t in trips) <= 2816 else 0
poff, off = np.zeros(16 * items, int), 0
for j in range(16 * items):
    poff[j] = off
    off += 4 * (trips[j // 16] | odd)


def cycles(addrs):
    tot = 0
    for h in range(2):
        per = {}
        for a in addrs[32 * h:32 * h + 32]:
            per.setdefault(a % 32, set()).add(a)
        tot += max(len(v) for v in per.values())
    return tot


ideal = conf_p = conf_w = 0
for i in range(items):
    for t in range(trips[i]):
        first = lambda l: int(lo[(l >> 2) + 16 * i]) if (l >> 2) + 16 * i < n_mels else 0
        conf_p += cycles([first(l) + 4 * t + (l & 3) for l in range(64)]) - 2
        conf_w += cycles([int(poff[(l >> 2) + 16 * i]) + 4 * t + (l & 3) for l in range(64)]) - 2
        ideal += 2
fft = 16 * 4 + 16 * 2 + 16 * 2 + 16 * 4 + 8 * 4 + 16 * 4 + 8 * 8          # the accesses listed by emu_logmel2_accesses, cycles per kind
print(f"{sr} Hz, {n_mels} mels: trips per item {trips}, band stride {'odd' if odd else 'as the trips'}")
print(f"mel stage per frame: power reads {ideal} ideal + {conf_p} conflict cycles; weight reads {ideal} ideal + {conf_w} conflict cycles")
print(f"conflict-free FFT / window / untangling accesses: {fft} cycles per frame")
print(f"model conflict share: {(conf_p + conf_w) / (fft + 2 * ideal + conf_p + conf_w):.0%} of LDS-active cycles")
