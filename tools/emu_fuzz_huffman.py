#!/usr/bin/env python3
"""The byte-identical code builder (zh_huffman.hip: the replay of deflate.nim:13-151 huffmanCodes, the heap by the
wave) against the oracle's huffman_codes on random histograms, on the CPU emulator: ties everywhere, a few huge
frequencies among ones, exponential ones, powers of two (trees deeper than the limit: the quicksort path) and plain
random ones; the three alphabets' sizes.      python tools/emu_fuzz_huffman.py [seed] [histograms]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)
import emu, oracle
eng = emu.engine()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 5)
bad = 0; N = int(sys.argv[2]) if len(sys.argv)>2 else 400
for it in range(N):
    n, minc, limit = rnd.choice(((286,257,15),(30,2,15),(19,19,7)))
    used = rnd.randrange(2, n+1)
    f = np.zeros(n, np.uint32)
    style = rnd.randrange(5)
    for i in rnd.sample(range(n), used):
        if style == 0: f[i] = rnd.randrange(1, 4)            # ties everywhere
        elif style == 1: f[i] = rnd.choice((1, 1, 2, 3, 5, 8, 1000, 100000))
        elif style == 2: f[i] = max(1, int(rnd.expovariate(1.0/rnd.choice((2,50,5000,400000)))))
        elif style == 3: f[i] = 1 << rnd.randrange(0, 22)    # deep trees: the limit path
        else: f[i] = rnd.randrange(1, 1<<22)
    wc, wl = oracle.huffman_codes(f, minc, limit)
    c, l = eng.debug_huffman(f, minc, limit, contract=False)
    if list(l) != list(wl) or list(c) != list(wc):
        bad += 1; print("MISMATCH", it, n, used, style)
print("huffman replay fuzz: %d histograms, %d bad" % (N, bad))
